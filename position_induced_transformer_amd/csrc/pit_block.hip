// Fused processor blocks for batch-free meshes, small (latency-bound) regime.
//
// The processor of pit.py:114-122 is n_blocks x [ posatt.forward (self attention on the latent mesh, locality 1:
// nothing masked, pit.py:102) -> kaiming_mlp -> gelu ].  For the batch-free operators (posatt_fixed / _periodic1d /
// _periodic2d) the softmax weights of EVERY block depend on the latent mesh and the block's lmda only - not on the
// activations - so they leave the dependent chain of the step:
//
//   block_weights_kernel  (one launch, all blocks)  E[l,h,n,j] = exp(-c_lh m[n,j])      (un-normalised, SYMMETRIC:
//                                                    m[n,j] = m[j,n] bit for bit, and S_min = 0 on the diagonal)
//                                                   Q[l,h,n,j] = E (m - mbar_n) / rowsum_n   (the d(scale) weights)
//                                                   1/rowsum, and rowstat = {T, S_min, 1/rowsum, mbar} as pit_posatt_fwd saves it
//   block_fwd_kernel      (one launch per block)    16-row slab of one sample: O_h = (E_h X) / rowsum  as plain
//                                                   v_mfma_f32_16x16x4_f32 contractions (operands straight from memory, no
//                                                   weight formation), the concat tile [X | O_0 | O_1] stays in LDS and feeds the
//                                                   block's MLP (mlp_fwd16_kernel's phases) in the same launch
//   block_bwd_kernel      (one launch per block)    16-key slab: d(values) = residual + sum_h E_h^T (dO_h / rowsum) - E is
//                                                   symmetric, so this is the same row-major contraction - followed by the data
//                                                   path of the PREVIOUS block's MLP backward (mlp_bwd16_kernel's phases);
//                                                   the block's d(scale) (Q X contracted with dO, fp64 partials into the layer's
//                                                   accumulators) and the weight-gradient reductions of the block's own MLP
//                                                   (`rider`, gemm_rd_body) ride along as extra workgroups.
//
// The forward chain of a block is ONE launch instead of two (attention, MLP), the backward chain ONE instead of two
// (MLP data path, attention pair), and the weights cost one launch per step instead of being re-formed by all 256
// workgroups of every attention launch (31-39 VALU instructions per MFMA in posatt_rows_kernel / posatt_bwd_pair).
// Exact fp32 products (PIT_MATH_FP32 only); per-sample meshes keep the recompute-from-coordinates kernels.
#include "pit_common.h"
#include "pit_gemm_rd.h"
#include "pit_block_dev.h"
#include <cstdlib>

namespace {

#ifdef PIT_STAMPS
// diagnostic build only (tools/block_bench.py): shader-clock stamps of wave 0 of one workgroup, never read by the kernels
__device__ unsigned long long pit_block_stamps[32];
#define BSTAMP(i_) do { if (blockIdx.x == 5 && threadIdx.x == 0) pit_block_stamps[i_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BSTAMP(i_) do { } while (0)
#endif

#ifdef PIT_WGREC
// per workgroup of the LAST block_bwd launch: entry, exit (100 MHz), HW_ID, XCC_ID (tools/stamp_block.py; -DPIT_WGREC build)
__device__ unsigned long long pit_block_wgrec[1024 * 4];
#define BREC(slot_) do { if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                       \
        pit_block_wgrec[blockIdx.x * 4 + (slot_)] = __builtin_amdgcn_s_memrealtime();                                       \
        if ((slot_) == 0) { pit_block_wgrec[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   \
                            pit_block_wgrec[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); } } } while (0)
#else
#define BREC(slot_) do { } while (0)
#endif
#ifdef PIT_STAMPS_DBG
__device__ int pit_block_dbg = 0;      // experiments (tools/block_bench.py): 1 = linear workgroup -> slab map, 2 / 4 = no B / A loads
#define PIT_BLOCK_DBG pit_block_dbg
#else
#define PIT_BLOCK_DBG 0
#endif
// timing experiments (tools/block_variants.sh; results void): 1 = forward contraction issues half its MFMAs (all loads kept),
// 2 = forward GEMM1 half its k-steps, 4 = backward contraction half its MFMAs, 8 = backward phase C one tile per wave,
// 32 = backward: the rider workgroups leave at once
#ifndef PIT_BLOCK_EXP
#define PIT_BLOCK_EXP 0
#endif

// ---------------------------------------------------------------------------------------------- weights (body: pit_block_dev.h)
__global__ __launch_bounds__(256) void block_weights_kernel(WeightsArgs a) { block_weights_body(a, blockIdx.x); }

// ---------------------------------------------------------------------------------------------- shared pieces
// One wave's share of a slab contraction: keys [jb, je) (32 per trip) for ALL NH heads.
//   acc[h or 0][t] (16 rows x 16 columns {4c + t}) += sum_key  W_h[row0 + i][key] * (sc_h[key]) * V_h[key][4c + t]
// W_h = w + h*wstride, row-major with pitch L (A operand: lane (i = l&15, kq = l>>4) takes 4 consecutive keys with one 16-B
// load - the key order inside a group of 16 is permuted identically for both operands); V rows ldv floats apart (B operand:
// one 16-B load = this lane's column of all four tiles).
//   SHARED (forward, d(scale)): every head contracts the SAME value rows - they are fetched once for all heads (the L2 -> CU
//     path, ~15 B/clk per CU, is what bounds these kernels: a workgroup of the first version pulled 160 KB, two thirds of it
//     the second head's copy of the value rows) and the heads keep separate accumulators;
//   !SHARED (d(values)): head h reads its own columns V + h*vhead, rows scaled by sc_h[key] (1/rowsum), one accumulator.
// All loads of a trip are requested before its first MFMA (sched_barrier: left to itself the scheduler sinks each load to
// its use and the trip becomes a chain of dependent memory round trips).
template <int NH, bool SHARED>
__device__ __forceinline__ void slab_contract(const float* __restrict__ w, long wstride, int L, const float* __restrict__ v,
                                              long ldv, int vhead, const float* __restrict__ sc, int jb, int je, int l15, int kq,
                                              f32x4_t (&acc)[SHARED ? NH : 1][4]) {
    constexpr int NS = SHARED ? 2 : 1;                 // 16-key groups per trip (d(values) fetches per-head rows: half trips
                                                       // keep the kernel within 128 VGPRs, the path is bandwidth-bound anyway)
    const float* wp = w + (long)l15 * L + 4 * kq;
    const float* vp = v + 4 * l15;
    for (int j0 = jb; j0 < je; j0 += 16 * NS) {
        float4 av[NH][NS], sv[NH][NS], bv[SHARED ? 1 : NH][NS][4];
        const int dbg = PIT_BLOCK_DBG;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                av[h][s] = (dbg & 4) ? make_float4(1.f, 2.f, 3.f, 4.f) : *reinterpret_cast<const float4*>(wp + h * wstride + j0 + 16 * s);
                if (!SHARED) sv[h][s] = *reinterpret_cast<const float4*>(sc + (long)h * L + j0 + 16 * s + 4 * kq);
            }
#pragma unroll
        for (int h = 0; h < (SHARED ? 1 : NH); ++h)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    bv[h][s][m] = (dbg & 2) ? make_float4(1.f, 2.f, 3.f, 4.f)
                                            : *reinterpret_cast<const float4*>(vp + h * vhead + (long)(j0 + 16 * s + 4 * kq + m) * ldv);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    if ((PIT_BLOCK_EXP & (SHARED ? 1 : 4)) && (SHARED ? s == 1 : h == 1)) {
                        asm volatile("" :: "v"(av[h][s].x), "v"(av[h][s].y), "v"(av[h][s].z), "v"(av[h][s].w));
                        asm volatile("" :: "v"(bv[SHARED ? 0 : h][s][m].x), "v"(bv[SHARED ? 0 : h][s][m].y), "v"(bv[SHARED ? 0 : h][s][m].z), "v"(bv[SHARED ? 0 : h][s][m].w));
                        if (!SHARED) asm volatile("" :: "v"(sv[h][s].x), "v"(sv[h][s].y), "v"(sv[h][s].z), "v"(sv[h][s].w));
                        continue;
                    }
                    const float* ap = &av[h][s].x;
                    const float* sp = &sv[h][s].x;
                    const float aw = SHARED ? ap[m] : ap[m] * sp[m];       // (row scale folded into the A operand)
                    const float4 bb = bv[SHARED ? 0 : h][s][m];
                    f32x4_t (&ac)[4] = acc[SHARED ? h : 0];
                    ac[0] = mfma_16x16x4(aw, bb.x, ac[0]);
                    ac[1] = mfma_16x16x4(aw, bb.y, ac[1]);
                    ac[2] = mfma_16x16x4(aw, bb.z, ac[2]);
                    ac[3] = mfma_16x16x4(aw, bb.w, ac[3]);
                }
    }
}

__device__ __forceinline__ bool slab_of(int id, int batch, int slabs, int& sample, int& slab) {
    if (PIT_BLOCK_DBG & 1) return slab_of_linear(id, batch, slabs, sample, slab);
    return slab_of_xcd(id, batch, slabs, sample, slab);
}

// ---------------------------------------------------------------------------------------------- forward
struct BlockFwdArgs {
    const float *e, *inv;               // this layer: (H, L, L), (H, L)
    int L, batch;
    float* xcat;                        // (batch*L, (1+H)*64): columns [0,64) given, head columns written here
    const float *w1, *b1, *w2, *b2;     // the block's MLP ((1+H)*64 -> 64 -> 64)
    int out_gelu;
    float *z1, *h, *z2, *y; long ldy;
};

// WLDS (round 5): the MLP's weights reach the MFMAs through LDS.  As B-operand fragments straight from memory (lane = weight row,
// 16 B of it per instruction) every wave instruction touches 16 different cache lines for 64 useful bytes each - 4 096 line
// look-ups per workgroup for W1 | W2 against 512 when all eight waves copy the 64 KB with unit-stride 16-byte loads; the copy lands
// in LDS before the barrier in front of GEMM1 and the fragments are ds_read_b128s.  Needs 64 KB more LDS (one workgroup per CU):
// taken when the launch has no more workgroups than the chip has CUs.
template <int H, bool WLDS>
__global__ __launch_bounds__(512, WLDS ? 1 : 2) void block_fwd_kernel(BlockFwdArgs g) {      // (WLDS: one workgroup per CU, 256 registers)
    constexpr int W = (1 + H) * BD;                     // concat width = K of the first contraction
    constexpr int XP = W + 4, HP = BD + 4;              // LDS pitches
    constexpr int KS = W / 16;
    constexpr int NW1 = BD * W / 4 / 512, NW2 = BD * BD / 4 / 512;      // 16-byte pieces of W1 / W2 per thread (6 / 2 at H = 2)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* pk = smem;                                   // [H * PARK_FLOATS]: slot wave*H + head
    float* xs = smem + H * PARK_FLOATS;                 // [16][XP] concat tile
    float* hs = xs + 16 * XP;                           // [16][HP] hidden tile
    float* w1s = hs + 16 * HP;                          // WLDS: [64][XP] image of W1, then [64][HP] image of W2
    float* w2s = w1s + BD * XP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int slabs = g.L / 16;
    int b, slab;
    if (!slab_of(blockIdx.x, g.batch, slabs, b, slab)) return;
    const int n0 = slab * 16;
    const long m0 = ((long)b * slabs + slab) * 16;      // first row of the slab in the (batch*L) row space
    const int klen = g.L / BW;                          // every wave: its eighth of the keys, all heads

    // the slab's own X rows (first 64 columns of the concat tile): requested first, parked after the attention loads
    float4 xown = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 256) xown = *reinterpret_cast<const float4*>(g.xcat + (m0 + (tid >> 4)) * W + 4 * (tid & 15));
    // 1 / rowsum of the row this thread normalises after the reduction (item = tid when H = 2: one per thread): requested here -
    // round 5: loaded inside the reduction it was a dependent memory round trip in the middle of the chain
    const float rinv0 = g.inv[(long)((tid >> 8) % H) * g.L + n0 + 4 * ((tid & 63) >> 4) + ((tid >> 6) & 3)];

    BSTAMP(0);
    // MLP operands (waves 0..3 own the four hidden / output tiles): requested before (EARLYW) or after the attention
    // contraction, consumed after the reduction
    const bool mlp_wave = wave < 4;
    const int c1 = (wave & 3) * 16 + l15;
    float4 bv[KS], w2v[4];
    float bias = 0.0f, bias2 = 0.0f;
    auto load_weights = [&]() {
        if (!mlp_wave) return;
        if (!WLDS) {
#pragma unroll
            for (int s = 0; s < KS; ++s) bv[s] = *reinterpret_cast<const float4*>(g.w1 + (long)c1 * W + 16 * s + 4 * kq);
#pragma unroll
            for (int s = 0; s < 4; ++s) w2v[s] = *reinterpret_cast<const float4*>(g.w2 + (long)c1 * BD + 16 * s + 4 * kq);
        }
        bias = g.b1[c1];
        bias2 = g.b2[c1];
    };
    // WLDS: every thread's unit-stride pieces of W1 | W2, requested in front of the contraction's operands (the copy must be in LDS
    // one barrier before GEMM1); plain locals, not lambda captures (those went to scratch)
    f32x4_t wq1[WLDS ? NW1 : 1], wq2[WLDS ? NW2 : 1];
    if (WLDS) {
#pragma unroll
        for (int p = 0; p < NW1; ++p) wq1[p] = reinterpret_cast<const f32x4_t*>(g.w1)[p * 512 + tid];
#pragma unroll
        for (int p = 0; p < NW2; ++p) wq2[p] = reinterpret_cast<const f32x4_t*>(g.w2)[p * 512 + tid];
        load_weights();
    }
    f32x4_t acc[H][4];
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[hh][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // ELDS: the slab's E rows (A operand) through LDS as well, when they fit the region the weights take later (L <= 256: one
    // 32-key trip per wave).  As fragments, lane = row: 16 cache lines per wave instruction (2 048 look-ups per workgroup); as
    // unit-stride copies 256.  The value rows (B operand: lanes along the row) are requested first, in the same round trip.
    constexpr int EP = 256 + 4;
    const bool elds = WLDS && g.L == 256 && !(PIT_BLOCK_EXP & 64);
    if (elds) {
        float* es = w1s;                                // [H][16][EP], overwritten by the weights after the next barrier but one
        const float* vp = g.xcat + (long)b * g.L * W + 4 * l15;
        const int j0 = wave * 32;
        f32x4_t eq[H * 2];
#pragma unroll
        for (int p = 0; p < H * 2; ++p) {               // piece q: head q >> 10, row (q >> 6) & 15, keys 4 (q & 63) ..
            const int q = p * 512 + tid;
            eq[p] = *reinterpret_cast<const f32x4_t*>(g.e + (long)(q >> 10) * g.L * g.L + (long)(n0 + ((q >> 6) & 15)) * g.L + 4 * (q & 63));
        }
        f32x4_t bvv[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m) bvv[s][m] = *reinterpret_cast<const f32x4_t*>(vp + (long)(j0 + 16 * s + 4 * kq + m) * W);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < H * 2; ++p) {
            const int q = p * 512 + tid;
            *reinterpret_cast<f32x4_t*>(es + ((q >> 10) * 16 + ((q >> 6) & 15)) * EP + 4 * (q & 63)) = eq[p];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4_t av[H];
#pragma unroll
            for (int hh = 0; hh < H; ++hh) av[hh] = *reinterpret_cast<const f32x4_t*>(es + (hh * 16 + l15) * EP + j0 + 16 * s + 4 * kq);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    acc[hh][0] = mfma_16x16x4(av[hh][m], bvv[s][m][0], acc[hh][0]);
                    acc[hh][1] = mfma_16x16x4(av[hh][m], bvv[s][m][1], acc[hh][1]);
                    acc[hh][2] = mfma_16x16x4(av[hh][m], bvv[s][m][2], acc[hh][2]);
                    acc[hh][3] = mfma_16x16x4(av[hh][m], bvv[s][m][3], acc[hh][3]);
                }
        }
    } else {
        slab_contract<H, true>(g.e + (long)n0 * g.L, (long)g.L * g.L, g.L, g.xcat + (long)b * g.L * W, W, 0, nullptr,
                               wave * klen, (wave + 1) * klen, l15, kq, acc);
    }
    BSTAMP(1);
    if (!WLDS) load_weights();
#pragma unroll
    for (int hh = 0; hh < H; ++hh) park(pk, wave * H + hh, lane, acc[hh]);
    if (tid < 256) *reinterpret_cast<float4*>(xs + (tid >> 4) * XP + 4 * (tid & 15)) = xown;
    __syncthreads();
    BSTAMP(2);
    // reduce over the key splits, normalise, -> concat tile (LDS) and concat buffer (memory: the backward needs it)
    for (int item = tid; item < H * 256; item += 512) {
        const int hh = item >> 8, i = (item >> 6) & 3, ln = item & 63;
        const int r = 4 * (ln >> 4) + i, col = 4 * (ln & 15);
        float4 s = parked_sum(pk, hh, BW, H, i, ln);
        const float rinv = item == tid ? rinv0 : g.inv[(long)hh * g.L + n0 + r];
        s.x *= rinv; s.y *= rinv; s.z *= rinv; s.w *= rinv;
        *reinterpret_cast<float4*>(xs + r * XP + BD + hh * BD + col) = s;
        *reinterpret_cast<float4*>(g.xcat + (m0 + r) * W + BD + hh * BD + col) = s;
    }
    if (WLDS) {
#pragma unroll
        for (int p = 0; p < NW1; ++p) {
            const int q = p * 512 + tid, row = q / (W / 4), c4 = q % (W / 4);
            *reinterpret_cast<f32x4_t*>(w1s + row * XP + 4 * c4) = wq1[p];
        }
#pragma unroll
        for (int p = 0; p < NW2; ++p) {
            const int q = p * 512 + tid, row = q / (BD / 4), c4 = q % (BD / 4);
            *reinterpret_cast<f32x4_t*>(w2s + row * HP + 4 * c4) = wq2[p];
        }
    }
    BSTAMP(3);
    __syncthreads();
    BSTAMP(4);
    // ---- the block's MLP on the 16 x W tile (the phases of mlp_fwd16_kernel, A operand from LDS)
    if (mlp_wave) {
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if ((PIT_BLOCK_EXP & 2) && (s & 1)) { if (!WLDS) asm volatile("" :: "v"(bv[s].x), "v"(bv[s].y), "v"(bv[s].z), "v"(bv[s].w)); continue; }
            const float4 a = *reinterpret_cast<const float4*>(xs + l15 * XP + 16 * s + 4 * kq);
            const float4 bw = WLDS ? *reinterpret_cast<const float4*>(w1s + c1 * XP + 16 * s + 4 * kq) : bv[s];
            a0 = mfma_16x16x4(a.x, bw.x, a0);
            a1 = mfma_16x16x4(a.y, bw.y, a1);
            a0 = mfma_16x16x4(a.z, bw.z, a0);
            a1 = mfma_16x16x4(a.w, bw.w, a1);
        }
        BSTAMP(5);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            const float z = a0[i] + a1[i] + bias;
            const float hv = gelu_erf(z);
            hs[r * HP + c1] = hv;
            g.z1[(m0 + r) * BD + c1] = z;
            g.h[(m0 + r) * BD + c1] = hv;
        }
    }
    BSTAMP(6);
    __syncthreads();
    BSTAMP(7);
    if (!mlp_wave) return;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(hs + l15 * HP + 16 * s + 4 * kq);
        const float4 bw = WLDS ? *reinterpret_cast<const float4*>(w2s + c1 * HP + 16 * s + 4 * kq) : w2v[s];
        o0 = mfma_16x16x4(a.x, bw.x, o0);
        o1 = mfma_16x16x4(a.y, bw.y, o1);
        o0 = mfma_16x16x4(a.z, bw.z, o0);
        o1 = mfma_16x16x4(a.w, bw.w, o1);
    }
    BSTAMP(8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long r = m0 + 4 * kq + i;
        float v = o0[i] + o1[i] + bias2;
        if (g.out_gelu) { g.z2[r * BD + c1] = v; v = gelu_erf(v); }
        g.y[r * g.ldy + c1] = v;
    }
    BSTAMP(9);
}

// ---------------------------------------------------------------------------------------------- backward
struct BlockBwdArgs {
    const float *e, *inv, *qw;          // this layer: (H,L,L), (H,L), (H,L,L)
    int L, batch;
    const float* d_xcat;                // (batch*L, (1+H)*64): gradient of this block's concat tensor (complete)
    const float* xcat;                  // this block's concat tensor (columns [0,64) = the attention's values)
    double* dscale;                     // this layer's accumulators (n_head * PIT_DSCALE_SLOTS), or null: no d(scale) part
    // previous block's MLP (null w1 = none: the slab's d(values) is written to d_values)
    const float *w1, *w2, *z1, *z2; int out_gelu, n0p;      // n0p = input width of that MLP
    float *d_xprev; long ld_dxprev;     // (rows, n0p)
    float *dz1, *dz2;                   // scratch of that MLP's backward: (rows, 64) each
    float* d_values; long ld_dvalues;   // (rows, 64) when there is no previous MLP
    int n_chain, n_ds;                  // workgroup ranges: [0, n_chain) chain slabs, then n_ds d(scale) slabs, then the rider
};

template <int H>
__device__ __forceinline__ void block_bwd_chain(const BlockBwdArgs& g, float* smem, int slab) {
    constexpr int W = (1 + H) * BD;
    constexpr int P1 = BD + 4;
    float* pk = smem;
    float* ds2 = smem + PARK_FLOATS;                    // [16][P1] dZ2 tile (= dY of the previous MLP after its gelu')
    float* ds1 = ds2 + 16 * P1;                         // [16][P1] dZ1 tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int slabs = g.L / 16;
    int b, sl;
    if (!slab_of(slab, g.batch, slabs, b, sl)) return;
    const int j0 = sl * 16;
    const long m0 = ((long)b * slabs + sl) * 16;
    const int klen = g.L / BW;

    // residual d_out[b, j, 0:64] of torch.cat((inputs, conv), -1) (pit.py:44) and the gelu' argument: requested first.
    // Every thread owns two adjacent elements of the slab's 16 x 64 d(values) tile.
    const int oi = (tid >> 6) & 3, oln = tid & 63, oh = tid >> 8;             // register i, parking lane, column pair
    const int orow = 4 * (oln >> 4) + oi, ocol = 4 * (oln & 15) + 2 * oh;
    const bool has_mlp = g.w1 != nullptr;
    const float2 res = *reinterpret_cast<const float2*>(g.d_xcat + (m0 + orow) * W + ocol);
    float2 z2v = make_float2(0.f, 0.f);
    if (has_mlp && g.out_gelu) z2v = *reinterpret_cast<const float2*>(g.z2 + (m0 + orow) * BD + ocol);
    const int c1 = (wave & 3) * 16 + l15;
    BSTAMP(10);
    // d(values)[j] = sum_h sum_n E_h[j][n] * (inv_h[n] * dO_h[n]):  E is symmetric, row j of E is column j
    f32x4_t acc[1][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[0][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    slab_contract<H, false>(g.e + (long)j0 * g.L, (long)g.L * g.L, g.L, g.d_xcat + (long)b * g.L * W + BD, W, BD, g.inv,
                            wave * klen, (wave + 1) * klen, l15, kq, acc);
    BSTAMP(11);
    // operands of the MLP phases (phase B: waves 0..3 own the four dZ1 tiles; phase C: dX tiles wave, wave + 8): requested
    // once the contraction's operand registers are free (the kernel must stay within 128 VGPRs: two workgroups per CU, the
    // d(scale) and rider workgroups run beside the chain), their latency hides behind the reduction
    float w2v[4][4], z1v[4], w1v[2][4][4];
    if (has_mlp) {
        if (wave < 4) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) w2v[s][ee] = g.w2[(long)(16 * s + 4 * kq + ee) * BD + c1];   // B(k,n) = w2[k][n]
#pragma unroll
            for (int i = 0; i < 4; ++i) z1v[i] = g.z1[(m0 + 4 * kq + i) * BD + c1];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = (wave + t * BW) * 16 + l15;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee)
                    w1v[t][s][ee] = (col < g.n0p) ? g.w1[(long)(16 * s + 4 * kq + ee) * g.n0p + col] : 0.0f;
        }
    }
    park(pk, wave, lane, acc[0]);
    __syncthreads();
    BSTAMP(12);
    {
        float2 s = res;
#pragma unroll
        for (int w = 0; w < BW; ++w) {
            s.x += pk[((w * 4 + 2 * oh) * 4 + oi) * 64 + oln];
            s.y += pk[((w * 4 + 2 * oh + 1) * 4 + oi) * 64 + oln];
        }
        if (!has_mlp) {
            *reinterpret_cast<float2*>(g.d_values + (m0 + orow) * g.ld_dvalues + ocol) = s;
        } else {
            if (g.out_gelu) { s.x *= gelu_erf_grad(z2v.x); s.y *= gelu_erf_grad(z2v.y); }
            *reinterpret_cast<float2*>(g.dz2 + (m0 + orow) * BD + ocol) = s;
            *reinterpret_cast<float2*>(ds2 + orow * P1 + ocol) = s;
        }
    }
    if (!has_mlp) return;
    BSTAMP(13);
    __syncthreads();
    BSTAMP(14);
    // ---- phase B: dZ1 = (dZ2 W2) * gelu'(Z1)
    if (wave < 4) {
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(ds2 + l15 * P1 + 16 * s + 4 * kq);
            a0 = mfma_16x16x4(a.x, w2v[s][0], a0);
            a1 = mfma_16x16x4(a.y, w2v[s][1], a1);
            a0 = mfma_16x16x4(a.z, w2v[s][2], a0);
            a1 = mfma_16x16x4(a.w, w2v[s][3], a1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            const float v = (a0[i] + a1[i]) * gelu_erf_grad(z1v[i]);
            ds1[r * P1 + c1] = v;
            g.dz1[(m0 + r) * BD + c1] = v;
        }
    }
    BSTAMP(15);
    __syncthreads();
    BSTAMP(16);
    if (!g.d_xprev) return;
    // ---- phase C: dX tiles = dZ1 W1
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tile = wave + t * BW;
        if (tile * 16 >= g.n0p) break;
        if ((PIT_BLOCK_EXP & 8) && t == 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(w1v[t][s][0]), "v"(w1v[t][s][1]), "v"(w1v[t][s][2]), "v"(w1v[t][s][3]));
            break;
        }
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(ds1 + l15 * P1 + 16 * s + 4 * kq);
            o0 = mfma_16x16x4(a.x, w1v[t][s][0], o0);
            o1 = mfma_16x16x4(a.y, w1v[t][s][1], o1);
            o0 = mfma_16x16x4(a.z, w1v[t][s][2], o0);
            o1 = mfma_16x16x4(a.w, w1v[t][s][3], o1);
        }
        const int col = tile * 16 + l15;
        if (col < g.n0p) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g.d_xprev[(m0 + 4 * kq + i) * g.ld_dxprev + col] = o0[i] + o1[i];
        }
    }
    BSTAMP(17);
}

// d c_h -= sum_{n, d} dO_h[n, d] * sum_j Q_h[n, j] U[j, d]   (SURVEY appendix B rearranged; Q carries the centring and 1/rowsum)
template <int H>
__device__ __forceinline__ void block_bwd_dscale(const BlockBwdArgs& g, float* smem, int slab) {
    constexpr int W = (1 + H) * BD;
    float* pk = smem;                                   // [H * PARK_FLOATS]
    double* wred = reinterpret_cast<double*>(smem + H * PARK_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int slabs = g.L / 16;
    int b, sl;
    if (!slab_of(slab, g.batch, slabs, b, sl)) return;
    const int n0 = sl * 16;
    const long m0 = ((long)b * slabs + sl) * 16;
    const int klen = g.L / BW;
    // this thread's d_out element group (head hh, row, 4 columns): requested before the contraction
    const bool own = tid < H * 256;
    const int hh = tid >> 8, i = (tid >> 6) & 3, ln = tid & 63;
    const int r = 4 * (ln >> 4) + i, col = 4 * (ln & 15);
    float4 dov = make_float4(0.f, 0.f, 0.f, 0.f);
    if (own) dov = *reinterpret_cast<const float4*>(g.d_xcat + (m0 + r) * W + BD + hh * BD + col);
    f32x4_t acc[H][4];
#pragma unroll
    for (int h2 = 0; h2 < H; ++h2)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[h2][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    slab_contract<H, true>(g.qw + (long)n0 * g.L, (long)g.L * g.L, g.L, g.xcat + (long)b * g.L * W, W, 0, nullptr,
                           wave * klen, (wave + 1) * klen, l15, kq, acc);
#pragma unroll
    for (int h2 = 0; h2 < H; ++h2) park(pk, wave * H + h2, lane, acc[h2]);
    __syncthreads();
    double part = 0.0;
    if (own) {
        const float4 s = parked_sum(pk, hh, BW, H, i, ln);
        part = (double)s.x * (double)dov.x + (double)s.y * (double)dov.y + (double)s.z * (double)dov.z + (double)s.w * (double)dov.w;
    }
    part = wave_sum_d(part);
    if (lane == 0) wred[wave] = part;
    __syncthreads();
    if (tid < H) {                                      // waves [4h, 4h + 4) hold head h
        double tot = 0.0;
        for (int w = 4 * tid; w < 4 * tid + 4; ++w) tot += wred[w];
        atomicAdd(g.dscale + (long)tid * PIT_DSCALE_SLOTS + ((int)(m0 >> 4) & (PIT_DSCALE_SLOTS - 1)), -tot);
    }
}

// NDW riders (0, 1 or 2 pairs of weight-gradient reductions): the block's own MLP, and a slice of a LARGER postponed job
// (the decoder MLP's: 14 792 rows at Darcy b=8 - inside the decoder attention's own launch it cost more than a launch of
// its own, spread over the block launches it runs on compute units the chain leaves idle)
template <int H, int NDW>
__global__ __launch_bounds__(512) void block_bwd_kernel(BlockBwdArgs g, pit_detail::DwPair w, pit_detail::DwPair w2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int id = blockIdx.x;
    BREC(0);
    // (placement, tools/stamp_block.py: the dispatcher deals workgroup ids to the CUs of an XCD in turn and starts over after one
    //  each, so the riders - dealt third - share the compute units of the FIRST range.  d(scale) slabs first, i.e. riders beside
    //  the 5 us d(scale) workgroups instead of the 7 us chain workgroups, was measured: 11.4 against 11.1 us in the step - a rider
    //  beside a d(scale) slab lives 8 us, 4.8 alone)
    if (id < g.n_chain) { block_bwd_chain<H>(g, smem, id); BREC(1); return; }
    id -= g.n_chain;
    if (id < g.n_ds) { block_bwd_dscale<H>(g, smem, id); BREC(1); return; }
    id -= g.n_ds;
    if (PIT_BLOCK_EXP & 32) return;                     // (experiment: riders leave at once - the launch without their work)
    if (NDW >= 1 && id < w.n1 + w.n2) { dw_pair_body(w, id, smem); BREC(1); return; }
    if (NDW >= 2) dw_pair_body(w2, id - (w.n1 + w.n2), smem);
    BREC(1);
}

constexpr size_t fwd_smem(int H, bool wlds) {
    return ((size_t)H * PARK_FLOATS + 16 * ((1 + H) * BD + 4) + 16 * (BD + 4) + (wlds ? BD * ((1 + H) * BD + 4) + BD * (BD + 4) : 0)) * sizeof(float);
}
constexpr size_t bwd_smem(int H, bool dscale) {
    return std::max((size_t)(PARK_FLOATS + 2 * 16 * (BD + 4)) * sizeof(float),
                    dscale ? ((size_t)H * PARK_FLOATS) * sizeof(float) + BW * sizeof(double) : (size_t)0);
}

}  // namespace

// 1 when the fused processor-block path covers this shape (include/pit_hip.h)
extern "C" int pit_block_supported(int n_pts, int n_head, int dim, int batch) {
    if (dim != BD || (n_head != 1 && n_head != 2)) return 0;
    if (n_pts <= 0 || batch <= 0 || n_pts % (32 * BW) != 0) return 0;                  // 32-key trips, keys split over 8 waves
    // the path materialises E and Q: n_layers * n_head * n_pts^2 floats EACH, kept for the backward (the per-layer kernels keep
    // nothing of that size) - 2048 points = 32 MiB per layer and head is where it stops paying anyway
    if (n_pts > 2048) return 0;
    const long rows = (long)batch * n_pts;
    return rows >= 256 && rows <= 16384;                // the latency regime; above, the tiled kernels of pit_posatt.hip
}

extern "C" int pit_block_weights(const float* mesh, int n_pts, int space_dim, int metric, float period, int n_layers,
                                 const float* const* heads, int head_is_scale, int n_head, float* e, float* q,
                                 float* inv, float* rowstat, float* scale_out, void* stream) {
    WeightsArgs a;
    if (int rc = fill_weights_args(a, mesh, n_pts, space_dim, metric, period, n_layers, heads, head_is_scale, n_head, e, q, inv,
                                   rowstat, scale_out)) return rc;
    const long rows = (long)n_layers * n_head * n_pts;
    hipLaunchKernelGGL(block_weights_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_block_fwd(const float* e, const float* inv, int n_pts, int n_head, int dim, int batch, float* xcat,
                             const float* w1, const float* b1, const float* w2, const float* b2, int out_gelu,
                             float* z1, float* h, float* z2, float* y, long ldy, int math_mode, void* stream) {
    if (!e || !inv || !xcat || !w1 || !b1 || !w2 || !b2 || !z1 || !h || !y || (out_gelu && !z2)) return PIT_ERR_NULL;
    if (math_mode != PIT_MATH_FP32 || !pit_block_supported(n_pts, n_head, dim, batch)) return PIT_ERR_UNSUPPORTED;
    if (ldy < dim || !aligned16(e) || !aligned16(xcat) || !aligned16(w1) || !aligned16(w2)) return PIT_ERR_SIZE;
    BlockFwdArgs g;
    g.e = e; g.inv = inv; g.L = n_pts; g.batch = batch; g.xcat = xcat;
    g.w1 = w1; g.b1 = b1; g.w2 = w2; g.b2 = b2; g.out_gelu = out_gelu;
    g.z1 = z1; g.h = h; g.z2 = z2; g.y = y; g.ldy = ldy;
    const dim3 grid((unsigned)slab_grid(batch, n_pts / 16)), block(64 * BW);
    static const bool no_wlds = getenv("PIT_NO_BLOCK_WLDS") != nullptr;                  // (A/B switch, read once)
    static const int n_cus = [] {                       // (one device per process: read once)
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        return cus;
    }();
    const bool earlyw = !no_wlds && (int)grid.x <= n_cus;      // WLDS: the weights through LDS (one workgroup per CU: 64 KB more LDS)
    const size_t FWD_SMEM = fwd_smem(n_head, earlyw);
#define PIT_BLOCK_FWD(H_, E_)                                                                                              \
    do {                                                                                                                   \
        static bool once = ((void)hipFuncSetAttribute((const void*)block_fwd_kernel<H_, E_>,                              \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);            \
        (void)once;                                                                                                        \
        hipLaunchKernelGGL((block_fwd_kernel<H_, E_>), grid, block, FWD_SMEM, (hipStream_t)stream, g);                     \
    } while (0)
    if (n_head == 1) { if (earlyw) PIT_BLOCK_FWD(1, true); else PIT_BLOCK_FWD(1, false); }
    else { if (earlyw) PIT_BLOCK_FWD(2, true); else PIT_BLOCK_FWD(2, false); }
#undef PIT_BLOCK_FWD
    PIT_CHECK_LAUNCH();
    return 0;
}

#ifdef PIT_WGREC
extern "C" int pit_block_read_wgrec(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pit_block_wgrec), (size_t)n * 4 * sizeof(unsigned long long));
}
#endif
#ifdef PIT_STAMPS_DBG
extern "C" int pit_block_set_dbg(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(pit_block_dbg), &v, sizeof(int)); }
#endif
#ifdef PIT_STAMPS
extern "C" int pit_block_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pit_block_stamps), sizeof(unsigned long long) * 32);
}
#endif

extern "C" int pit_block_bwd(const float* e, const float* inv, const float* qw, int n_pts, int n_head, int dim, int batch,
                             const float* d_xcat, const float* xcat, double* dscale,
                             const float* w1, const float* w2, const float* z1, const float* z2, int out_gelu, int n0_prev,
                             float* d_xprev, long ld_dxprev, float* scratch_prev,
                             float* d_values, long ld_dvalues,
                             const pit_mlp_params_job* rider, const pit_mlp_params_job* rider2, int math_mode, void* stream) {
    if (!e || !inv || !d_xcat) return PIT_ERR_NULL;
    if (dscale && (!qw || !xcat)) return PIT_ERR_NULL;
    if (math_mode != PIT_MATH_FP32 || !pit_block_supported(n_pts, n_head, dim, batch)) return PIT_ERR_UNSUPPORTED;
    const bool has_mlp = w1 != nullptr;
    if (has_mlp) {
        if (!w2 || !z1 || !scratch_prev || (out_gelu && !z2)) return PIT_ERR_NULL;
        if (n0_prev <= 0 || n0_prev > 16 * 16 || (d_xprev && ld_dxprev < n0_prev)) return PIT_ERR_SIZE;   // <= 2 dX tiles per wave
    } else if (!d_values || ld_dvalues < dim || ld_dvalues % 4 != 0 || !aligned16(d_values)) {
        return d_values ? PIT_ERR_SIZE : PIT_ERR_NULL;
    }
    if (!aligned16(e) || !aligned16(d_xcat) || (qw && !aligned16(qw)) || (xcat && !aligned16(xcat)) ||
        (has_mlp && (!aligned16(scratch_prev) || (z2 && !aligned16(z2))))) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)batch * n_pts;
    BlockBwdArgs g;
    g.e = e; g.inv = inv; g.qw = qw; g.L = n_pts; g.batch = batch; g.d_xcat = d_xcat; g.xcat = xcat; g.dscale = dscale;
    g.w1 = w1; g.w2 = w2; g.z1 = z1; g.z2 = z2; g.out_gelu = out_gelu; g.n0p = n0_prev;
    g.d_xprev = d_xprev; g.ld_dxprev = ld_dxprev;
    g.dz1 = scratch_prev; g.dz2 = has_mlp ? scratch_prev + rows * dim : nullptr;      // the layout pit_mlp_bwd_data uses
    g.d_values = d_values; g.ld_dvalues = ld_dvalues;
    g.n_chain = slab_grid(batch, n_pts / 16);
    g.n_ds = dscale ? g.n_chain : 0;
    pit_detail::DwPair dw = pit_detail::DwPair(), dw2 = pit_detail::DwPair();
    const pit_mlp_params_job* jobs[2] = {rider, rider2};
    pit_detail::DwPair* plans[2] = {&dw, &dw2};
    int ndw = 0;                                        // carried riders are packed into dw, dw2 in order
    bool carried[2] = {false, false};
    for (int r = 0; r < 2; ++r)
        if (jobs[r] && pit_detail::plan_dw_pair(*jobs[r], BW, plans[ndw], 768, true)) { carried[r] = true; ++ndw; }
    const int n_dw = (ndw >= 1 ? dw.n1 + dw.n2 : 0) + (ndw >= 2 ? dw2.n1 + dw2.n2 : 0);
    const bool any_rr = (ndw >= 1 && (dw.rr1 || dw.rr2)) || (ndw >= 2 && (dw2.rr1 || dw2.rr2));
    const size_t sm = std::max(std::max(bwd_smem(n_head, dscale != nullptr), ndw ? (size_t)BW * 16 * 64 * sizeof(float) : (size_t)0),
                               any_rr ? (size_t)4 * pit_detail::RR_BK * 64 * sizeof(float) : (size_t)0);
    const dim3 grid((unsigned)(g.n_chain + g.n_ds + n_dw)), block(64 * BW);
#define PIT_BLOCK_BWD(H_, DW_)                                                                                             \
    do {                                                                                                                   \
        static bool once = ((void)hipFuncSetAttribute((const void*)block_bwd_kernel<H_, DW_>,                             \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);                 \
        (void)once;                                                                                                        \
        hipLaunchKernelGGL((block_bwd_kernel<H_, DW_>), grid, block, sm, s, g, dw, dw2);                                   \
    } while (0)
#define PIT_BLOCK_BWD_H(H_) do { if (ndw == 2) PIT_BLOCK_BWD(H_, 2); else if (ndw == 1) PIT_BLOCK_BWD(H_, 1); else PIT_BLOCK_BWD(H_, 0); } while (0)
    if (n_head == 1) PIT_BLOCK_BWD_H(1); else PIT_BLOCK_BWD_H(2);
#undef PIT_BLOCK_BWD_H
#undef PIT_BLOCK_BWD
    PIT_CHECK_LAUNCH();
    for (int r = 0; r < 2; ++r)
        if (jobs[r] && !carried[r]) {                   // too large to ride: the launches pit_mlp_bwd_params would have made
            const pit_mlp_params_job* j = jobs[r];
            const int rc = pit_mlp_bwd_params(j->x, j->ldx, j->rows, j->n0, j->n1, j->n2, j->h, j->out_gelu, j->d_y, j->ld_dy,
                                              j->d_w1, j->d_b1, j->d_w2, j->d_b2, j->accumulate, j->scratch, j->math_mode, stream);
            if (rc) return rc;
        }
    return 0;
}
