// Fused encoder-side and decoder-side launches of the small (latency-bound) regime on batch-free meshes.
//
// Around the fused processor (pit_block.hip) a Darcy-sized step still ran eleven launches: the down-projection (pit.py:109,
// posatt_cross_fixed on candidate lists), the encoder MLP + gelu (pit.py:110-111), the up-projection (pit.py:125), the
// decoder MLP (pit.py:126), the loss (utils.py:86-98) and their backward counterparts - each 5-17 us of dependent round trips at
// <= 9 % MFMA busy, with the (batch, n_out, H*hid) up-projection output and its gradient making a full trip through memory between
// two launches in both directions.  Here each side is ONE launch per direction, a workgroup owning 16 consecutive mesh rows of one
// sample from the attention to the end of the MLP:
//
//   slab_plan_kernel     once per (mesh_out, mesh_in) pair - the meshes are batch-free and fixed, so everything about a 16-row
//                        slab that does not depend on lmda is static: the candidates' squared distances (sq_dist3, the same
//                        function every other kernel uses), the sorted UNION of the slab's candidate keys and each candidate's
//                        slot in it.  No coordinate is read on the step's path any more.
//   decoder_fwd_kernel   up-projection as a union-tile contraction (P: 16 x U weights scattered into LDS, the U <= 64 value rows
//                        of the union fetched ONCE per slab, v_mfma_f32_16x16x4_f32) -> the 16 x H*hid tile stays in LDS as the A
//                        operand of the decoder MLP (GEMM1 + bias + gelu, thin out_dim <= 4 output as row dots) -> optionally the
//                        slab's partial sums of the RelLp loss (fp64, one slot per slab: deterministic, nothing to zero).
//   decoder_bwd_kernel   d(pred) (given, or formed from the loss's partial sums) -> dZ1 -> dX = dZ1 W1 (16 x H*hid, LDS only: the
//                        7.6 MB gradient of the up-projection's output never exists) -> d(scale) as (Q U) . dX and d(values) as
//                        P^T dX on MFMA against the same union tile, added to memory with one pass of fp32 atomics per slab.
//   encoder_fwd_kernel   down-projection (a wave per row, lane = candidate: 40 keys x 3 channels is no contraction worth an MFMA)
//                        -> encoder MLP + gelu on the slab -> the processor's first concat buffer; the processor's block weights
//                        (block_weights_body) ride as extra workgroups, the step's gradient accumulators are cleared on the way.
//   encoder_bwd_kernel   encoder MLP backward (data path) -> dX (16 x H*(s+in_dim), LDS only) -> the down-projection's d(scale).
//
// Exactness: mask decisions use the same fp32 expression tree as every other kernel (S = fl(c m), T = lerp(fl(c m_k), fl(c m_k1), w),
// keep = S <= T) on distances computed by the same sq_dist3; tests/test_gpu_round5.py compares each launch with the oracle.
#include "pit_common.h"
#include "pit_gemm_rd.h"
#include "pit_block_dev.h"

#ifndef PIT_EDGE_MAX_ROWS
#define PIT_EDGE_MAX_ROWS (1 << 20) // rows (batch x mesh points) up to which a layer takes the fused launch: measured faster than the per-layer
                                   // kernels at every Darcy batch from 8 to 256 (473 k decoder rows); 32-bit buffer offsets hold to 8 M rows
#endif
#ifndef PIT_EDGE_DBG
#define PIT_EDGE_DBG 0      // diagnostic builds (tools/edge_variants.sh): bits switch parts of the launches off to time them
#endif

namespace {

#if PIT_EDGE_DBG & 0x100
// diagnostic build (tools/edge_variants.sh 256): REFCLK stamps (100 MHz) of wave 0 of one workgroup per kernel, never read by the kernels
__device__ unsigned long long pit_edge_stamps[4][16];
#define ESTAMP(k_, i_) do { if (blockIdx.x == 300 && threadIdx.x == 0) pit_edge_stamps[k_][i_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ESTAMP(k_, i_) do { } while (0)
#endif

constexpr int ER = 16;                  // rows per slab
constexpr int EU = PIT_SLAB_UNION_MAX;  // union keys a slab tile holds (64)

// Predicated loads WITHOUT a branch: hipcc turns `ok ? p[i] : 0` - and, by sinking the load to its only use, even
// `v = p[ok ? i : 0]; ok ? v : 0` - into an exec-masked block that ends in s_waitcnt vmcnt(0): every such load a serial memory
// round trip (eleven of them in the first version of decoder_fwd_kernel).  As in pit_common.h the load goes through a raw buffer
// descriptor whose range check returns 0: an unconditional instruction whose result is used unconditionally.  The descriptor
// spans 2 GiB from the (wave-uniform) base pointer - the tensors of this regime are a few MB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void* p) { return make_rsrc(p, 0x7ffffff0u); }
constexpr unsigned OOB = 0x7ffffff8u;
__device__ __forceinline__ float ldg_if(const float* p, long i, bool ok) {
    return buf_load(wide_rsrc(p), ok ? (unsigned)(i * 4) : OOB);
}
__device__ __forceinline__ int ldi_if(const int* p, long i, bool ok) {
    return __builtin_amdgcn_raw_buffer_load_b32(wide_rsrc(p), (int)(ok ? (unsigned)(i * 4) : OOB), 0, 0);
}
__device__ __forceinline__ float4 ldg4_if(const float* p, long i, bool ok) {
    float v[4];
    buf_load4(wide_rsrc(p), ok ? (unsigned)(i * 4) : OOB, v);
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float pow_abs_p(float x, int p) {
    const float ax = fabsf(x);
    return p == 1 ? ax : ax * ax;
}

// ------------------------------------------------------------------------------------------------ static slab plan
struct SlabBuildArgs {
    const float *mesh_out, *mesh_in;
    int n_out, n_in, sdim, used, periodic; float period;
    const int *idx, *cnt; int cap, umax;
    float* m; unsigned short* slot; int* keys; int* nkeys; int* report;      // report[0] = max union, [1] = 1 if a list overflowed, [2] = max count
    int er;                                                                  // rows per slab: 16 (the fused launches), 64 / 128 / 256 (pit_fold.hip)
};

__global__ __launch_bounds__(256) void slab_plan_kernel(SlabBuildArgs a) {
    __shared__ unsigned bm[512];          // n_in <= 16384 keys
    __shared__ int pref[513];
    const int tid = threadIdx.x, slab = blockIdx.x;
    const int nwords = (a.n_in + 31) >> 5;
    for (int w = tid; w < nwords; w += 256) bm[w] = 0u;
    __syncthreads();
    const int er = a.er;
    const int total = er * a.cap;
    for (int e = tid; e < total; e += 256) {
        const int r = e / a.cap, i = e - r * a.cap;
        const int row = slab * er + r;
        float mv = 0.0f;
        if (row < a.n_out) {
            const int c = a.cnt[row];
            if (c > a.cap && i == 0) atomicMax(a.report + 1, 1);
            if (i == 0) atomicMax(a.report + 2, min(c, a.cap));
            if (i < min(c, a.cap)) {
                const int j = a.idx[(long)row * a.cap + i];
                const float* xo = a.mesh_out + (long)row * a.sdim;
                const float* xi = a.mesh_in + (long)j * a.sdim;
                mv = sq_dist3(xo[0], a.used > 1 ? xo[1] : 0.0f, a.used > 2 ? xo[2] : 0.0f,
                              xi[0], a.used > 1 ? xi[1] : 0.0f, a.used > 2 ? xi[2] : 0.0f, a.periodic != 0, a.period);
                atomicOr(&bm[j >> 5], 1u << (j & 31));
            }
        }
        a.m[(long)(slab * er + r) * a.cap + i] = mv;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < nwords; ++w) { pref[w] = run; run += __popc(bm[w]); }
        pref[nwords] = run;
        a.nkeys[slab] = run;
        atomicMax(a.report, run);
    }
    __syncthreads();
    const int nk = pref[nwords];
    for (int w = tid; w < nwords; w += 256) {
        unsigned bits = bm[w];
        int s = pref[w];
        while (bits) {
            const int bpos = __ffs(bits) - 1;
            bits &= bits - 1u;
            if (s < a.umax) a.keys[(long)slab * a.umax + s] = w * 32 + bpos;
            ++s;
        }
    }
    for (int s = nk + tid; s < a.umax; s += 256) a.keys[(long)slab * a.umax + s] = 0;      // padding: a valid key, weight 0
    for (int e = tid; e < total; e += 256) {
        const int r = e / a.cap, i = e - r * a.cap;
        const int row = slab * er + r;
        int s = 0;
        if (row < a.n_out && i < min(a.cnt[row], a.cap)) {
            const int j = a.idx[(long)row * a.cap + i];
            s = pref[j >> 5] + __popc(bm[j >> 5] & ((1u << (j & 31)) - 1u));
        }
        a.slot[(long)(slab * er + r) * a.cap + i] = (unsigned short)min(s, 65535);
    }
}

// ------------------------------------------------------------------------------------------------ shared pieces
// The head scales of the launch: c_h as the caller gives it, or from lmda (pit_common.h).
template <int H>
__device__ __forceinline__ void head_scales(const float (&raw)[H], int is_scale, float (&c)[H]) {
#pragma unroll
    for (int h = 0; h < H; ++h) c[h] = is_scale ? raw[h] : head_scale_from_lmda(raw[h]);
}

// ------------------------------------------------------------------------------------------------ decoder weights
// The up-projection's softmax weights depend on (mesh pair, lmda) only - not on the sample, not on the activations: ONE workgroup
// per slab forms them once per step (as extra workgroups of the encoder-side launch, or a launch of its own), every
// (sample, slab) workgroup of the decoder launches then reads its 16 x U tile as an MFMA operand straight from memory.
//   pw[slab][h][row][slot] = P (normalised), qw[...] = Q = P (m - mbar) (the d(scale) weights; NULL: not needed), zeros elsewhere.
struct DecWArgs {
    pit_slab_plan p;
    const float* head; int head_is_scale, n_head, um, lpr;
    float *pw, *qw, *scale_out;
    unsigned short *pw16, *qw16;               // optional (pit_fold_weights, bf16 math mode): the same tiles rounded to bf16
    const float* w1; float* w1f; int dim;     // optional: the decoder MLP's W1 (dim, n_head*dim) copied in MFMA-fragment order
    int rb;                                   // rows per slab of the plan (16; pit_fold.hip: 64 / 128 / 256): a workgroup still forms 16 rows
};
__device__ __forceinline__ float seg_sum_rt(float v, int lpr) {
    v += dpp_f<0xB1, 0xf>(v);
    v += dpp_f<0x4E, 0xf>(v);
    v += dpp_f<0x124, 0xf>(v);
    v += dpp_f<0x128, 0xf>(v);
    if (lpr >= 32) v += __shfl_xor(v, 16, 64);
    if (lpr >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
// 256 threads; `tile`: 2 * 2 * 16 * PIT_SLAB_UNION_MAX floats of LDS.  `slab` counts 16-row groups: group c of the plan's slab
// blk = slab / (rb / 16) (rb = 16: the slab itself); head h's rows of it go to pw + ((blk * H + h) * rb + 16 c) * um.
__device__ __forceinline__ void dec_weights_body(const DecWArgs& g, int slab, float* tile) {
    const pit_slab_plan& p = g.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = g.n_head, um = g.um, lpr = g.lpr;
    const int per = H * ER * um;                            // floats of one tile (P; Q follows)
    for (int e = tid; e < 2 * per; e += 256) tile[e] = 0.0f;
    float c[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float raw = g.head[h < H ? h : 0];
        c[h] = g.head_is_scale ? raw : head_scale_from_lmda(raw);
    }
    __syncthreads();
    const int rpw = 64 / lpr;
    for (int r0 = 0; r0 < ER; r0 += 4 * rpw) {
        const int row_l = r0 + wave * rpw + lane / lpr, i = lane % lpr, n = slab * ER + row_l;
        const int craw = ldi_if(p.cnt, n, n < p.n_out);
        const long off = (long)(slab * ER + row_l) * p.cap + (i < p.cap ? i : 0);
        const float m = p.m[off];
        const int slot = p.slot[off];
        const bool valid = i < min(craw, p.cap);
        const int nn = n < p.n_out ? n : p.n_out - 1;
        const float mk = p.stats[nn], mk1 = p.stats[p.n_out + nn], mmin = p.stats[2 * (long)p.n_out + nn];
        for (int h = 0; h < H; ++h) {
            const float T = quantile_lerp(__fmul_rn(c[h], mk), __fmul_rn(c[h], mk1), p.rank_w);
            const float smin = __fmul_rn(c[h], mmin);
            const float sv = __fmul_rn(m, c[h]);
            const bool keep = valid && sv <= T;
            const float pv = keep ? __expf(smin - sv) : 0.0f;
            const float rs = seg_sum_rt(pv, lpr), qsum = seg_sum_rt(pv * m, lpr);
            const float inv = rs > 0.0f ? 1.0f / rs : 0.0f;
            if (valid && slot < um) {
                tile[(h * ER + row_l) * um + slot] = pv * inv;
                tile[per + (h * ER + row_l) * um + slot] = pv * inv * (m - qsum * inv);
            }
        }
    }
    __syncthreads();
    {
        const int gpb = g.rb / ER, blk = slab / gpb, cg = slab - blk * gpb, perh = ER * um;
        for (int e = 4 * tid; e < per; e += 4 * 256) {
            const int h = e / perh, r = e - h * perh;
            const long o = ((long)(blk * H + h) * g.rb + cg * ER) * um + r;
            const float4 pv4 = *reinterpret_cast<const float4*>(tile + e), qv4 = *reinterpret_cast<const float4*>(tile + per + e);
            *reinterpret_cast<float4*>(g.pw + o) = pv4;
            if (g.qw) *reinterpret_cast<float4*>(g.qw + o) = qv4;
            if (g.pw16) *reinterpret_cast<uint2*>(g.pw16 + o) = make_uint2((unsigned)f_to_bf16(pv4.x) | ((unsigned)f_to_bf16(pv4.y) << 16), (unsigned)f_to_bf16(pv4.z) | ((unsigned)f_to_bf16(pv4.w) << 16));
            if (g.qw16) *reinterpret_cast<uint2*>(g.qw16 + o) = make_uint2((unsigned)f_to_bf16(qv4.x) | ((unsigned)f_to_bf16(qv4.y) << 16), (unsigned)f_to_bf16(qv4.z) | ((unsigned)f_to_bf16(qv4.w) << 16));
        }
    }
    if (slab == 0 && tid < H && g.scale_out) g.scale_out[tid] = c[tid];
    // W1 of the decoder MLP in the order decoder_fwd_kernel's lanes consume it: 16-byte piece o = (wave * KS + s) * 64 + lane holds
    // w1[16 wave + (lane & 15)][16 s + 4 (lane >> 4) ..].  As fragments of the row-major matrix (lane = weight row) a wave instruction
    // touches 16 cache lines for 64 useful bytes each - 2 048 line look-ups per workgroup, in 928 workgroups; from this copy 256:
    // decoder_fwd 13.5 -> 11.5 us at Darcy b=8, 188 -> 170 us at b=256.  Once per step, here, because W1 changes once per step.
    if (g.w1f) {
        const int K0 = H * g.dim, KS = K0 / 16, n4 = g.dim * K0 / 4;
        for (int o = slab * 256 + tid; o < n4; o += p.n_slabs * (g.rb / ER) * 256) {
            const int ln = o & 63, s_ = (o >> 6) % KS, w = (o >> 6) / KS;
            reinterpret_cast<float4*>(g.w1f)[o] =
                *reinterpret_cast<const float4*>(g.w1 + (long)(16 * w + (ln & 15)) * K0 + 16 * s_ + 4 * (ln >> 4));
        }
    }
}
__global__ __launch_bounds__(256) void decoder_weights_kernel(DecWArgs g) {
    __shared__ __attribute__((aligned(16))) float tile[2 * 2 * ER * EU];
    dec_weights_body(g, blockIdx.x, tile);
}

// ------------------------------------------------------------------------------------------------ decoder forward
struct DecFwdArgs {
    pit_slab_plan p;
    const float* values; long ld_values, values_bstride; int batch;
    const float* pw;                          // (n_slabs, H, 16, um): the step's normalised weights (dec_weights_body)
    const float *w1, *b1, *w2, *b2; int n2;
    const float* w1f;                         // W1 in fragment order (dec_weights_body), or null: fragments of the row-major w1
    float *x, *z1, *h, *y;
    float* zero_buf; long zero_n;
    const float *tru, *lscale, *lshift; int loss_p; double* lpart;
    int um;                                   // slots of the union tiles in LDS: 32, 48 or 64 >= the plan's largest union
};

// The slab's union value rows: thread (r0 = tid / (D/4), q = tid % (D/4)) owns the 16-byte piece q of slots r0, r0 + 16, ... .
// The keys come straight from the plan (padded with key 0 - a valid row - beyond the union; the select zeroes those): two dependent
// round trips, no LDS hand-off in front of the second.  The upper half of the tile only for unions above 32 keys (wave-uniform).
template <int D>
__device__ __forceinline__ void union_keys(const pit_slab_plan& p, int slab, int tid, int nkup, int (&key)[EU / 16]) {
    const int r0 = tid / (D / 4);
    const int* kp = p.keys + (long)slab * p.umax + r0;
#pragma unroll
    for (int u = 0; u < EU / 16; ++u) key[u] = kp[16 * u];         // (all four: no branch on the union's size in front of them)
}
template <int D>
__device__ __forceinline__ void gather_union(const float* __restrict__ vb, long ldv, const int (&key)[EU / 16], int nk, int nkup, int tid,
                                             float4 (&uv)[EU / 16]) {
    const int r0 = tid / (D / 4), q = tid % (D / 4);
    // (slots beyond the union load out of range through the buffer descriptor: four unconditional loads, no branch)
#pragma unroll
    for (int u = 0; u < EU / 16; ++u)
        uv[u] = (PIT_EDGE_DBG & 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : ldg4_if(vb, (long)key[u] * ldv + 4 * q, r0 + 16 * u < nk);
}
template <int D>
__device__ __forceinline__ void park_union(float* ut, int nkup, int tid, const float4 (&uv)[EU / 16]) {
    const int r0 = tid / (D / 4), q = tid % (D / 4);
#pragma unroll
    for (int u = 0; u < EU / 16; ++u)
        if (16 * u < nkup) *reinterpret_cast<float4*>(ut + (r0 + 16 * u) * (D + 4) + 4 * q) = uv[u];
}

// A 16-row weight tile (H * 16 rows x um slots, contiguous in memory) goes through LDS: requested ONCE per workgroup as 16-byte
// pieces with the other loads, parked row-major with pitch um + 4, read by every wave as its MFMA A operand.  (Each wave loading
// its own fragments from memory was measured slower - decoder_fwd 13.1 -> 14.9 us: the L1 / address path, already carrying 32 KB
// of W1 per workgroup, is what bounds these launches.)
constexpr int WCP = (2 * ER * EU / 4 + 255) / 256;          // 16-byte pieces per thread at 256 threads, H = 2, um = 64
template <int NT>
__device__ __forceinline__ void wtile_request(const float* tile, int npc, int tid, float4 (&cp)[WCP * 256 / NT]) {
#pragma unroll
    for (int u = 0; u < WCP * 256 / NT; ++u) cp[u] = ldg4_if(tile, 4L * (tid + u * NT), tid + u * NT < npc);
}
template <int NT>
__device__ __forceinline__ void wtile_park(float* lds, int um, int npc, int tid, const float4 (&cp)[WCP * 256 / NT]) {
#pragma unroll
    for (int u = 0; u < WCP * 256 / NT; ++u) {
        const int e = tid + u * NT;
        if (e < npc) *reinterpret_cast<float4*>(lds + ((4 * e) / um) * (um + 4) + (4 * e) % um) = cp[u];
    }
}
// acc (16 rows x columns [16 wave, 16 wave + 16)) = W[16 x nkup] (row-major in LDS, pitch `up`) @ ut[nkup x D]
template <int D>
__device__ __forceinline__ f32x4_t tile_times_union(const float* wrow, int up, const float* ut, int nkup, int wave, int l15, int kq) {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nkup / 16; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(wrow + l15 * up + 16 * s + 4 * kq);
        const float* bp = ut + (16 * s + 4 * kq) * (D + 4) + 16 * wave + l15;
        a0 = mfma_16x16x4(a.x, bp[0], a0);
        a1 = mfma_16x16x4(a.y, bp[D + 4], a1);
        a0 = mfma_16x16x4(a.z, bp[2 * (D + 4)], a0);
        a1 = mfma_16x16x4(a.w, bp[3 * (D + 4)], a1);
    }
    f32x4_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = a0[i] + a1[i];
    return r;
}

template <int H, int D, bool LOSS>
__global__ __launch_bounds__(4 * D) void decoder_fwd_kernel(DecFwdArgs g) {
    constexpr int NT = 4 * D, NW = D / 16, K0 = H * D, KS = K0 / 16, XP = K0 + 4, HP = D + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int UP = g.um + 4;                   // pitch of the weight tile (the plan's largest union decides the LDS a launch takes)
    float* pt = smem;                          // [H][16][UP] the step's normalised weights of this slab
    float* ut = pt + H * ER * UP;              // [um][D + 4] union value rows
    float* xs = ut + g.um * (D + 4);           // [16][XP]
    float* hs = xs + ER * XP;                  // [16][HP]
    float* w2s = hs + ER * HP;                 // [4][D], then b2[4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const pit_slab_plan& p = g.p;
    // (ONE slab per workgroup.  A loop over 2-4 slabs with the MLP operands - 32 KB of W1, what loads the L1 / address path most -
    // fetched once was measured SLOWER: at Darcy b=256, 29 696 slabs, decoder_fwd 182 -> 202 us with four slabs per workgroup - they
    // run one after the other and wave 0's output phase stalls the rest; and the loop alone, with one slab, cost 3 us at b=8: the
    // loads of its body no longer overlap the operands requested in front of it.)
    int b, slab;
    if (!slab_of_xcd(blockIdx.x, g.batch, p.n_slabs, b, slab)) return;
    const int nk = min(p.nkeys[slab], EU), nkup = (nk + 15) & ~15;
    const long row0 = (long)b * p.n_out + slab * ER;             // first row of the slab in the (batch * n_out) row space
    const int c1 = wave * 16 + l15;
    ESTAMP(0, 0);
    // ---- ONE block of loads, nothing consumed inside it: the union keys first (the gather waits for them alone), then the
    // slab's weight tile, the MLP operands, the loss's operands
    int key[EU / 16];
    union_keys<D>(p, slab, tid, nkup, key);
    const int npc = H * ER * g.um / 4;
    float4 pcp[WCP * 256 / NT];
    wtile_request<NT>(g.pw + (long)slab * H * ER * g.um, npc, tid, pcp);
    float4 bv[KS];
    const float* w1base = g.w1f ? g.w1f : g.w1;          // (a select on the address, not a branch around the loads)
#pragma unroll
    for (int s = 0; s < KS; ++s)
        bv[s] = (PIT_EDGE_DBG & 1) ? make_float4(1.f, 2.f, 3.f, 4.f)
              : *reinterpret_cast<const float4*>(w1base + (g.w1f ? ((wave * KS + s) * 64 + lane) * 4 : c1 * K0 + 16 * s + 4 * kq));
    const float bias1 = g.b1[c1];
    const float w2r = ldg_if(g.w2, tid, tid < g.n2 * D);          // (n2 * D <= 4 * D = NT: one element per thread)
    const float b2r = ldg_if(g.b2, tid, tid < g.n2);
    // the loss's operands of this lane's row, output channel 0 (wave 0 finishes the slab; every wave requests them: no branch)
    float l_t0 = 0.0f, l_sc0 = 1.0f, l_sh0 = 0.0f;
    if (LOSS) {
        const int n = slab * ER + l15;
        const bool rv = n < p.n_out;
        l_t0 = ldg_if(g.tru, (row0 + l15) * g.n2, rv);
        const bool aff = rv && g.lscale != nullptr;              // (no affine map: out-of-range loads, the values are not used)
        l_sc0 = ldg_if(g.lscale, (long)n * g.n2, aff);
        l_sh0 = ldg_if(g.lshift, (long)n * g.n2, aff);
    }
    // ---- the union's value rows (second round trip), weights formed while they fly
    float4 uv[EU / 16];
    gather_union<D>(g.values + (long)b * g.values_bstride, g.ld_values, key, nk, nkup, tid, uv);
    __builtin_amdgcn_sched_barrier(0);       // (left to itself the scheduler sinks each load to its use: dependent round trips again)
    ESTAMP(0, 1);
    w2s[tid] = w2r;
    if (tid < 4) w2s[4 * D + tid] = b2r;
    wtile_park<NT>(pt, g.um, npc, tid, pcp);
    ESTAMP(0, 2);
    park_union<D>(ut, nkup, tid, uv);
    ESTAMP(0, 3);
    __syncthreads();
    ESTAMP(0, 4);
    // ---- attention output tile: X[:, h*D + 16 wave ..] = P_h @ U
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const f32x4_t o = tile_times_union<D>(pt + h * ER * UP, UP, ut, nkup, wave, l15, kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[(4 * kq + i) * XP + h * D + c1] = o[i];
    }
    ESTAMP(0, 5);
    __syncthreads();
    ESTAMP(0, 6);
    // the tile goes to memory as 16-byte pieces (the backward's weight-gradient reductions read it)
    if (g.x && !(PIT_EDGE_DBG & 4)) {
#pragma unroll
        for (int u = 0; u < (ER * K0 / 4 + NT - 1) / NT; ++u) {
            const int e = tid + u * NT, r = e / (K0 / 4), q = e % (K0 / 4);
            if (e < ER * K0 / 4 && slab * ER + r < p.n_out)
                *reinterpret_cast<float4*>(g.x + (row0 + r) * K0 + 4 * q) = *reinterpret_cast<const float4*>(xs + r * XP + 4 * q);
        }
    }
    // ---- decoder MLP, first layer: Z1 = X W1^T + b1, H = gelu(Z1)
    {
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(xs + l15 * XP + 16 * s + 4 * kq);
            a0 = mfma_16x16x4(a.x, bv[s].x, a0);
            a1 = mfma_16x16x4(a.y, bv[s].y, a1);
            a0 = mfma_16x16x4(a.z, bv[s].z, a0);
            a1 = mfma_16x16x4(a.w, bv[s].w, a1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            const float z = a0[i] + a1[i] + bias1;
            hs[r * HP + c1] = gelu_erf(z);
            if (g.z1 && slab * ER + r < p.n_out && !(PIT_EDGE_DBG & 4)) g.z1[(row0 + r) * D + c1] = z;
        }
    }
    ESTAMP(0, 7);
    __syncthreads();
    ESTAMP(0, 8);
    if (g.h && !(PIT_EDGE_DBG & (4 | 1024))) {
        const int r = tid / (D / 4), q = tid % (D / 4);             // NT = 16 * D / 4: one 16-byte piece per thread
        if (slab * ER + r < p.n_out)
            *reinterpret_cast<float4*>(g.h + (row0 + r) * D + 4 * q) = *reinterpret_cast<const float4*>(hs + r * HP + 4 * q);
    }
    if (wave != 0) {
        // the other waves clear this workgroup's share of the buffer the backward adds to (fire and forget, off wave 0's path)
        if (g.zero_buf) {
            const long wgs = (long)g.batch * p.n_slabs, me = (long)b * p.n_slabs + slab;
            const long per = ((g.zero_n + wgs - 1) / wgs + 3) & ~3L;
            const long beg = me * per, end = min(g.zero_n, beg + per);
            for (long e = beg + 4 * (tid - 64); e < end; e += 4 * (NT - 64)) {
                if (e + 4 <= end) *reinterpret_cast<float4*>(g.zero_buf + e) = make_float4(0.f, 0.f, 0.f, 0.f);
                else for (long q = e; q < end; ++q) g.zero_buf[q] = 0.0f;
            }
        }
        return;
    }
    // ---- thin output layer (out_dim <= 4, pit.py:106): a row dot per output; the slab's share of the loss
    const int n = slab * ER + l15;
    const bool rv = n < p.n_out;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o >= g.n2) break;
        float part = 0.0f;
#pragma unroll
        for (int k = 0; k < D / 4; k += 4) {
            const float4 a = *reinterpret_cast<const float4*>(hs + l15 * HP + kq * (D / 4) + k);
            const float4 w = *reinterpret_cast<const float4*>(w2s + o * D + kq * (D / 4) + k);
            part += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        const float pred = part + w2s[4 * D + o];
        if (kq == 0 && rv) g.y[(row0 + l15) * g.n2 + o] = pred;
        if (LOSS) {
            double num = 0.0, den = 0.0;
            float t = l_t0, sc = l_sc0, sh = l_sh0;
            if (o > 0) {                                          // (further channels: their operands now - a round trip each)
                t = ldg_if(g.tru, (row0 + l15) * g.n2 + o, rv);
                sc = ldg_if(g.lscale, (long)n * g.n2 + o, rv && g.lscale != nullptr);
                sh = ldg_if(g.lshift, (long)n * g.n2 + o, rv && g.lscale != nullptr);
            }
            if (kq == 0 && rv) {
                const float q = g.lscale ? pred * sc + sh : pred;
                num = (double)pow_abs_p(t - q, g.loss_p);
                den = (double)pow_abs_p(t, g.loss_p);
            }
            num = wave_sum_d(num);
            den = wave_sum_d(den);
            if (lane == 0) {
                double* dst = g.lpart + (((long)b * g.n2 + o) * p.n_slabs + slab) * 2;
                dst[0] = num; dst[1] = den;
            }
        }
    }
    ESTAMP(0, 9);
}

// ------------------------------------------------------------------------------------------------ decoder backward
struct DecBwdArgs {
    pit_slab_plan p;
    const float* values; long ld_values, values_bstride; int batch;
    const float *pw, *qw;                     // the step's weights (dec_weights_body): P and Q = P (m - mbar)
    const float *w1, *w2; int n2;
    const float* z1;
    const float* d_y; long ld_dy;
    float* dz1;
    float* d_values; long dvalues_bstride;
    double* dscale;
    // the loss inside (d_y == NULL): d(pred) from the forward's partial sums, written to d_pred for the weight-gradient reductions
    const float *pred, *tru, *lscale, *lshift, *gseed; int loss_p; const double* lpart; float* d_pred; float* loss_out; float* norms_out;
    int um;
};

// ||t - q||_p and ||t||_p of series (b, o) from the slabs' partial sums (every lane gets both)
__device__ __forceinline__ void series_norms(const double* lpart, int b, int o, int n2, int slabs, int p, int lane, double& nn, double& dn) {
    double num = 0.0, den = 0.0;
    const double* src = lpart + ((long)b * n2 + o) * slabs * 2;
    for (int s = lane; s < slabs; s += 64) { num += src[2 * s]; den += src[2 * s + 1]; }
    num = wave_sum_d(num);
    den = wave_sum_d(den);
    nn = p == 1 ? num : sqrt(num);
    dn = p == 1 ? den : sqrt(den);
}

template <int H, int D, bool LOSS>
__global__ __launch_bounds__(4 * D) void decoder_bwd_kernel(DecBwdArgs g) {
    constexpr int NT = 4 * D, NW = D / 16, K0 = H * D, S1 = D / 16, XP = K0 + 4, P1 = D + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int UP = g.um + 4;
    float* ps_ = smem;                         // [H][16][UP]    P (normalised weights), read transposed by the d(values) contraction
    float* qs = ps_ + H * ER * UP;             // [H][16][UP]    Q = P (m - mbar)
    float* ut = qs + H * ER * UP;              // [um][D + 4]
    float* ds1 = ut + g.um * (D + 4);          // [16][P1]       dZ1
    float* dxs = ds1 + ER * P1;                // [16][XP]       dX
    double* wred = reinterpret_cast<double*>(dxs + ER * XP);     // [NW][H]
    int* keys_s = reinterpret_cast<int*>(wred + NW * H);         // [EU]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const pit_slab_plan& p = g.p;
    int b, slab;
    if (!slab_of_xcd(blockIdx.x, g.batch, p.n_slabs, b, slab)) return;
    const int nk = min(p.nkeys[slab], EU), nkup = (nk + 15) & ~15;
    const long row0 = (long)b * p.n_out + slab * ER;
    const int c1 = wave * 16 + l15;

    ESTAMP(1, 0);
    // ---- ONE block of loads: union keys, candidate records + saved row statistics, d(pred) or the loss's operands, gelu'
    // arguments, W2, W1 as the B operand of dX (B(k, n) = w1[k][n])
    int key[EU / 16];
    union_keys<D>(p, slab, tid, nkup, key);
    const int akey = p.keys[(long)slab * p.umax + (tid & (EU - 1))];   // slot -> key for the d(values) adds, through LDS
    const int npc = H * ER * g.um / 4;
    float4 pcp[WCP * 256 / NT], qcp[WCP * 256 / NT];
    wtile_request<NT>(g.pw + (long)slab * H * ER * g.um, npc, tid, pcp);
    wtile_request<NT>(g.qw + (long)slab * H * ER * g.um, npc, tid, qcp);
    // this thread's four dZ1 elements: rows er[u] (element e = tid + u * NT of the 16 x D tile), column ec
    const int ec = tid % D;
    // (output channel 0 - the only one of most models - is requested here with everything else; further channels, out_dim <= 4,
    // by the loop below: holding all of them would cost 60 more registers and a wave per SIMD)
    float z1v[4], dyv[4], prv[4], trv[4], lsc[4], lsh[4];
    const float w2c0 = g.w2[ec];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = (tid + u * NT) / D;
        const bool ok = slab * ER + r < p.n_out;
        z1v[u] = ldg_if(g.z1, (row0 + r) * D + ec, ok);
        dyv[u] = 0.0f; prv[u] = 0.0f; trv[u] = 0.0f; lsc[u] = 1.0f; lsh[u] = 0.0f;
        if (!LOSS) {
            dyv[u] = ldg_if(g.d_y, (row0 + r) * g.ld_dy, ok);
        } else {
            prv[u] = ldg_if(g.pred, (row0 + r) * g.n2, ok);
            trv[u] = ldg_if(g.tru, (row0 + r) * g.n2, ok);
            const bool aff = ok && g.lscale != nullptr;           // (no affine map: out-of-range loads give 0; scale 1 below)
            const float a = ldg_if(g.lscale, (long)(slab * ER + r) * g.n2, aff), bq = ldg_if(g.lshift, (long)(slab * ER + r) * g.n2, aff);
            lsc[u] = g.lscale ? a : 1.0f;
            lsh[u] = bq;
        }
    }
    // ---- the union's value rows (second round trip)
    // (the loss's partial sums of this sample's series, channel 0: the first 256 slabs' pairs with the other loads)
    float4 np4[4];
#pragma unroll
    for (int cq = 0; cq < 4; ++cq)
        np4[cq] = ldg4_if(reinterpret_cast<const float*>(g.lpart) + (long)b * g.n2 * p.n_slabs * 4, 4L * (lane + 64 * cq),
                          LOSS && lane + 64 * cq < p.n_slabs);
    float4 uv[EU / 16];
    gather_union<D>(g.values + (long)b * g.values_bstride, g.ld_values, key, nk, nkup, tid, uv);
    __builtin_amdgcn_sched_barrier(0);
    ESTAMP(1, 1);
    if (tid < EU) keys_s[tid] = akey;
    // ---- the loss of the step (utils.py:86-98) finished here: every wave sums this sample's norms itself (no hand-off), the
    // scalar is the first workgroup's last wave's
    float nrm0 = 0.0f, nrm1 = 1.0f;
    if (LOSS) {
        {
            double num = 0.0, den = 0.0;
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                num += __hiloint2double(__float_as_int(np4[cq].y), __float_as_int(np4[cq].x));
                den += __hiloint2double(__float_as_int(np4[cq].w), __float_as_int(np4[cq].z));
            }
            const double* src = g.lpart + (long)b * g.n2 * p.n_slabs * 2;
            for (int sl = lane + 256; sl < p.n_slabs; sl += 64) { num += src[2 * sl]; den += src[2 * sl + 1]; }
            num = wave_sum_d(num);
            den = wave_sum_d(den);
            nrm0 = (float)(g.loss_p == 1 ? num : sqrt(num));
            nrm1 = (float)(g.loss_p == 1 ? den : sqrt(den));
        }
        if (blockIdx.x == 0 && wave == NW - 1) {
            double tot = 0.0;
            for (int bb = 0; bb < g.batch; ++bb)
                for (int o = 0; o < g.n2; ++o) {
                    double nn, dn;
                    series_norms(g.lpart, bb, o, g.n2, p.n_slabs, g.loss_p, lane, nn, dn);
                    tot += (double)((float)(nn / dn / g.n2));
                    if (lane == 0 && g.norms_out) {
                        g.norms_out[((long)bb * g.n2 + o) * 2] = (float)nn;
                        g.norms_out[((long)bb * g.n2 + o) * 2 + 1] = (float)dn;
                    }
                }
            if (lane == 0) *g.loss_out = (float)tot;
        }
    }
    ESTAMP(1, 2);
    // ---- the weight tiles into LDS (row-major, pitch UP)
    wtile_park<NT>(ps_, g.um, npc, tid, pcp);
    wtile_park<NT>(qs, g.um, npc, tid, qcp);
    ESTAMP(1, 3);
    // ---- dZ1 = (dZ2 W2) * gelu'(Z1), dZ2 = d(pred): elementwise for the thin output layer
    const float gs = (LOSS && g.gseed) ? g.gseed[0] : 1.0f;
    auto dpred = [&](float pr, float tr, float sc, float sh, float nn, float dn) {
        const float d = pr * sc + sh - tr;
        float dnorm;
        if (g.loss_p == 1) dnorm = (d > 0.0f) ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        else dnorm = (nn > 0.0f) ? d / nn : 0.0f;
        return gs * dnorm * sc / (dn * g.n2);
    };
    float acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = (tid + u * NT) / D;
        const bool ok = slab * ER + r < p.n_out;
        float v = dyv[u];
        if (LOSS) {
            v = dpred(prv[u], trv[u], lsc[u], lsh[u], nrm0, nrm1);
            if (ec == 0 && ok) g.d_pred[(row0 + r) * g.n2] = v;
        }
        acc[u] = v * w2c0;
    }
    for (int o = 1; o < g.n2; ++o) {                                 // further output channels (Sod: 3): a round trip each
        const float w2o = g.w2[(long)o * D + ec];
        float nn = 0.0f, dn = 1.0f;
        if (LOSS) {
            double a, bq;
            series_norms(g.lpart, b, o, g.n2, p.n_slabs, g.loss_p, lane, a, bq);
            nn = (float)a; dn = (float)bq;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (tid + u * NT) / D;
            const bool ok = slab * ER + r < p.n_out;
            float v;
            if (!LOSS) {
                v = ldg_if(g.d_y, (row0 + r) * g.ld_dy + o, ok);
            } else {
                const long e = (long)(slab * ER + r) * g.n2 + o;
                const bool aff = ok && g.lscale != nullptr;
                const float a = ldg_if(g.lscale, e, aff), bq = ldg_if(g.lshift, e, aff);
                v = dpred(ldg_if(g.pred, (row0 + r) * g.n2 + o, ok), ldg_if(g.tru, (row0 + r) * g.n2 + o, ok),
                          g.lscale ? a : 1.0f, bq, nn, dn);
                if (ec == o && ok) g.d_pred[(row0 + r) * g.n2 + o] = v;
            }
            acc[u] += v * w2o;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = (tid + u * NT) / D;
        const bool ok = slab * ER + r < p.n_out;
        const float v = ok ? acc[u] * ((PIT_EDGE_DBG & 32) ? 1.0f : gelu_erf_grad(z1v[u])) : 0.0f;
        ds1[r * P1 + ec] = v;
        if (ok && !(PIT_EDGE_DBG & 512)) g.dz1[(row0 + r) * D + ec] = v;
    }
    ESTAMP(1, 4);
    park_union<D>(ut, nkup, tid, uv);
    ESTAMP(1, 5);
    // W1 as the B operand of dX (B(k, n) = w1[k][n]): requested only now - 32 registers that would otherwise be live across the
    // whole first half and cost the kernel a wave per SIMD; the barrier overlaps most of their (L2) latency
    float w1v[H][S1][4];
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
        for (int s = 0; s < S1; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) w1v[hh][s][e] = (PIT_EDGE_DBG & 8) ? 0.5f
                : g.w1[(long)(16 * s + 4 * kq + e) * K0 + hh * D + c1];
    __syncthreads();
    ESTAMP(1, 6);
    // ---- dX = dZ1 W1 (16 x H*D), LDS only
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S1; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(ds1 + l15 * P1 + 16 * s + 4 * kq);
            o0 = mfma_16x16x4(a.x, w1v[hh][s][0], o0);
            o1 = mfma_16x16x4(a.y, w1v[hh][s][1], o1);
            o0 = mfma_16x16x4(a.z, w1v[hh][s][2], o0);
            o1 = mfma_16x16x4(a.w, w1v[hh][s][3], o1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dxs[(4 * kq + i) * XP + hh * D + c1] = o0[i] + o1[i];
    }
    ESTAMP(1, 7);
    __syncthreads();
    ESTAMP(1, 8);
    // ---- d(values)[key(slot), 16 wave ..] += sum_{h, row} P_h[row][slot] dX[row, h*D + ..]   (first: its atomics are the long tail)
    for (int mt = 0; mt < nkup / 16; ++mt) {
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float* ap = ps_ + (h * ER + 4 * kq) * UP + 16 * mt + l15;        // A(i = slot, k = row): the tile read transposed
            const float* bp = dxs + (4 * kq) * XP + h * D + c1;
            a0 = mfma_16x16x4(ap[0], bp[0], a0);
            a1 = mfma_16x16x4(ap[UP], bp[XP], a1);
            a0 = mfma_16x16x4(ap[2 * UP], bp[2 * XP], a0);
            a1 = mfma_16x16x4(ap[3 * UP], bp[3 * XP], a1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = 16 * mt + 4 * kq + i;
            if (s < nk && !(PIT_EDGE_DBG & 16)) atomicAdd(g.d_values + (long)b * g.dvalues_bstride + (long)keys_s[s] * D + c1, a0[i] + a1[i]);
        }
    }
    ESTAMP(1, 9);
    // ---- d(scale): dc_h -= sum_{row, d} dX[row, h*D + d] * (Q_h U)[row, d]
    if (g.dscale && !(PIT_EDGE_DBG & 128)) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f32x4_t qu = tile_times_union<D>(qs + h * ER * UP, UP, ut, nkup, wave, l15, kq);
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) part += (double)qu[i] * (double)dxs[(4 * kq + i) * XP + h * D + c1];
            part = wave_sum_d(part);
            if (lane == 0) wred[wave * H + h] = part;
        }
        __syncthreads();
        if (tid < H) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += wred[w * H + tid];
            atomicAdd(g.dscale + (long)tid * PIT_DSCALE_SLOTS + ((int)blockIdx.x & (PIT_DSCALE_SLOTS - 1)), -tot);
        }
    }
    ESTAMP(1, 10);
}

// ------------------------------------------------------------------------------------------------ encoder forward
struct EncArgs {
    pit_slab_plan p;
    const float* mesh_in; int sdim, kd;                  // value channels [0, kd) are the key coordinates (train_darcy.py:51-55)
    const float* values; long ld_values, values_bstride; int dv, batch;
    const float* head; int head_is_scale;                // forward: lmda or c; backward: the saved c
    const float *w1, *b1, *w2, *b2;
    float* x;                                            // (batch * n_out, H * (kd + dv)): the attention's output, saved
    float *z1, *h, *z2, *y; long ldy;
    float* rowstat; float* scale_out;
    float* clear_buf; long clear_n;
    int n_att;
    // backward
    const float* d_y; long ld_dy; float* scratch; double* dscale;
};

// the candidate records of this wave's rows (a wave per row and pass, lane = candidate): first round trip
template <int NW>
struct EncRows { float m[ER / NW]; int cnt[ER / NW], j[ER / NW]; };
template <int NW>
__device__ __forceinline__ void enc_records(const pit_slab_plan& p, int slab, int wave, int lane, EncRows<NW>& r) {
#pragma unroll
    for (int ps = 0; ps < ER / NW; ++ps) {
        const int row_l = ps * NW + wave, n = slab * ER + row_l;
        r.cnt[ps] = ldi_if(p.cnt, n, n < p.n_out);
        r.m[ps] = p.m[(long)(slab * ER + row_l) * p.cap + (lane < p.cap ? lane : 0)];
        r.j[ps] = ldi_if(p.idx, (long)n * p.cap + lane, n < p.n_out && lane < p.cap);    // (slots beyond the count: never dereferenced)
    }
}
// ... and their value channels (second round trip): channel cc < kd is coordinate cc of the key, the others come from `values`;
// channels beyond kd + dv and lanes without a candidate load out of range (0).  EC = 4 or 8 channels at compile time: no branch.
template <int NW, int EC>
__device__ __forceinline__ void enc_values(const EncArgs& g, int b, int lane, const EncRows<NW>& r, float (&v)[ER / NW][EC]) {
    const int dc = g.kd + g.dv;
    const float* vb = g.values + (long)b * g.values_bstride;
#pragma unroll
    for (int ps = 0; ps < ER / NW; ++ps) {
        const bool valid = lane < min(r.cnt[ps], g.p.cap);
#pragma unroll
        for (int cc = 0; cc < EC; ++cc) {
            const bool coord = cc < g.kd;                        // (wave-uniform: selects the descriptor's base)
            const float* base = coord ? g.mesh_in : vb;
            const long off = coord ? (long)r.j[ps] * g.sdim + cc : (long)r.j[ps] * g.ld_values + (cc - g.kd);
            v[ps][cc] = ldg_if(base, off, valid && cc < dc && !(PIT_EDGE_DBG & 64));
        }
    }
}

template <int H, int D, int EC>
__global__ __launch_bounds__(4 * D) void encoder_fwd_kernel(EncArgs g, WeightsArgs wj, DecWArgs dj, int n_w) {
    constexpr int NT = 4 * D, NW = D / 16, S2 = D / 16, HP = D + 4, XP = 20, NP = ER / NW;
    __shared__ __attribute__((aligned(16))) float xs[ER * XP];
    __shared__ __attribute__((aligned(16))) float hs[ER * HP];
    if ((int)blockIdx.x >= g.n_att) {       // riders (256-thread workgroups): the processor's block weights, the decoder's weight tiles
        if (D == 64) {
            __shared__ __attribute__((aligned(16))) float tile[2 * 2 * ER * EU];
            const int id = (int)blockIdx.x - g.n_att;
            if (id < n_w) block_weights_body(wj, id); else dec_weights_body(dj, id - n_w, tile);
        }
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const pit_slab_plan& p = g.p;
    int b, slab;
    const bool mine = slab_of_xcd(blockIdx.x, g.batch, p.n_slabs, b, slab);
    if (mine) {
        const long row0 = (long)b * p.n_out + slab * ER;
        const int dc = g.kd + g.dv, K0 = H * dc;
        const int c1 = wave * 16 + l15;
        // ---- ONE block of loads: candidate records, row statistics, MLP operands
        EncRows<NW> rec;
        enc_records<NW>(p, slab, wave, lane, rec);
        float hraw[H];
#pragma unroll
        for (int h = 0; h < H; ++h) hraw[h] = g.head[h];
        float st_k[NP], st_k1[NP], st_min[NP];
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int n = slab * ER + ps * NW + wave, nn = n < p.n_out ? n : p.n_out - 1;
            st_k[ps] = p.stats[nn]; st_k1[ps] = p.stats[p.n_out + nn]; st_min[ps] = p.stats[2 * (long)p.n_out + nn];
        }
        float bv1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv1[e] = ldg_if(g.w1, (long)c1 * K0 + 4 * kq + e, 4 * kq + e < K0);
        float4 w2v[S2];
#pragma unroll
        for (int s = 0; s < S2; ++s) w2v[s] = *reinterpret_cast<const float4*>(g.w2 + (long)c1 * D + 16 * s + 4 * kq);
        const float bias1 = g.b1[c1], bias2 = g.b2[c1];
        // ---- second round trip: the candidates' value channels
        float v[NP][EC];
        enc_values<NW, EC>(g, b, lane, rec, v);
        __builtin_amdgcn_sched_barrier(0);
        // (this wave's rows of the tile: cleared by the wave itself, in order with its own writes below)
        for (int e = lane; e < NP * XP; e += 64) xs[((e / XP) * NW + wave) * XP + e % XP] = 0.0f;
        float c[H];
        head_scales<H>(hraw, g.head_is_scale, c);
        // ---- down-projection: one row per wave and pass
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int row_l = ps * NW + wave, n = slab * ER + row_l;
            const bool valid = lane < min(rec.cnt[ps], p.cap);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float T = quantile_lerp(__fmul_rn(c[h], st_k[ps]), __fmul_rn(c[h], st_k1[ps]), p.rank_w);
                const float smin = __fmul_rn(c[h], st_min[ps]);
                const float sv = __fmul_rn(rec.m[ps], c[h]);
                const bool keep = valid && sv <= T;
                const float pv = keep ? __expf(smin - sv) : 0.0f;
                const float rs = wave_sum(pv), qsum = wave_sum(pv * rec.m[ps]);
                const float inv = rs > 0.0f ? 1.0f / rs : 0.0f;
                float o[EC];
#pragma unroll
                for (int cc = 0; cc < EC; ++cc) o[cc] = wave_sum(pv * v[ps][cc]) * inv;
                if (lane < dc) {                                 // lane cc keeps channel cc
                    float mine_o = o[0];
#pragma unroll
                    for (int cc = 1; cc < EC; ++cc) mine_o = lane == cc ? o[cc] : mine_o;
                    xs[row_l * XP + h * dc + lane] = mine_o;
                    if (g.x && n < p.n_out) g.x[(row0 + row_l) * K0 + h * dc + lane] = mine_o;
                }
                if (lane == 0 && n < p.n_out && b == 0 && g.rowstat) {
                    float4 st; st.x = T; st.y = smin; st.z = inv; st.w = qsum * inv;
                    *reinterpret_cast<float4*>(g.rowstat + ((long)h * p.n_out + n) * 4) = st;
                    if (n == 0 && g.scale_out) g.scale_out[h] = c[h];
                }
            }
        }
        __syncthreads();
        // ---- encoder MLP (pit.py:110-111): one 16-k step for the first layer (K0 <= 16)
        {
            const float4 a = *reinterpret_cast<const float4*>(xs + l15 * XP + 4 * kq);
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            a0 = mfma_16x16x4(a.x, bv1[0], a0);
            a1 = mfma_16x16x4(a.y, bv1[1], a1);
            a0 = mfma_16x16x4(a.z, bv1[2], a0);
            a1 = mfma_16x16x4(a.w, bv1[3], a1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * kq + i;
                const float z = a0[i] + a1[i] + bias1;
                const float hv = gelu_erf(z);
                hs[r * HP + c1] = hv;
                if (g.z1 && slab * ER + r < p.n_out) {
                    g.z1[(row0 + r) * D + c1] = z;
                    g.h[(row0 + r) * D + c1] = hv;
                }
            }
        }
        __syncthreads();
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S2; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(hs + l15 * HP + 16 * s + 4 * kq);
            o0 = mfma_16x16x4(a.x, w2v[s].x, o0);
            o1 = mfma_16x16x4(a.y, w2v[s].y, o1);
            o0 = mfma_16x16x4(a.z, w2v[s].z, o0);
            o1 = mfma_16x16x4(a.w, w2v[s].w, o1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            if (slab * ER + r >= p.n_out) continue;
            const float vv = o0[i] + o1[i] + bias2;
            if (g.z2) g.z2[(row0 + r) * D + c1] = vv;
            g.y[(row0 + r) * g.ldy + c1] = gelu_erf(vv);
        }
    }
    if (g.clear_buf) {                                          // the step's gradient accumulators (engine.TrainStep): zeroed on the way out,
        const long nthreads = (long)g.n_att * NT;               // long before the first backward launch adds to them
        for (long i = (long)blockIdx.x * NT + tid; i < g.clear_n; i += nthreads) g.clear_buf[i] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------ encoder backward
template <int H, int D, int EC>
__global__ __launch_bounds__(4 * D) void encoder_bwd_kernel(EncArgs g) {
    constexpr int NT = 4 * D, NW = D / 16, S = D / 16, P1 = D + 4, XP = 20, NP = ER / NW;
    __shared__ __attribute__((aligned(16))) float ds2[ER * P1];
    __shared__ __attribute__((aligned(16))) float ds1[ER * P1];
    __shared__ __attribute__((aligned(16))) float dxs[ER * XP];
    __shared__ double wred[NW * H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const pit_slab_plan& p = g.p;
    int b, slab;
    if (!slab_of_xcd(blockIdx.x, g.batch, p.n_slabs, b, slab)) return;
    const long row0 = (long)b * p.n_out + slab * ER;
    const long rows = (long)g.batch * p.n_out;
    const int dc = g.kd + g.dv, K0 = H * dc;
    const int c1 = wave * 16 + l15;
    // ---- ONE block of loads: candidate records + saved row statistics, d_y and the gelu' arguments, W2 (B(k, n) = w2[k][n])
    EncRows<NW> rec;
    enc_records<NW>(p, slab, wave, lane, rec);
    float4 rs4[NP][H];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const int n = slab * ER + ps * NW + wave, nn = n < p.n_out ? n : p.n_out - 1;
#pragma unroll
        for (int h = 0; h < H; ++h) rs4[ps][h] = *reinterpret_cast<const float4*>(g.rowstat + ((long)h * p.n_out + nn) * 4);
    }
    float dyv[4], z2v[4], z1v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + u * NT, r = e / D, cc = e % D;
        const bool ok = slab * ER + r < p.n_out;
        dyv[u] = ldg_if(g.d_y, (row0 + r) * g.ld_dy + cc, ok);
        z2v[u] = ldg_if(g.z2, (row0 + r) * D + cc, ok);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) z1v[i] = ldg_if(g.z1, (row0 + 4 * kq + i) * D + c1, slab * ER + 4 * kq + i < p.n_out);
    float w2v[S][4];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) w2v[s][e] = g.w2[(long)(16 * s + 4 * kq + e) * D + c1];
    // ---- second round trip: the candidates' value channels
    float v[NP][EC];
    enc_values<NW, EC>(g, b, lane, rec, v);
    __builtin_amdgcn_sched_barrier(0);
    // ---- dZ2 = dY * gelu'(Z2)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + u * NT, r = e / D, cc = e % D;
        const float vv = dyv[u] * gelu_erf_grad(z2v[u]);
        ds2[r * P1 + cc] = vv;
        if (slab * ER + r < p.n_out) g.scratch[rows * D + (row0 + r) * D + cc] = vv;
    }
    // W1 (B(k, n) = w1[k][n], n < K0) for the one dX tile, wave 0: requested now, used after two barriers
    float w1x[S][4];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) w1x[s][e] = ldg_if(g.w1, (long)(16 * s + 4 * kq + e) * K0 + l15, wave == 0 && l15 < K0);
    __syncthreads();
    // ---- dZ1 = (dZ2 W2) * gelu'(Z1)
    {
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(ds2 + l15 * P1 + 16 * s + 4 * kq);
            a0 = mfma_16x16x4(a.x, w2v[s][0], a0);
            a1 = mfma_16x16x4(a.y, w2v[s][1], a1);
            a0 = mfma_16x16x4(a.z, w2v[s][2], a0);
            a1 = mfma_16x16x4(a.w, w2v[s][3], a1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            const float vv = (a0[i] + a1[i]) * gelu_erf_grad(z1v[i]);
            ds1[r * P1 + c1] = vv;
            if (slab * ER + r < p.n_out) g.scratch[(row0 + r) * D + c1] = vv;
        }
    }
    __syncthreads();
    // ---- dX = dZ1 W1 (16 x K0 <= 16 columns): one tile, wave 0
    if (wave == 0) {
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(ds1 + l15 * P1 + 16 * s + 4 * kq);
            o0 = mfma_16x16x4(a.x, w1x[s][0], o0);
            o1 = mfma_16x16x4(a.y, w1x[s][1], o1);
            o0 = mfma_16x16x4(a.z, w1x[s][2], o0);
            o1 = mfma_16x16x4(a.w, w1x[s][3], o1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dxs[(4 * kq + i) * XP + l15] = o0[i] + o1[i];
    }
    __syncthreads();
    // ---- d(scale) of the down-projection: dc_h -= sum_{row, cc} dX[row, h*dc + cc] * sum_j Q_h[row, j] V[j, cc]
    double part[H];
#pragma unroll
    for (int h = 0; h < H; ++h) part[h] = 0.0;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const int row_l = ps * NW + wave;
        const bool valid = lane < min(rec.cnt[ps], p.cap);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float sv = __fmul_rn(rec.m[ps], g.head[h]);
            const bool keep = valid && sv <= rs4[ps][h].x;
            const float qv = keep ? __expf(rs4[ps][h].y - sv) * rs4[ps][h].z * (rec.m[ps] - rs4[ps][h].w) : 0.0f;
            float dot = 0.0f;
#pragma unroll
            for (int cc = 0; cc < EC; ++cc) dot += v[ps][cc] * dxs[row_l * XP + min(h * dc + cc, XP - 1)];      // (v is 0 beyond dc)
            part[h] += (double)qv * (double)dot;
        }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const double sum = wave_sum_d(part[h]);
        if (lane == 0) wred[wave * H + h] = sum;
    }
    __syncthreads();
    if (tid < H) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += wred[w * H + tid];
        atomicAdd(g.dscale + (long)tid * PIT_DSCALE_SLOTS + ((int)blockIdx.x & (PIT_DSCALE_SLOTS - 1)), -tot);
    }
}

// ------------------------------------------------------------------------------------------------ union attention, any width
// The same union-tile contraction as the decoder's, WITHOUT the MLP, for value widths that are multiples of 64 (Vorticity / Cylinder:
// hid 256): a workgroup owns (sample, 16-row slab, 64-column chunk).  Forward `out = P U`; backward d(values) `+= P^T dO` (fp32
// atomics) and d(scale) `-= (Q U) . dO` from the same tiles, `d_out` read ONCE (the candidate-list backward gathers every row of it
// once per listing key, ~9x: posatt_sparse_bwd_kernel 170-187 us at Vorticity b=20).  `out` / `d_out` fp32 or bf16 (PIT_IO_*).
struct UAttArgs {
    pit_slab_plan p; int um, batch, dim;
    const float* values; long ld_values, values_bstride;
    const float *pw, *qw;
    void* out; long ld_out, out_bstride; int out16;
    const void* d_out; long ld_dout, dout_bstride; int dout16;
    float* d_values; long ld_dvalues, dvalues_bstride; double* dscale;
};

template <int H>
__global__ __launch_bounds__(256) void union_att_fwd_kernel(UAttArgs g) {
    constexpr int D = 64, NT = 256, XP = H * D + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int UP = g.um + 4;
    float* pt = smem;                          // [H][16][UP]
    float* ut = pt + H * ER * UP;              // [um][D + 4]
    float* xs = ut + g.um * (D + 4);           // [16][XP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const pit_slab_plan& p = g.p;
    const int chunks = g.dim / D;
    const int chunk = (int)blockIdx.x % chunks;
    int b, slab;
    if (!slab_of_xcd((int)blockIdx.x / chunks, g.batch, p.n_slabs, b, slab)) return;
    const int nk = min(p.nkeys[slab], EU), nkup = (nk + 15) & ~15;
    const int c1 = wave * 16 + l15;
    int key[EU / 16];
    union_keys<D>(p, slab, tid, nkup, key);
    const int npc = H * ER * g.um / 4;
    float4 pcp[WCP];
    wtile_request<NT>(g.pw + (long)slab * H * ER * g.um, npc, tid, pcp);
    float4 uv[EU / 16];
    gather_union<D>(g.values + (long)b * g.values_bstride + chunk * D, g.ld_values, key, nk, nkup, tid, uv);
    __builtin_amdgcn_sched_barrier(0);
    wtile_park<NT>(pt, g.um, npc, tid, pcp);
    park_union<D>(ut, nkup, tid, uv);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const f32x4_t o = tile_times_union<D>(pt + h * ER * UP, UP, ut, nkup, wave, l15, kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[(4 * kq + i) * XP + h * D + c1] = o[i];
    }
    __syncthreads();
    // the tile to memory as 4-column pieces (16 B of fp32, 8 B of bf16): column h*64 + 4q of the tile is h*dim + chunk*64 + 4q of out
#pragma unroll
    for (int u = 0; u < H * ER * D / 4 / NT; ++u) {
        const int e = tid + u * NT, r = e / (H * D / 4), q4 = e % (H * D / 4), h = q4 / (D / 4), q = q4 % (D / 4);
        const int n = slab * ER + r;
        if (n >= p.n_out) continue;
        const float4 v = *reinterpret_cast<const float4*>(xs + r * XP + h * D + 4 * q);
        const long o = (long)b * g.out_bstride + (long)n * g.ld_out + (long)h * g.dim + chunk * D + 4 * q;
        if (g.out16) {
            uint2 pk;
            pk.x = (unsigned)f_to_bf16(v.x) | ((unsigned)f_to_bf16(v.y) << 16);
            pk.y = (unsigned)f_to_bf16(v.z) | ((unsigned)f_to_bf16(v.w) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.out) + o) = pk;
        } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + o) = v;
        }
    }
}

template <int H>
__global__ __launch_bounds__(256) void union_att_bwd_kernel(UAttArgs g) {
    constexpr int D = 64, NT = 256, XP = H * D + 4, NW = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int UP = g.um + 4;
    float* ps_ = smem;                         // [H][16][UP]
    float* qs = ps_ + H * ER * UP;             // [H][16][UP]
    float* ut = qs + H * ER * UP;              // [um][D + 4]
    float* dxs = ut + g.um * (D + 4);          // [16][XP]   this chunk's columns of d_out, both heads
    double* wred = reinterpret_cast<double*>(dxs + ER * XP);     // [NW][H]
    int* keys_s = reinterpret_cast<int*>(wred + NW * H);         // [EU]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const pit_slab_plan& p = g.p;
    const int chunks = g.dim / D;
    const int chunk = (int)blockIdx.x % chunks;
    int b, slab;
    if (!slab_of_xcd((int)blockIdx.x / chunks, g.batch, p.n_slabs, b, slab)) return;
    const int nk = min(p.nkeys[slab], EU), nkup = (nk + 15) & ~15;
    const int c1 = wave * 16 + l15;
    int key[EU / 16];
    union_keys<D>(p, slab, tid, nkup, key);
    const int akey = p.keys[(long)slab * p.umax + (tid & (EU - 1))];
    const int npc = H * ER * g.um / 4;
    float4 pcp[WCP], qcp[WCP];
    wtile_request<NT>(g.pw + (long)slab * H * ER * g.um, npc, tid, pcp);
    wtile_request<NT>(g.qw + (long)slab * H * ER * g.um, npc, tid, qcp);
    // this chunk of d_out: 4-column pieces (rows beyond the mesh and nothing else load out of range)
    float4 dv[H * ER * D / 4 / NT];
#pragma unroll
    for (int u = 0; u < H * ER * D / 4 / NT; ++u) {
        const int e = tid + u * NT, r = e / (H * D / 4), q4 = e % (H * D / 4), h = q4 / (D / 4), q = q4 % (D / 4);
        const int n = slab * ER + r;
        const long o = (long)b * g.dout_bstride + (long)n * g.ld_dout + (long)h * g.dim + chunk * D + 4 * q;
        if (g.dout16) {
            const __amdgpu_buffer_rsrc_t r16 = wide_rsrc(g.d_out);
            const unsigned off = n < p.n_out ? (unsigned)(o * 2) : OOB;
            const unsigned lo = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r16, (int)off, 0, 0);
            const unsigned hi = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r16, (int)(n < p.n_out ? off + 4u : OOB), 0, 0);
            dv[u] = make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                                __uint_as_float(hi & 0xffff0000u));
        } else {
            dv[u] = ldg4_if(reinterpret_cast<const float*>(g.d_out), o, n < p.n_out);
        }
    }
    float4 uv[EU / 16];
    gather_union<D>(g.values + (long)b * g.values_bstride + chunk * D, g.ld_values, key, nk, nkup, tid, uv);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < EU) keys_s[tid] = akey;
    wtile_park<NT>(ps_, g.um, npc, tid, pcp);
    wtile_park<NT>(qs, g.um, npc, tid, qcp);
#pragma unroll
    for (int u = 0; u < H * ER * D / 4 / NT; ++u) {
        const int e = tid + u * NT, r = e / (H * D / 4), q4 = e % (H * D / 4);
        *reinterpret_cast<float4*>(dxs + r * XP + 4 * q4) = dv[u];
    }
    park_union<D>(ut, nkup, tid, uv);
    __syncthreads();
    // d(values)[key(slot), chunk*64 + 16 wave ..] += sum_{h, row} P_h[row][slot] dO[row, h*dim + chunk*64 + ..]
    if (g.d_values) {
        for (int mt = 0; mt < nkup / 16; ++mt) {
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float* ap = ps_ + (h * ER + 4 * kq) * UP + 16 * mt + l15;
                const float* bp = dxs + (4 * kq) * XP + h * D + c1;
                a0 = mfma_16x16x4(ap[0], bp[0], a0);
                a1 = mfma_16x16x4(ap[UP], bp[XP], a1);
                a0 = mfma_16x16x4(ap[2 * UP], bp[2 * XP], a0);
                a1 = mfma_16x16x4(ap[3 * UP], bp[3 * XP], a1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sl = 16 * mt + 4 * kq + i;
                if (sl < nk) atomicAdd(g.d_values + (long)b * g.dvalues_bstride + (long)keys_s[sl] * g.ld_dvalues + chunk * D + c1, a0[i] + a1[i]);
            }
        }
    }
    if (g.dscale) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f32x4_t qu = tile_times_union<D>(qs + h * ER * UP, UP, ut, nkup, wave, l15, kq);
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) part += (double)qu[i] * (double)dxs[(4 * kq + i) * XP + h * D + c1];
            part = wave_sum_d(part);
            if (lane == 0) wred[wave * H + h] = part;
        }
        __syncthreads();
        if (tid < H) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += wred[w * H + tid];
            atomicAdd(g.dscale + (long)tid * PIT_DSCALE_SLOTS + ((int)blockIdx.x & (PIT_DSCALE_SLOTS - 1)), -tot);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool plan_ok(const pit_slab_plan* p, bool needs_union) {
    if (!p || !p->stats || !p->idx || !p->cnt || !p->m) return false;
    if (p->rows != ER) return false;                 // (plans of taller slabs belong to pit_fold.hip)
    if (p->n_out <= 0 || p->n_in <= 0 || p->cap <= 0 || p->cap > 64 || p->n_slabs != (p->n_out + ER - 1) / ER) return false;
    if (needs_union && (!p->slot || !p->keys || !p->nkeys || p->umax != EU)) return false;
    return true;
}
bool hid_ok(int n_head, int dim) { return (n_head == 1 || n_head == 2) && (dim == 32 || dim == 64); }

// slots of the union tiles a launch reserves in LDS (what decides how many workgroups a CU holds: Darcy's decoder backward takes
// 30 KB at 32 slots, 39 KB at 48 - four or more workgroups per CU, all 928 resident at once - and 47 KB at 64: three, a second round)
int union_slots(int max_union) { return max_union <= 32 ? 32 : (max_union <= 48 ? 48 : 64); }
template <int H, int D>
size_t dec_fwd_smem(int um) { return (size_t)(H * ER * (um + 4) + um * (D + 4) + ER * (H * D + 4) + ER * (D + 4) + 4 * D + 4) * 4; }
template <int H, int D>
size_t dec_bwd_smem(int um) {
    return (size_t)(2 * H * ER * (um + 4) + um * (D + 4) + ER * (D + 4) + ER * (H * D + 4)) * 4 + (size_t)(D / 16) * H * 8 + EU * 4;
}

}  // namespace

extern "C" int pit_edge_supported(int n_head, int dim, int batch, int rows_per_sample) {
    if (!hid_ok(n_head, dim) || batch <= 0 || rows_per_sample <= 0) return 0;
    const long rows = (long)batch * rows_per_sample;
    return rows >= 256 && rows <= PIT_EDGE_MAX_ROWS;
}

extern "C" int pit_slab_plan_build(const float* mesh_out, const float* mesh_in, int n_out, int n_in, int space_dim, int metric,
                                   float period, const int* nbr_idx, const int* nbr_cnt, int cap, int rows_per_slab, float* m,
                                   unsigned short* slot, int* keys, int* nkeys, int* report, void* stream) {
    if (!mesh_out || !mesh_in || !nbr_idx || !nbr_cnt || !m || !slot || !keys || !nkeys || !report) return PIT_ERR_NULL;
    if (n_out <= 0 || n_in <= 0 || n_in > 16384 || space_dim < 1 || space_dim > 3 || cap <= 0) return PIT_ERR_SIZE;
    if (rows_per_slab != 16 && rows_per_slab != 64 && rows_per_slab != 128 && rows_per_slab != 256) return PIT_ERR_SIZE;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    SlabBuildArgs a;
    a.mesh_out = mesh_out; a.mesh_in = mesh_in; a.n_out = n_out; a.n_in = n_in; a.sdim = space_dim;
    a.used = (metric == PIT_METRIC_PERIODIC1D) ? 1 : space_dim;
    a.periodic = metric != PIT_METRIC_EUCLID; a.period = period;
    a.idx = nbr_idx; a.cnt = nbr_cnt; a.cap = cap; a.umax = EU;
    a.m = m; a.slot = slot; a.keys = keys; a.nkeys = nkeys; a.report = report; a.er = rows_per_slab;
    hipLaunchKernelGGL(slab_plan_kernel, dim3((unsigned)((n_out + rows_per_slab - 1) / rows_per_slab)), dim3(256), 0, (hipStream_t)stream, a);
    PIT_CHECK_LAUNCH();
    return 0;
}

#define PIT_EDGE_DISPATCH(H_, D_, CALL_)                                         \
    do {                                                                         \
        if ((H_) == 1 && (D_) == 32) { CALL_(1, 32); }                           \
        else if ((H_) == 1) { CALL_(1, 64); }                                    \
        else if ((D_) == 32) { CALL_(2, 32); }                                   \
        else { CALL_(2, 64); }                                                   \
    } while (0)

namespace {
// pit_decoder_weights' argument checks -> kernel arguments
int fill_dec_weights(DecWArgs& g, const pit_slab_plan* plan, const float* head, int head_is_scale, int n_head, int max_union,
                     int max_count, float* pw, float* qw, float* scale_out, const float* w1, float* w1f, int dim) {
    if (!plan_ok(plan, true) || !head || !pw) return PIT_ERR_NULL;
    if ((w1 == nullptr) != (w1f == nullptr)) return PIT_ERR_NULL;
    if (w1f && (dim < 16 || dim % 16 != 0 || !aligned16(w1) || !aligned16(w1f))) return PIT_ERR_SIZE;
    g.w1 = w1; g.w1f = w1f; g.dim = dim; g.rb = ER;
    if ((n_head != 1 && n_head != 2) || max_union < 1 || max_union > EU || max_count < 1 || max_count > 64) return PIT_ERR_UNSUPPORTED;
    if (!aligned16(pw) || (qw && !aligned16(qw))) return PIT_ERR_SIZE;
    g.p = *plan; g.head = head; g.head_is_scale = head_is_scale; g.n_head = n_head; g.um = union_slots(max_union);
    g.lpr = max_count <= 16 ? 16 : (max_count <= 32 ? 32 : 64);
    g.pw = pw; g.qw = qw; g.scale_out = scale_out;
    return 0;
}
}  // namespace

extern "C" int pit_decoder_weights(const pit_slab_plan* plan, const float* head, int head_is_scale, int n_head, int max_union,
                                   int max_count, float* pw, float* qw, float* scale_out, const float* w1, float* w1f, int dim,
                                   void* stream) {
    DecWArgs g = DecWArgs();
    if (int rc = fill_dec_weights(g, plan, head, head_is_scale, n_head, max_union, max_count, pw, qw, scale_out, w1, w1f, dim)) return rc;
    hipLaunchKernelGGL(decoder_weights_kernel, dim3((unsigned)plan->n_slabs), dim3(256), 0, (hipStream_t)stream, g);
    PIT_CHECK_LAUNCH();
    return 0;
}

// The same weights for a plan whose slabs are 64 / 128 / 256 rows tall (pit_fold.hip): a workgroup still forms 16 rows; head h's
// rows of slab s are the `rows` x um block at pw + (s * n_head + h) * rows * um.
extern "C" int pit_fold_weights(const pit_slab_plan* plan, const float* head, int head_is_scale, int n_head, int max_union,
                                int max_count, float* pw, float* qw, float* scale_out, unsigned short* pw16, unsigned short* qw16,
                                void* stream) {
    if (!plan || !plan->stats || !plan->idx || !plan->cnt || !plan->m || !plan->slot || !plan->keys || !plan->nkeys || !head || !pw)
        return PIT_ERR_NULL;
    if ((plan->rows != 64 && plan->rows != 128 && plan->rows != 256) || plan->umax != EU || plan->cap <= 0 || plan->cap > 64 ||
        plan->n_slabs != (plan->n_out + plan->rows - 1) / plan->rows) return PIT_ERR_SIZE;
    if ((n_head != 1 && n_head != 2) || max_union < 1 || max_union > EU || max_count < 1 || max_count > 64) return PIT_ERR_UNSUPPORTED;
    if (!aligned16(pw) || (qw && !aligned16(qw))) return PIT_ERR_SIZE;
    DecWArgs g = DecWArgs();
    g.p = *plan; g.head = head; g.head_is_scale = head_is_scale; g.n_head = n_head; g.um = union_slots(max_union);
    g.lpr = max_count <= 16 ? 16 : (max_count <= 32 ? 32 : 64);
    g.pw = pw; g.qw = qw; g.scale_out = scale_out; g.rb = plan->rows; g.pw16 = pw16; g.qw16 = qw16;
    hipLaunchKernelGGL(decoder_weights_kernel, dim3((unsigned)(plan->n_slabs * (plan->rows / ER))), dim3(256), 0, (hipStream_t)stream, g);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_decoder_fwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                               int n_head, int dim, const float* pw,
                               const float* w1, const float* w1f, const float* b1, const float* w2, const float* b2, int n2,
                               float* x, float* z1, float* h, float* y, float* zero_buf, long zero_n,
                               const float* loss_true, const float* loss_scale, const float* loss_shift, int loss_p, double* loss_part,
                               int max_union, void* stream) {
    if (max_union < 1 || max_union > EU) return PIT_ERR_UNSUPPORTED;
    if (w1f && !aligned16(w1f)) return PIT_ERR_SIZE;
    if (!plan_ok(plan, true) || !values || !pw || !w1 || !b1 || !w2 || !b2 || !y) return PIT_ERR_NULL;
    if (!hid_ok(n_head, dim) || n2 < 1 || n2 > 4 || batch <= 0) return PIT_ERR_UNSUPPORTED;
    if ((z1 == nullptr) != (h == nullptr)) return PIT_ERR_NULL;
    if (ld_values % 4 || values_bstride % 4 || !aligned16(values) || !aligned16(w1) || !aligned16(pw) || (zero_buf && !aligned16(zero_buf)))
        return PIT_ERR_SIZE;
    if (loss_part && (!loss_true || (loss_p != 1 && loss_p != 2) || (loss_scale == nullptr) != (loss_shift == nullptr))) return PIT_ERR_UNSUPPORTED;
    DecFwdArgs g;
    g.p = *plan; g.values = values; g.ld_values = ld_values; g.values_bstride = values_bstride; g.batch = batch;
    g.pw = pw; g.w1 = w1; g.w1f = w1f; g.b1 = b1; g.w2 = w2; g.b2 = b2; g.n2 = n2;
    g.x = x; g.z1 = z1; g.h = h; g.y = y; g.zero_buf = zero_buf; g.zero_n = zero_buf ? zero_n : 0;
    g.tru = loss_true; g.lscale = loss_scale; g.lshift = loss_shift; g.loss_p = loss_p; g.lpart = loss_part;
    g.um = union_slots(max_union);
    const dim3 grid((unsigned)slab_grid(batch, plan->n_slabs));
    hipStream_t s = (hipStream_t)stream;
#define PIT_DF(H_, D_)                                                                                                                \
    do {                                                                                                                              \
        if (loss_part) hipLaunchKernelGGL((decoder_fwd_kernel<H_, D_, true>), grid, dim3(4 * D_), (dec_fwd_smem<H_, D_>(g.um)), s, g); \
        else hipLaunchKernelGGL((decoder_fwd_kernel<H_, D_, false>), grid, dim3(4 * D_), (dec_fwd_smem<H_, D_>(g.um)), s, g);          \
    } while (0)
    PIT_EDGE_DISPATCH(n_head, dim, PIT_DF);
#undef PIT_DF
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_decoder_bwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                               int n_head, int dim, const float* pw, const float* qw,
                               const float* w1, const float* w2, int n2, const float* z1,
                               const float* d_y, long ld_dy, float* dz1, float* d_values, long dvalues_bstride, double* dscale,
                               const float* loss_pred, const float* loss_true, const float* loss_scale, const float* loss_shift,
                               const float* loss_seed, int loss_p, const double* loss_part, float* d_pred, float* loss_out,
                               float* norms_out, int max_union, void* stream) {
    if (max_union < 1 || max_union > EU) return PIT_ERR_UNSUPPORTED;
    if (!plan_ok(plan, true) || !values || !pw || !qw || !w1 || !w2 || !z1 || !dz1 || !d_values) return PIT_ERR_NULL;
    if (!hid_ok(n_head, dim) || n2 < 1 || n2 > 4 || batch <= 0) return PIT_ERR_UNSUPPORTED;
    if (!d_y && (!loss_pred || !loss_true || !loss_part || !d_pred || !loss_out || (loss_p != 1 && loss_p != 2) ||
                 (loss_scale == nullptr) != (loss_shift == nullptr))) return PIT_ERR_NULL;
    if (d_y && ld_dy < n2) return PIT_ERR_SIZE;
    if (ld_values % 4 || values_bstride % 4 || !aligned16(values) || !aligned16(pw) || !aligned16(qw)) return PIT_ERR_SIZE;
    DecBwdArgs g;
    g.p = *plan; g.values = values; g.ld_values = ld_values; g.values_bstride = values_bstride; g.batch = batch;
    g.pw = pw; g.qw = qw; g.w1 = w1; g.w2 = w2; g.n2 = n2; g.z1 = z1; g.d_y = d_y; g.ld_dy = ld_dy;
    g.dz1 = dz1; g.d_values = d_values; g.dvalues_bstride = dvalues_bstride; g.dscale = dscale;
    g.pred = loss_pred; g.tru = loss_true; g.lscale = loss_scale; g.lshift = loss_shift; g.gseed = loss_seed; g.loss_p = loss_p;
    g.lpart = loss_part; g.d_pred = d_pred; g.loss_out = loss_out; g.norms_out = norms_out;
    g.um = union_slots(max_union);
    const dim3 grid((unsigned)slab_grid(batch, plan->n_slabs));
    hipStream_t s = (hipStream_t)stream;
#define PIT_DB(H_, D_)                                                                                                            \
    do {                                                                                                                          \
        if (!d_y) hipLaunchKernelGGL((decoder_bwd_kernel<H_, D_, true>), grid, dim3(4 * D_), (dec_bwd_smem<H_, D_>(g.um)), s, g);  \
        else hipLaunchKernelGGL((decoder_bwd_kernel<H_, D_, false>), grid, dim3(4 * D_), (dec_bwd_smem<H_, D_>(g.um)), s, g);      \
    } while (0)
    PIT_EDGE_DISPATCH(n_head, dim, PIT_DB);
#undef PIT_DB
    PIT_CHECK_LAUNCH();
    return 0;
}

namespace {
int fill_enc(EncArgs& g, const pit_slab_plan* plan, const float* mesh_in, int space_dim, int coord_dims, const float* values,
             long ld_values, long values_bstride, int dv, int batch, int n_head, int dim) {
    if (!plan_ok(plan, false) || !mesh_in || !values) return PIT_ERR_NULL;
    if (!hid_ok(n_head, dim) || batch <= 0 || space_dim < 1 || space_dim > 3 || coord_dims < 0 || coord_dims > space_dim || dv < 0 ||
        coord_dims + dv < 1 || coord_dims + dv > 8 || n_head * (coord_dims + dv) > 16) return PIT_ERR_UNSUPPORTED;
    g = EncArgs();
    g.p = *plan; g.mesh_in = mesh_in; g.sdim = space_dim; g.kd = coord_dims; g.values = values; g.ld_values = ld_values;
    g.values_bstride = values_bstride; g.dv = dv; g.batch = batch;
    return 0;
}
}  // namespace

extern "C" int pit_encoder_fwd(const pit_slab_plan* plan, const float* mesh_in, int space_dim, int coord_dims,
                               const float* values, long ld_values, long values_bstride, int value_dim, int batch,
                               int n_head, int dim, const float* head, int head_is_scale,
                               const float* w1, const float* b1, const float* w2, const float* b2,
                               float* x, float* z1, float* h, float* z2, float* y, long ldy, float* rowstat, float* scale_out,
                               float* clear_buf, long clear_n, const pit_block_weights_job* weights,
                               const pit_decoder_weights_job* dec_weights, void* stream) {
    EncArgs g;
    if (int rc = fill_enc(g, plan, mesh_in, space_dim, coord_dims, values, ld_values, values_bstride, value_dim, batch, n_head, dim)) return rc;
    if (!head || !w1 || !b1 || !w2 || !b2 || !y) return PIT_ERR_NULL;
    if ((z1 == nullptr) != (h == nullptr) || ldy < dim || !aligned16(w2)) return PIT_ERR_SIZE;
    g.head = head; g.head_is_scale = head_is_scale; g.w1 = w1; g.b1 = b1; g.w2 = w2; g.b2 = b2;
    g.x = x; g.z1 = z1; g.h = h; g.z2 = z2; g.y = y; g.ldy = ldy; g.rowstat = rowstat; g.scale_out = scale_out;
    g.clear_buf = clear_n > 0 ? clear_buf : nullptr; g.clear_n = clear_n;
    g.n_att = slab_grid(batch, plan->n_slabs);
    WeightsArgs wa = WeightsArgs();
    int n_w = 0;
    bool own_launch = false;
    if (weights) {
        if (int rc = fill_weights_args(wa, weights->mesh, weights->n_pts, weights->space_dim, weights->metric, weights->period,
                                       weights->n_layers, weights->heads, weights->head_is_scale, weights->n_head, weights->e, weights->q,
                                       weights->inv, weights->rowstat, weights->scale_out)) return rc;
        if (dim == 64) n_w = block_weights_wgs(wa.n_layers, wa.n_head, wa.L); else own_launch = true;
    }
    DecWArgs da = DecWArgs();
    int n_dw = 0;
    if (dec_weights) {
        if (int rc = fill_dec_weights(da, dec_weights->plan, dec_weights->head, dec_weights->head_is_scale, dec_weights->n_head,
                                      dec_weights->max_union, dec_weights->max_count, dec_weights->pw, dec_weights->qw,
                                      dec_weights->scale_out, dec_weights->w1, dec_weights->w1f, dec_weights->dim)) return rc;
        if (dim == 64) n_dw = da.p.n_slabs;
    }
    const dim3 grid((unsigned)(g.n_att + n_w + n_dw));
    hipStream_t s = (hipStream_t)stream;
#define PIT_EF(H_, D_) do { if (coord_dims + value_dim <= 4) hipLaunchKernelGGL((encoder_fwd_kernel<H_, D_, 4>), grid, dim3(4 * D_), 0, s, g, wa, da, n_w); \
                            else hipLaunchKernelGGL((encoder_fwd_kernel<H_, D_, 8>), grid, dim3(4 * D_), 0, s, g, wa, da, n_w); } while (0)
    PIT_EDGE_DISPATCH(n_head, dim, PIT_EF);
#undef PIT_EF
    PIT_CHECK_LAUNCH();
    if (dec_weights && dim != 64)
        if (int rc = pit_decoder_weights(dec_weights->plan, dec_weights->head, dec_weights->head_is_scale, dec_weights->n_head,
                                         dec_weights->max_union, dec_weights->max_count, dec_weights->pw, dec_weights->qw,
                                         dec_weights->scale_out, dec_weights->w1, dec_weights->w1f, dec_weights->dim, stream)) return rc;
    if (own_launch)
        return pit_block_weights(weights->mesh, weights->n_pts, weights->space_dim, weights->metric, weights->period, weights->n_layers,
                                 weights->heads, weights->head_is_scale, weights->n_head, weights->e, weights->q, weights->inv,
                                 weights->rowstat, weights->scale_out, stream);
    return 0;
}

extern "C" int pit_encoder_bwd(const pit_slab_plan* plan, const float* mesh_in, int space_dim, int coord_dims,
                               const float* values, long ld_values, long values_bstride, int value_dim, int batch,
                               int n_head, int dim, const float* scale, const float* rowstat,
                               const float* w1, const float* w2, const float* z1, const float* z2,
                               const float* d_y, long ld_dy, float* scratch, double* dscale, void* stream) {
    EncArgs g;
    if (int rc = fill_enc(g, plan, mesh_in, space_dim, coord_dims, values, ld_values, values_bstride, value_dim, batch, n_head, dim)) return rc;
    if (!w1 || !w2 || !z1 || !z2 || !d_y || !scratch) return PIT_ERR_NULL;
    if (!dscale || !scale || !rowstat || !aligned16(rowstat)) return PIT_ERR_NULL;
    if (ld_dy < dim) return PIT_ERR_SIZE;
    g.head = scale; g.head_is_scale = 1; g.rowstat = const_cast<float*>(rowstat); g.w1 = w1; g.w2 = w2;
    g.z1 = const_cast<float*>(z1); g.z2 = const_cast<float*>(z2); g.d_y = d_y; g.ld_dy = ld_dy; g.scratch = scratch; g.dscale = dscale;
    const dim3 grid((unsigned)slab_grid(batch, plan->n_slabs));
    hipStream_t s = (hipStream_t)stream;
#define PIT_EB(H_, D_) do { if (coord_dims + value_dim <= 4) hipLaunchKernelGGL((encoder_bwd_kernel<H_, D_, 4>), grid, dim3(4 * D_), 0, s, g); \
                            else hipLaunchKernelGGL((encoder_bwd_kernel<H_, D_, 8>), grid, dim3(4 * D_), 0, s, g); } while (0)
    PIT_EDGE_DISPATCH(n_head, dim, PIT_EB);
#undef PIT_EB
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_union_att_supported(int n_head, int dim, int batch, int rows_per_sample) {
    if ((n_head != 1 && n_head != 2) || dim < 64 || dim % 64 != 0 || batch <= 0 || rows_per_sample <= 0) return 0;
    return (long)batch * rows_per_sample * dim <= (1L << 28);      // 32-bit buffer offsets of the bf16 / fp32 tensors
}

extern "C" int pit_union_att_fwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                                 int n_head, int dim, const float* pw, void* out, long ld_out, long out_bstride, int out_bf16,
                                 int max_union, void* stream) {
    if (!plan_ok(plan, true) || !values || !pw || !out) return PIT_ERR_NULL;
    if (!pit_union_att_supported(n_head, dim, batch, plan->n_out) || max_union < 1 || max_union > EU) return PIT_ERR_UNSUPPORTED;
    if (ld_values % 4 || values_bstride % 4 || !aligned16(values) || !aligned16(pw) || ld_out % 4 || out_bstride % 4 ||
        (reinterpret_cast<uintptr_t>(out) & (out_bf16 ? 7 : 15))) return PIT_ERR_SIZE;
    UAttArgs g = UAttArgs();
    g.p = *plan; g.um = union_slots(max_union); g.batch = batch; g.dim = dim;
    g.values = values; g.ld_values = ld_values; g.values_bstride = values_bstride; g.pw = pw;
    g.out = out; g.ld_out = ld_out; g.out_bstride = out_bstride; g.out16 = out_bf16;
    const dim3 grid((unsigned)(slab_grid(batch, plan->n_slabs) * (dim / 64)));
    const size_t sm1 = (size_t)(1 * ER * (g.um + 4) + g.um * 68 + ER * (1 * 64 + 4)) * 4;
    const size_t sm2 = (size_t)(2 * ER * (g.um + 4) + g.um * 68 + ER * (2 * 64 + 4)) * 4;
    if (n_head == 1) hipLaunchKernelGGL(union_att_fwd_kernel<1>, grid, dim3(256), sm1, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(union_att_fwd_kernel<2>, grid, dim3(256), sm2, (hipStream_t)stream, g);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_union_att_bwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                                 int n_head, int dim, const float* pw, const float* qw,
                                 const void* d_out, long ld_dout, long dout_bstride, int dout_bf16,
                                 float* d_values, long ld_dvalues, long dvalues_bstride, double* dscale, int max_union, void* stream) {
    if (!plan_ok(plan, true) || !values || !pw || !qw || !d_out || (!d_values && !dscale)) return PIT_ERR_NULL;
    if (!pit_union_att_supported(n_head, dim, batch, plan->n_out) || max_union < 1 || max_union > EU) return PIT_ERR_UNSUPPORTED;
    if (ld_values % 4 || values_bstride % 4 || !aligned16(values) || !aligned16(pw) || !aligned16(qw) || ld_dout % 4 || dout_bstride % 4 ||
        (reinterpret_cast<uintptr_t>(d_out) & (dout_bf16 ? 7 : 15))) return PIT_ERR_SIZE;
    UAttArgs g = UAttArgs();
    g.p = *plan; g.um = union_slots(max_union); g.batch = batch; g.dim = dim;
    g.values = values; g.ld_values = ld_values; g.values_bstride = values_bstride; g.pw = pw; g.qw = qw;
    g.d_out = d_out; g.ld_dout = ld_dout; g.dout_bstride = dout_bstride; g.dout16 = dout_bf16;
    g.d_values = d_values; g.ld_dvalues = ld_dvalues; g.dvalues_bstride = dvalues_bstride; g.dscale = dscale;
    const dim3 grid((unsigned)(slab_grid(batch, plan->n_slabs) * (dim / 64)));
    const size_t sm1 = (size_t)(2 * 1 * ER * (g.um + 4) + g.um * 68 + ER * (1 * 64 + 4)) * 4 + 4 * 1 * 8 + EU * 4;
    const size_t sm2 = (size_t)(2 * 2 * ER * (g.um + 4) + g.um * 68 + ER * (2 * 64 + 4)) * 4 + 4 * 2 * 8 + EU * 4;
    if (n_head == 1) hipLaunchKernelGGL(union_att_bwd_kernel<1>, grid, dim3(256), sm1, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(union_att_bwd_kernel<2>, grid, dim3(256), sm2, (hipStream_t)stream, g);
    PIT_CHECK_LAUNCH();
    return 0;
}

#if PIT_EDGE_DBG & 0x100
extern "C" int pit_edge_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pit_edge_stamps), sizeof(unsigned long long) * 64);
}
#endif
