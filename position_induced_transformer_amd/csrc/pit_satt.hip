// Dense self-attention of the processor (pit.py:37-57 with locality 1.0: every key kept) on bf16 MFMA (round 6, bf16 math mode).
//
// The processor layers of the mid-sized configurations - Elasticity 972 points x hid 256 x 2 heads per sample, NACA 728 x 128 x 1,
// Vorticity 256 x 256 x 2 - ran on the fp32-era kernels with their operands rounded in registers (legacy v_mfma_f32_32x32x8_bf16,
// every value element converted by every workgroup that touches it, keys split over eight waves and reduced through LDS):
// 46 + 145 us per Elasticity layer in bf16 mode, 48 % of the step, at 15-47 % MFMA busy.  Here:
//   * the values are rounded to bf16 ONCE per layer (satt_prep_*: X16; the backward's G_h = dO_h / rowsum_h likewise) and reach the
//     matrix pipe as [key][column] images through ds_read_b64_tr_b16;
//   * a wavefront owns 16 rows x ALL columns (16 accumulator tiles at hid 256): the softmax weights exp(-c m) of its rows - formed
//     in registers directly in the A-fragment layout of v_mfma_f32_16x16x32_bf16, 8 keys per lane and step, from key coordinates
//     staged once per workgroup - are amortised over the whole row, no cross-wave reduction, both heads against the same value tile;
//   * forward (MODE 0): O_h = (E_h X) / rowsum_h, rowsum and mbar = sum_j P m from the fp32 weights - and the ROUNDED weights stay behind
//     as A-fragment tiles ([mesh sample][head][16-row tile][32-key step][lane][8] bf16: pit_satt_tiles_elems);
//   * backward d(values) (MODE 1): E is symmetric, so d(values) = residual + sum_h E_h G_h is the same contraction with both heads
//     accumulating into ONE set of tiles - on the forward's tiles (PRE): 16 bytes per lane and step instead of eight weights formed, and
//     what is left of a step is its B tiles arriving from L2 (128-row workgroups, RT = 2, halve that stream);
//   * d(scale) (MODE 2): -(sum G_h . (E_h (m - mbar_h)) X), operand = tile x (m - mbar) (PRE) or formed from scratch, fp64 partial sums
//     into the layer's slots.
// The values' bf16 copy (x16) and G16 = bf16(dO_h / rowsum_h) come from the MLP chains either side of the layer (pit_chain.hip) or, for a
// first layer, from satt_prep_kernel.  Distances, c, rowsum, mbar and the d(scale) reduction stay fp32 / fp64; only MFMA operands are
// bf16.  The fp32 mode never comes here.
#include "pit_common.h"
#include "pit_block_dev.h"
#include "pit_gemm_rd.h"
#include <cstdlib>

namespace {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int KC = 32;       // keys per step (one v_mfma_f32_16x16x32_bf16 k)
constexpr int CGS = KC + 1;  // key rows per column group of a B tile in LDS: 33 x 32 B - with 32 the 16-byte stores that park a tile
                             // (a wavefront = 2 keys x 32 column pieces) hit every column group at the same banks: 16-way conflicts
#ifndef PIT_SATT_DBG
#define PIT_SATT_DBG 0       // diagnostic builds (tools/variant_build.sh): 1 no tile loads / parks in the loop, 2 constant weights, 4 no
                             // fragment reads / MFMAs, 8 no barrier in the loop, 16 no epilogue, 32 two steps only (results are void,
                             // times are not)
#endif

struct SattArgs {
    const float* mesh; int mesh_batch, L, sdim, used; float period;
    int batch, dim, n_head, tiles;             // tiles = ceil(L / 64) row tiles per sample
    const float* head; int head_is_scale;
    const unsigned short* b16;                 // MODE 0 / 2: X16 (batch, L, dim); MODE 1: G16 (batch, H, L, dim)
    float* out; long ld_out, out_bstride; int out_col0;
    float* rowstat; float* scale_out;          // MODE 0 writes (mesh_batch, H, L, 4) = {T, S_min, 1/rowsum, mbar}
    const float* rowstat_r;                    // MODE 1 / 2
    const float* d_out; long ld_dout, dout_bstride;
    float* d_values; long ld_dv, dv_bstride; int add_residual;
    double* dscale;
    unsigned short* e_out;                     // MODE 0, optional: the rounded weights as A-fragment tiles (satt_tiles_elems)
    const unsigned short* e_in;                // MODE 1, PRE: those tiles
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void* p) { return make_rsrc(p, 0x7ffffff0u); }
constexpr unsigned OOB = 0x7ffffff8u;

// B fragment of column group ct from a tile stored [column group of 16][key 0..31 (+1 pad row)][16 columns].  The k order inside a 32-key step is PERMUTED - lane quarter kq holds keys 4 kq + 0..3 (element 0..3) and
// 16 + 4 kq + 0..3 (element 4..7); the weights (A operand) are formed for the same keys - so that the two lane quarters of a
// half-wave read 8 CONSECUTIVE keys = 256 contiguous bytes per ds_read_b64_tr_b16: conflict-free.  (With the natural order,
// keys 8 kq + j, the quarters read rows 0-3 and 8-11 of any padded row-major image: 2-way conflicts at best - every padding was
// tried on paper - on the reads that carry most of the kernel's LDS traffic.)
__device__ __forceinline__ v8s_t frag_tr(const unsigned short* t, int ct, int l15, int kq) {
    const unsigned short* a0 = t + (ct * CGS + 4 * kq + (l15 >> 2)) * 16 + 4 * (l15 & 3);
    typedef v4s_t __attribute__((address_space(3))) * lds_v4;
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0 + 16 * 16));
    v8s_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
__device__ __forceinline__ f32x4_t mma(v8s_t a, v8s_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, a), __builtin_bit_cast(v8bf_t, b), c, 0, 0, 0);
}

// 32 keys x W columns (from column col0 on) of a bf16 row-major tensor of DIM columns (rows beyond `nrows` load zeros):
// W * 32 / 8 / NT 16-byte pieces per thread
template <int W, int NT = 256> struct Tile { u32x4_t v[W * KC / 8 / NT]; };
template <int W, int DIM, int NT = 256>
__device__ __forceinline__ void tile_load(const unsigned short* src, long nrows, int k0, int tid, Tile<W, NT>& t) {
    const __amdgpu_buffer_rsrc_t r = wide_rsrc(src);
#pragma unroll
    for (int u = 0; u < W * KC / 8 / NT; ++u) {
        const int e = tid + NT * u, k = e / (W / 8), c = e % (W / 8);
        const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((k0 + k) < nrows ? (unsigned)((((long)(k0 + k)) * DIM + 8 * c) * 2) : OOB), 0, 0);
        t.v[u] = u32x4_t{(unsigned)q.x, (unsigned)q.y, (unsigned)q.z, (unsigned)q.w};
    }
}
template <int W, int NT = 256>
__device__ __forceinline__ void tile_park(unsigned short* dst, int tid, const Tile<W, NT>& t) {
#pragma unroll
    for (int u = 0; u < W * KC / 8 / NT; ++u) {
        const int e = tid + NT * u, k = e / (W / 8), c = e % (W / 8);                // 8 columns 8 c .. of key k
        *reinterpret_cast<u32x4_t*>(dst + ((c >> 1) * CGS + k) * 16 + 8 * (c & 1)) = t.v[u];
    }
}

// MODE 0: out; MODE 1: d(values); MODE 2: d(scale).  MODE 0 / 2 run ONE head per workgroup (their accumulators are per head: 64
// registers instead of 128, two or three workgroups per CU instead of one), MODE 1 both (the heads add into one set of tiles) - and
// with two heads HALF the columns: both heads' full-width tiles are 84 KB of LDS = one workgroup per CU and, at Elasticity's 160
// row tiles, 160 of 256 CUs; the halves re-form the weights (vector ALU that the idle CUs had to spare).
constexpr int satt_width(int H, int DIM, int MODE) { return (MODE == 1 && H == 2) ? DIM / 2 : DIM; }

// RT: 64-row tiles per workgroup (NT = 256 RT threads).  Two for the PRE d(values) of two heads x hid 256: with the weights read instead
// of formed, what is left of a step is the B-tile stream - 24 KB per workgroup and step from L2, and the CUs that held two 64-row
// workgroups pulled 53 GB/s, a CU's practical L2 rate: 0.9 us per step.  128 rows per workgroup = half the workgroups = half the bytes.
template <int H, int DIM, int MODE, bool PERIODIC, bool PRE = false, int RT = 1>
__device__ __forceinline__ void satt_body(const SattArgs& g, const int bid) {
    static_assert(!PRE || MODE != 0, "the backward reads the forward's weight tiles");
    constexpr int NT = 256 * RT;
    constexpr bool PRE1 = PRE && MODE == 1;    // d(values): the tiles ARE the A operand
    constexpr bool PRE2 = PRE && MODE == 2;    // d(scale): A = tile x (m - mbar) - the distance again, but no exp2, no padding select
    constexpr int W = satt_width(H, DIM, MODE), NCS = DIM / W, NCT = W / 16, TSZ = CGS * W, HW = (MODE == 1) ? H : 1, NB = (MODE == 1) ? H : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* keys = reinterpret_cast<float4*>(smem_raw);                        // [Lp] key coordinates (zero-padded to 3)
    const int Lp = (g.L + 2 * KC - 1) / (2 * KC) * (2 * KC);       // whole rounds of two steps: no conditional step (its loads would cost the counted waits)
    unsigned short* tb = reinterpret_cast<unsigned short*>(keys + Lp + KC);    // [2][NB][TSZ]  (keys: one step of zeros beyond Lp - the
                                                                               // weights of step s + 1 are formed in step s, unconditionally)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    int b, t;
    const int hsel = (MODE == 1) ? 0 : bid % H;                                // this workgroup's head (MODE 0 / 2)
    // (consecutive ids - dealt to the XCDs in turn - walk the tiles of a sample: with a sample pinned to one XCD, ten samples put two
    // on XCDs 0 and 1 and one workgroup per CU made that a second round)
    const int cs = (MODE == 1) ? bid % NCS : 0;                                // this workgroup's column half (MODE 1, two heads)
    if (!slab_of_linear((MODE == 1) ? bid / NCS : bid / H, g.batch, (g.tiles + RT - 1) / RT, b, t)) return;
    const int mb = g.mesh_batch == 1 ? 0 : b;
    const int row = t * 64 * RT + wave * 16 + l15;                             // the A-fragment row of this lane
    const int rowc = row < g.L ? row : g.L - 1;
    const float* mesh = g.mesh + (long)mb * g.L * g.sdim;
    const unsigned short* bsrc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) bsrc[q] = g.b16 + ((long)b * NB + q) * g.L * DIM + cs * W;
    constexpr int NR = PRE1 ? 4 : 2;            // register sets of B tiles in flight: a PRE step is ~0.3 us, two steps of lead are less than an L2 round trip
    Tile<W, NT> reg[NR][NB];
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
        for (int q = 0; q < NB; ++q) tile_load<W, DIM, NT>(bsrc[q], g.L, KC * s, tid, reg[s][q]);
    if (!PRE1)
    for (int j = tid; j < Lp + KC; j += NT) {
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < g.L) {
            p.x = mesh[(long)j * g.sdim];
            if (g.used > 1) p.y = mesh[(long)j * g.sdim + 1];
            if (g.used > 2) p.z = mesh[(long)j * g.sdim + 2];
        }
        keys[j] = p;
    }
    const float rx = PRE1 ? 0.0f : mesh[(long)rowc * g.sdim], ry = (!PRE1 && g.used > 1) ? mesh[(long)rowc * g.sdim + 1] : 0.0f,
                rz = (!PRE1 && g.used > 2) ? mesh[(long)rowc * g.sdim + 2] : 0.0f;
    float c[HW], c2[HW], mbar[HW];                                             // c2 = -c log2(e): exp(-c m) = exp2(c2 m), one multiply less per weight
#pragma unroll
    for (int h = 0; h < HW; ++h) {
        const int hh = hsel + h;
        c[h] = g.head_is_scale ? g.head[hh] : head_scale_from_lmda(g.head[hh]);
        c2[h] = -1.4426950408889634f * c[h];
        mbar[h] = (MODE == 2) ? g.rowstat_r[(((long)mb * H + hh) * g.L + rowc) * 4 + 3] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) tile_park<W, NT>(tb + q * TSZ, tid, reg[0][q]);
#pragma unroll
    for (int q = 0; q < NB; ++q) tile_load<W, DIM, NT>(bsrc[q], g.L, NR * KC, tid, reg[0][q]);
    f32x4_t acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float rs = 0.0f, sm = 0.0f;                                               // MODE 0: row sum, sum of weight x distance
    // the weights of step s_ in the A-fragment layout: element e of lane quarter kq = key 32 s + 16 (e >> 2) + 4 kq + (e & 3)
#define PIT_SATT_KEYS(s_, kp_)                                                                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                   \
        const int j = KC * (s_) + 16 * (e >> 2) + 4 * kq + (e & 3);                                                   \
        kp_[e] = keys[j];                                                                                             \
    }
#define PIT_SATT_WEIGHTS(s_, kp_, dst_, ring_)                                                                            \
    do {                                                                                                              \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                               \
            const int j = KC * (s_) + 16 * (e >> 2) + 4 * kq + (e & 3);                                               \
            const float m = sq_dist3t<PERIODIC>(rx, ry, rz, kp_[e].x, kp_[e].y, kp_[e].z, g.period);                  \
            _Pragma("unroll") for (int h = 0; h < HW; ++h) {                                                          \
                float ev;                                                                                             \
                if (PRE2) ev = __uint_as_float((e & 1) ? (er[ring_][e >> 1] & 0xffff0000u) : (er[ring_][e >> 1] << 16));      \
                else ev = (j < g.L) ? __builtin_amdgcn_exp2f(__fmul_rn(m, c2[h])) : 0.0f;                             \
                if (MODE == 0) { rs += ev; sm = fmaf(ev, m, sm); }                                                    \
                if (MODE == 2) ev *= (m - mbar[h]);                                                                   \
                dst_[h][e] = bf16_bits(ev);                                                                           \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    const int nsteps = (PIT_SATT_DBG & 32) ? 2 : Lp / KC;
    // MODE 0 leaves its rounded weights behind as A-fragment tiles [mesh sample][head][16-row tile][step][lane][8] (1 KB per wavefront
    // and step, contiguous): d(values) contracts the same (symmetric) matrix and reads them instead of forming every weight again
    // (twice, with its column halves).  Unconditional buffer stores: a NULL e_out is a resource of size 0 (every store dropped),
    // the step beyond the last and the repeated samples of a batch-free mesh go to an out-of-range offset.
    const long tile_base = (((long)(mb * H + hsel) * (g.tiles * 4) + t * 4 * RT + wave) * nsteps) * 64 + lane;     // in 16-byte units
    const __amdgpu_buffer_rsrc_t re = make_rsrc(g.e_out, (MODE == 0 && g.e_out) ? 0x7ffffff0u : 0u);
    const bool e_mine = g.mesh_batch > 1 || b == 0;
#define PIT_SATT_ESTORE(s_, frag_)                                                                                    \
    do { if (MODE == 0) {                                                                                             \
        const u32x4_t w_ = __builtin_bit_cast(u32x4_t, frag_);                                                       \
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)w_.x, (int)w_.y, (int)w_.z, (int)w_.w}, re,               \
                                               (int)((e_mine && (s_) < nsteps) ? (unsigned)((tile_base + (long)(s_) * 64) * 16) : OOB), 0, 0); \
    } } while (0)
    v8s_t af[2][HW];
    // PRE2: the forward's tile of step s in er[s & 3], requested four steps ahead (16 bytes per lane: the lane's eight weights)
    const u32x4_t* ap2 = reinterpret_cast<const u32x4_t*>(g.e_in) + (PRE2 ? (((long)(mb * H + hsel) * (g.tiles * 4) + min(t * 4 * RT + wave, g.tiles * 4 - 1)) * nsteps) * 64 + lane : 0);
    u32x4_t er[4];
    if constexpr (PRE2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) er[s] = ap2[(long)min(s, nsteps - 1) * 64];
    }
    if constexpr (!PRE1) {
        float4 kp0[8];
        __syncthreads();                                                      // (the key coordinates are in LDS)
        PIT_SATT_KEYS(0, kp0);
        PIT_SATT_WEIGHTS(0, kp0, af[0], 0);
        PIT_SATT_ESTORE(0, af[0][0]);
    }
    // step s contracts keys [32 s, 32 s + 32): its B tile(s) sit in buffer s & 1; register set (s + 1) & 1 holds tile s + 1 (parked
    // now, into the buffer step s - 1 just left) and is re-loaded with tile s + 3.  LDS reads are ISSUED FIRST - the next step's key
    // coordinates, then the B fragments one group of column tiles ahead of their MFMAs - and the next step's weights (vector ALU,
    // independent of this step's MFMAs) are formed while they arrive.  (First version: every key read and every fragment read
    // directly in front of its use = sixteen exposed LDS round trips per step: 1.4 us per step at NACA's 8 MFMAs.)
    constexpr int GS = 4 / HW, NG = NCT / GS;                                  // column tiles per fragment group (registers: two groups in flight)
#define PIT_SATT_FRAGS(tile_, g_, dst_)                                                                               \
    _Pragma("unroll") for (int u = 0; u < GS; ++u)                                                                    \
        _Pragma("unroll") for (int h = 0; h < HW; ++h) dst_[u][h] = frag_tr(tile_ + h * TSZ, (g_) * GS + u, l15, kq);
#define PIT_SATT_STEP(s_, j_)                                                                                          \
    do {                                                                                                              \
        if (!(PIT_SATT_DBG & 8)) __syncthreads();                                                                     \
        if (!(PIT_SATT_DBG & 1)) {                                                                                    \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) tile_park<W, NT>(tb + ((((s_) + 1) & 1) * NB + q) * TSZ, tid, reg[((j_) + 1) & 1][q]); \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) tile_load<W, DIM, NT>(bsrc[q], g.L, KC * ((s_) + 3), tid, reg[((j_) + 1) & 1][q]); \
        }                                                                                                             \
        const unsigned short* tile_ = tb + (((s_) & 1) * NB) * TSZ;                                                   \
        float4 kp_[8];                                                                                                \
        v8s_t f0_[GS][HW], f1_[GS][HW];                                                                               \
        if (!(PIT_SATT_DBG & 2)) { PIT_SATT_KEYS((s_) + 1, kp_); }                                                    \
        if (!(PIT_SATT_DBG & 4)) { PIT_SATT_FRAGS(tile_, 0, f0_); }                                                   \
        if (PRE2) er[(j_) & 3] = ap2[(long)min((s_) + 4, nsteps - 1) * 64];                                            \
        if (!(PIT_SATT_DBG & 2)) { PIT_SATT_WEIGHTS((s_) + 1, kp_, af[((j_) + 1) & 1], ((j_) + 1) & 3); PIT_SATT_ESTORE((s_) + 1, af[((j_) + 1) & 1][0]); } \
        else { _Pragma("unroll") for (int h = 0; h < HW; ++h) af[((j_) + 1) & 1][h] = af[(j_) & 1][h]; }              \
        if (!(PIT_SATT_DBG & 4))                                                                                      \
        _Pragma("unroll") for (int gi = 0; gi < NG; gi += 2) {                                                        \
            if (gi + 1 < NG) { PIT_SATT_FRAGS(tile_, gi + 1, f1_); }                                                  \
            _Pragma("unroll") for (int u = 0; u < GS; ++u)                                                            \
                _Pragma("unroll") for (int h = 0; h < HW; ++h) acc[gi * GS + u] = mma(af[(j_) & 1][h], f0_[u][h], acc[gi * GS + u]); \
            if (gi + 2 < NG) { PIT_SATT_FRAGS(tile_, gi + 2, f0_); }                                                  \
            if (gi + 1 < NG) {                                                                                        \
                _Pragma("unroll") for (int u = 0; u < GS; ++u)                                                        \
                    _Pragma("unroll") for (int h = 0; h < HW; ++h) acc[(gi + 1) * GS + u] = mma(af[(j_) & 1][h], f1_[u][h], acc[(gi + 1) * GS + u]); \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    // PRE (d(values) on the forward's tiles): the A fragment of a step is ONE 16-byte load per lane and head, requested three steps
    // ahead into a ring of four register sets; no key coordinates, no vector ALU but the address
#define PIT_SATT_STEP_PRE(s_, j_)                                                                                      \
    do {                                                                                                              \
        __syncthreads();                                                                                              \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) tile_park<W, NT>(tb + ((((s_) + 1) & 1) * NB + q) * TSZ, tid, reg[((j_) + 1) & 3][q]); \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) tile_load<W, DIM, NT>(bsrc[q], g.L, KC * ((s_) + 5), tid, reg[((j_) + 1) & 3][q]); \
        _Pragma("unroll") for (int h = 0; h < HW; ++h) ar[((j_) + 3) & 3][h] = ap[h][(long)min((s_) + 3, nsteps - 1) * 64]; \
        const unsigned short* tile_ = tb + (((s_) & 1) * NB) * TSZ;                                                   \
        v8s_t f0_[GS][HW], f1_[GS][HW];                                                                               \
        PIT_SATT_FRAGS(tile_, 0, f0_);                                                                                \
        _Pragma("unroll") for (int gi = 0; gi < NG; gi += 2) {                                                        \
            if (gi + 1 < NG) { PIT_SATT_FRAGS(tile_, gi + 1, f1_); }                                                  \
            _Pragma("unroll") for (int u = 0; u < GS; ++u)                                                            \
                _Pragma("unroll") for (int h = 0; h < HW; ++h) acc[gi * GS + u] = mma(__builtin_bit_cast(v8s_t, ar[(j_) & 3][h]), f0_[u][h], acc[gi * GS + u]); \
            if (gi + 2 < NG) { PIT_SATT_FRAGS(tile_, gi + 2, f0_); }                                                  \
            if (gi + 1 < NG) {                                                                                        \
                _Pragma("unroll") for (int u = 0; u < GS; ++u)                                                        \
                    _Pragma("unroll") for (int h = 0; h < HW; ++h) acc[(gi + 1) * GS + u] = mma(__builtin_bit_cast(v8s_t, ar[(j_) & 3][h]), f1_[u][h], acc[(gi + 1) * GS + u]); \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    if constexpr (PRE1) {
        const u32x4_t* ap[HW];
#pragma unroll
        for (int h = 0; h < HW; ++h)
            ap[h] = reinterpret_cast<const u32x4_t*>(g.e_in) + (((long)(mb * H + h) * (g.tiles * 4) + min(t * 4 * RT + wave, g.tiles * 4 - 1)) * nsteps) * 64 + lane;
        u32x4_t ar[4][HW];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int h = 0; h < HW; ++h) ar[s][h] = ap[h][(long)min(s, nsteps - 1) * 64];
        int sb = 0;
        for (; sb + 4 <= nsteps; sb += 4) {
            PIT_SATT_STEP_PRE(sb, 0);
            PIT_SATT_STEP_PRE(sb + 1, 1);
            PIT_SATT_STEP_PRE(sb + 2, 2);
            PIT_SATT_STEP_PRE(sb + 3, 3);
        }
        if (sb < nsteps) {                                                    // (the step count is even)
            PIT_SATT_STEP_PRE(sb, 0);
            PIT_SATT_STEP_PRE(sb + 1, 1);
        }
    } else if constexpr (PRE2) {
        int sb = 0;
        for (; sb + 4 <= nsteps; sb += 4) {
            PIT_SATT_STEP(sb, 0);
            PIT_SATT_STEP(sb + 1, 1);
            PIT_SATT_STEP(sb + 2, 2);
            PIT_SATT_STEP(sb + 3, 3);
        }
        if (sb < nsteps) {
            PIT_SATT_STEP(sb, 0);
            PIT_SATT_STEP(sb + 1, 1);
        }
    } else {
        for (int sb = 0; sb < nsteps; sb += 2) {
            PIT_SATT_STEP(sb, 0);
            PIT_SATT_STEP(sb + 1, 1);
        }
    }
#undef PIT_SATT_STEP_PRE
#undef PIT_SATT_ESTORE
#undef PIT_SATT_STEP
#undef PIT_SATT_FRAGS
#undef PIT_SATT_WEIGHTS
#undef PIT_SATT_KEYS
    // (the weights of the step beyond the last were formed too: keys >= L give zeros, rs / sm are unchanged by them)
    // ---- epilogues.  Accumulator register i of this lane is row 4 kq + i of the wave's 16, column 16 ct + l15; the per-row
    // quantities live on the lanes whose l15 is the row (summed over the four key quarters)
    const int r0 = t * 64 * RT + wave * 16;
    if ((PIT_SATT_DBG & 16) && acc[0][0] != 123.456f) return;
    if (MODE == 0) {
        rs += __shfl_xor(rs, 16, 64); rs += __shfl_xor(rs, 32, 64);
        sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
        const float inv = 1.0f / rs;
        if (kq == 0 && row < g.L && (g.mesh_batch > 1 || b == 0)) {
            float4 st;
            st.x = 3.0e38f; st.y = 0.0f; st.z = inv; st.w = sm * inv;
            *reinterpret_cast<float4*>(g.rowstat + (((long)mb * H + hsel) * g.L + row) * 4) = st;
        }
        if (bid < H && tid == 0 && g.scale_out) g.scale_out[hsel] = c[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float iv = __shfl(inv, 4 * kq + i, 64);
            const int n = r0 + 4 * kq + i;
            if (n < g.L) {
                float* dst = g.out + (long)b * g.out_bstride + (long)n * g.ld_out + g.out_col0 + hsel * DIM + l15;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) dst[16 * ct] = acc[ct][i] * iv;
            }
        }
    } else if (MODE == 1) {
        // (the residual through a buffer resource, every load requested before the first add: as `add_residual ? res[..] : 0` it was a
        // branch, a load and s_waitcnt vmcnt(0) per element - 32 dependent round trips, 9 us of the launch)
        const __amdgpu_buffer_rsrc_t rres = wide_rsrc(g.d_out);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = r0 + 4 * kq + i;
            float rv[NCT];
            const unsigned roff = (unsigned)(((long)b * g.dout_bstride + (long)n * g.ld_dout + cs * W + l15) * 4);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) rv[ct] = buf_load(rres, (g.add_residual && n < g.L) ? roff + 64u * ct : OOB);
            if (n < g.L) {
                float* dst = g.d_values + (long)b * g.dv_bstride + (long)n * g.ld_dv + cs * W + l15;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) dst[16 * ct] = acc[ct][i] + rv[ct];
            }
        }
    } else {
        double part = 0.0;
        const float inv_l = g.rowstat_r[(((long)mb * H + hsel) * g.L + rowc) * 4 + 2];
        const __amdgpu_buffer_rsrc_t rd = wide_rsrc(g.d_out);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float iv = __shfl(inv_l, 4 * kq + i, 64);
            const int n = r0 + 4 * kq + i;
            float sd = 0.0f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float dv = buf_load(rd, n < g.L ? (unsigned)(((long)b * g.dout_bstride + (long)n * g.ld_dout + g.out_col0 + hsel * DIM + 16 * ct + l15) * 4) : OOB);
                sd = fmaf(acc[ct][i], dv, sd);
            }
            part += (double)sd * (double)iv;
        }
        part = wave_sum_d(part);
        __shared__ double wred[NT / 64];
        if (lane == 0) wred[wave] = part;
        __syncthreads();
        if (tid == 0)
        {
            double tot = 0.0;
            for (int w = 0; w < NT / 64; ++w) tot += wred[w];
            atomicAdd(g.dscale + (long)hsel * PIT_DSCALE_SLOTS + ((bid / H) & (PIT_DSCALE_SLOTS - 1)), -tot);
        }
    }
}

template <int H, int DIM, int MODE, bool PERIODIC, bool PRE = false, int RT = 1>
__global__ __launch_bounds__(256 * RT, 2) void satt_kernel(SattArgs g) {
    satt_body<H, DIM, MODE, PERIODIC, PRE, RT>(g, (int)blockIdx.x);
}
// the backward of a layer in ONE launch (both on the forward's tiles): workgroups [0, n1) are d(values), the rest d(scale) - the two
// are independent, a few hundred latency-bound workgroups each, and merged they share the chip instead of running back to back
// ... and, behind them, the weight-gradient reductions of the MLP whose backward produced this layer's d_out (`rider`: tiles of
// gemm_rr_tile; they depend on that MLP's backward only, and used to be a launch of their own in front of this one)
template <int H, int DIM, bool PERIODIC, int RT>
__global__ __launch_bounds__(256 * RT, 2) void satt_bwd_kernel(SattArgs g1, SattArgs g2, int n1, int n_att, pit_detail::DwPair w) {
    const int bid = (int)blockIdx.x;
    if (bid >= n_att) { rr_rider_pair(w, bid - n_att, pit_dyn_smem()); return; }
    if (bid < n1) satt_body<H, DIM, 1, false, true, RT>(g1, bid);
    else satt_body<H, DIM, 2, PERIODIC, true, RT>(g2, bid - n1);
}

// X16 = bf16(values[:, :, 0:dim]) (+ the concat's copy of the inputs), G16_h = bf16(d_out_h / rowsum_h): elementwise, 4 columns per thread
struct PrepArgs {
    const float* src; long ld, bstride; int col0, batch, L, dim, n_head, mesh_batch;
    unsigned short* dst; const float* rowstat;
    float* copy_dst; long copy_ld, copy_bstride;
};
__global__ __launch_bounds__(256) void satt_prep_kernel(PrepArgs g, int bwd) {
    const int q4 = g.dim / 4;
    const long total = (long)g.batch * (bwd ? g.n_head : 1) * g.L * q4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % q4);
        long r = i / q4;
        const int n = (int)(r % g.L); r /= g.L;
        const int h = bwd ? (int)(r % g.n_head) : 0;
        const int b = (int)(bwd ? r / g.n_head : r);
        float4 v = *reinterpret_cast<const float4*>(g.src + (long)b * g.bstride + (long)n * g.ld + g.col0 + (bwd ? h * g.dim : 0) + 4 * c);
        if (bwd) {
            const float iv = g.rowstat[(((long)(g.mesh_batch == 1 ? 0 : b) * g.n_head + h) * g.L + n) * 4 + 2];
            v.x *= iv; v.y *= iv; v.z *= iv; v.w *= iv;
        } else if (g.copy_dst) {
            *reinterpret_cast<float4*>(g.copy_dst + (long)b * g.copy_bstride + (long)n * g.copy_ld + 4 * c) = v;
        }
        uint2 pk;
        pk.x = (unsigned)f_to_bf16(v.x) | ((unsigned)f_to_bf16(v.y) << 16);
        pk.y = (unsigned)f_to_bf16(v.z) | ((unsigned)f_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2*>(g.dst + i * 4) = pk;
    }
}

size_t satt_smem(int L, int width, int nb) { return (size_t)((L + 2 * KC - 1) / (2 * KC) * (2 * KC) + KC) * 16 + (size_t)2 * nb * CGS * width * 2; }
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int H, int DIM, int MODE>
void launch_satt(const SattArgs& g, int periodic, hipStream_t s) {
    const int per_tile = MODE == 1 ? DIM / satt_width(H, DIM, MODE) : H;
    const size_t sm = satt_smem(g.L, satt_width(H, DIM, MODE), MODE == 1 ? H : 1);
#define PIT_SATT_GO(K_, RT_) do {                                                                                      \
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(K_), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), true); \
        (void)once;                                                                                                   \
        hipLaunchKernelGGL(K_, dim3((unsigned)(g.batch * ((g.tiles + (RT_) - 1) / (RT_)) * per_tile)), dim3(256 * (RT_)), sm, s, g); \
    } while (0)
    if constexpr (MODE == 1) {
        if (g.e_in) {
            if constexpr (H == 2 && DIM == 256) PIT_SATT_GO((satt_kernel<H, DIM, 1, false, true, 2>), 2);
            else PIT_SATT_GO((satt_kernel<H, DIM, 1, false, true, 1>), 1);
            return;
        }
    }
    if constexpr (MODE == 2) {
        if (g.e_in) {
            if constexpr (H == 2 && DIM == 256) {
                if (!periodic && (long)g.batch * g.tiles * per_tile > 256) { PIT_SATT_GO((satt_kernel<H, DIM, 2, false, true, 2>), 2); return; }
            }
            if (periodic) PIT_SATT_GO((satt_kernel<H, DIM, 2, true, true, 1>), 1);
            else PIT_SATT_GO((satt_kernel<H, DIM, 2, false, true, 1>), 1);
            return;
        }
    }
    // (128-row workgroups when the 64-row tiles outnumber the CUs: half the workgroups pull half the bytes from L2 - Elasticity's forward
    // 44.8 -> 39.1 us)
    if constexpr (H == 2 && DIM == 256) {
        if (!periodic && (long)g.batch * g.tiles * per_tile > 256) { PIT_SATT_GO((satt_kernel<H, DIM, MODE, false, false, 2>), 2); return; }
    }
    if (periodic) PIT_SATT_GO((satt_kernel<H, DIM, MODE, true>), 1);
    else PIT_SATT_GO((satt_kernel<H, DIM, MODE, false>), 1);
#undef PIT_SATT_GO
}
template <int H, int DIM>
void launch_satt_bwd(const SattArgs& g1, const SattArgs& g2, int periodic, const pit_detail::DwPair& w, hipStream_t s) {
    constexpr int RT = (H == 2 && DIM == 256) ? 2 : 1;
    const int tiles = (g1.tiles + RT - 1) / RT;
    const int n1 = g1.batch * tiles * (DIM / satt_width(H, DIM, 1)), n2 = g1.batch * tiles * H, n_att = n1 + n2, n_r = w.n1 + w.n2;
    const size_t sm = std::max(std::max(satt_smem(g1.L, satt_width(H, DIM, 1), H), satt_smem(g1.L, DIM, 1)), (size_t)(n_r ? 65536 : 0));
#define PIT_SATT_GO2(K_) do {                                                                                          \
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(K_), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), true); \
        (void)once;                                                                                                   \
        hipLaunchKernelGGL(K_, dim3((unsigned)(n_att + n_r)), dim3(256 * RT), sm, s, g1, g2, n1, n_att, w);            \
    } while (0)
    if (periodic) PIT_SATT_GO2((satt_bwd_kernel<H, DIM, true, RT>));
    else PIT_SATT_GO2((satt_bwd_kernel<H, DIM, false, RT>));
#undef PIT_SATT_GO2
}
void dispatch_satt_bwd(const SattArgs& g1, const SattArgs& g2, int periodic, const pit_detail::DwPair& w, hipStream_t s) {
    if (g1.n_head == 1) { if (g1.dim == 128) launch_satt_bwd<1, 128>(g1, g2, periodic, w, s); else launch_satt_bwd<1, 256>(g1, g2, periodic, w, s); }
    else { if (g1.dim == 128) launch_satt_bwd<2, 128>(g1, g2, periodic, w, s); else launch_satt_bwd<2, 256>(g1, g2, periodic, w, s); }
}
template <int MODE>
void dispatch_satt(const SattArgs& g, int periodic, hipStream_t s) {
    if (g.n_head == 1) { if (g.dim == 128) launch_satt<1, 128, MODE>(g, periodic, s); else launch_satt<1, 256, MODE>(g, periodic, s); }
    else { if (g.dim == 128) launch_satt<2, 128, MODE>(g, periodic, s); else launch_satt<2, 256, MODE>(g, periodic, s); }
}

int satt_fill(SattArgs& g, const float* mesh, int mesh_batch, int n_pts, int space_dim, int metric, float period, int batch, int n_head, int dim) {
    if (!mesh) return PIT_ERR_NULL;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    g = SattArgs();
    g.mesh = mesh; g.mesh_batch = mesh_batch; g.L = n_pts; g.sdim = space_dim; g.used = (metric == PIT_METRIC_PERIODIC1D) ? 1 : space_dim;
    g.period = period; g.batch = batch; g.dim = dim; g.n_head = n_head; g.tiles = (n_pts + 63) / 64;
    return 0;
}

}  // namespace

// 1 when pit_satt_fwd / _bwd cover a dense (locality 1.0) self-attention layer: 1-2 heads, hid 128 / 256, 64 <= points <= 2048 (the key
// coordinates of a sample stay in LDS), batch x points x dim below the 32-bit buffer-offset limit.  bf16 math mode only.
extern "C" int pit_satt_supported(int n_pts, int n_head, int dim, int batch, int mesh_batch) {
    if ((n_head != 1 && n_head != 2) || (dim != 128 && dim != 256) || n_pts < 64 || n_pts > 2048 || batch <= 0) return 0;
    if (mesh_batch != 1 && mesh_batch != batch) return 0;
    return (long)batch * n_pts * dim * (1 + n_head) * 4 < (1L << 31) - 65536;
}

// bf16 elements of the weight tiles of ONE (mesh sample, head): pit_satt_fwd's optional e_tiles is (mesh_batch, n_head, this)
extern "C" long pit_satt_tiles_elems(int n_pts) {
    const long tiles = (n_pts + 63) / 64;
    return tiles * 4 * (tiles * 64 / KC) * 512;
}

// out[b, n, out_col0 + h*dim + d] = sum_j softmax_j(-c_h m[n, j]) values[b, j, d]; copy_inputs: out[b, n, 0:dim] = values.  x16: scratch of
// batch*n_pts*dim bf16 (kept by the caller for pit_satt_bwd); rowstat (mesh_batch, n_head, n_pts, 4), scale_out (n_head) as pit_posatt_fwd.
extern "C" int pit_satt_fwd(const float* mesh, int mesh_batch, int n_pts, int space_dim, int metric, float period,
                            const float* values, long ld_values, long values_bstride, int batch, int dim,
                            const float* head, int n_head, int head_is_scale, unsigned short* x16,
                            float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                            float* rowstat, float* scale_out, unsigned short* e_tiles, int x16_ready, void* stream) {
    if (!values || !head || !x16 || !out || !rowstat) return PIT_ERR_NULL;
    if (x16_ready && copy_inputs) return PIT_ERR_UNSUPPORTED;                 // (the prep launch is also the concat's copy)
    if (e_tiles && !al16(e_tiles)) return PIT_ERR_SIZE;
    if (!pit_satt_supported(n_pts, n_head, dim, batch, mesh_batch) || space_dim < 1 || space_dim > 3) return PIT_ERR_UNSUPPORTED;
    if (ld_values % 4 || values_bstride % 4 || !al16(values) || !al16(x16) || !al16(rowstat) || (copy_inputs && (ld_out % 4 || out_bstride % 4 || !al16(out))))
        return PIT_ERR_SIZE;
    SattArgs g;
    if (int rc = satt_fill(g, mesh, mesh_batch, n_pts, space_dim, metric, period, batch, n_head, dim)) return rc;
    hipStream_t s = (hipStream_t)stream;
    PrepArgs p = PrepArgs();
    p.src = values; p.ld = ld_values; p.bstride = values_bstride; p.col0 = 0; p.batch = batch; p.L = n_pts; p.dim = dim; p.n_head = n_head;
    p.mesh_batch = mesh_batch; p.dst = x16;
    if (copy_inputs) { p.copy_dst = out; p.copy_ld = ld_out; p.copy_bstride = out_bstride; }
    const long total = (long)batch * n_pts * (dim / 4);
    if (!x16_ready) {
        hipLaunchKernelGGL(satt_prep_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, s, p, 0);
        PIT_CHECK_LAUNCH();
    }
    g.head = head; g.head_is_scale = head_is_scale; g.b16 = x16; g.out = out; g.ld_out = ld_out; g.out_bstride = out_bstride;
    g.out_col0 = out_col0; g.rowstat = rowstat; g.scale_out = scale_out; g.e_out = e_tiles;
    dispatch_satt<0>(g, metric != PIT_METRIC_EUCLID, s);
    PIT_CHECK_LAUNCH();
    return 0;
}

namespace {
int run_rider(const pit_mlp_params_job* r, void* stream) {      // (the launches pit_mlp_bwd_params would have made)
    return pit_mlp_bwd_params(r->x, r->ldx, r->rows, r->n0, r->n1, r->n2, r->h, r->out_gelu, r->d_y, r->ld_dy, r->d_w1, r->d_b1, r->d_w2,
                              r->d_b2, r->accumulate, const_cast<float*>(r->scratch), r->math_mode, stream);
}
}  // namespace

// Backward: d_values[b, j, :] = (add_residual ? d_out[b, j, 0:dim] : 0) + sum_h sum_n P_h[n, j] d_out[b, n, out_col0 + h*dim + :] (NULL: not
// needed) and the layer's d(scale) accumulators (PIT_HEAD_DEFER convention; NULL: not needed).  scale: the c of the forward; x16: the
// forward's; g16: scratch of batch*n_head*n_pts*dim bf16.
extern "C" int pit_satt_bwd(const float* mesh, int mesh_batch, int n_pts, int space_dim, int metric, float period,
                            int batch, int dim, const float* scale, int n_head, const float* rowstat,
                            const unsigned short* x16, unsigned short* g16,
                            const float* d_out, long ld_dout, long dout_bstride, int out_col0,
                            float* d_values, long ld_dvalues, long dvalues_bstride, int add_residual,
                            double* dscale, const unsigned short* e_tiles, int g16_ready, const pit_mlp_params_job* rider,
                            void* stream) {
    if (!scale || !rowstat || !x16 || !g16 || !d_out || (!d_values && !dscale)) return PIT_ERR_NULL;
    if (e_tiles && !al16(e_tiles)) return PIT_ERR_SIZE;
    if (!pit_satt_supported(n_pts, n_head, dim, batch, mesh_batch) || space_dim < 1 || space_dim > 3) return PIT_ERR_UNSUPPORTED;
    if (ld_dout % 4 || dout_bstride % 4 || out_col0 % 4 || !al16(d_out) || !al16(g16) || !al16(x16)) return PIT_ERR_SIZE;
    SattArgs g;
    if (int rc = satt_fill(g, mesh, mesh_batch, n_pts, space_dim, metric, period, batch, n_head, dim)) return rc;
    hipStream_t s = (hipStream_t)stream;
    g.head = scale; g.head_is_scale = 1; g.rowstat_r = rowstat; g.d_out = d_out; g.ld_dout = ld_dout; g.dout_bstride = dout_bstride;
    g.out_col0 = out_col0;
    static const bool split = getenv("PIT_SATT_SPLIT_BWD") != nullptr;          // (A/B: the two launches)
    const bool merged = d_values && dscale && e_tiles && !split;
    if (d_values) {
        PrepArgs p = PrepArgs();
        p.src = d_out; p.ld = ld_dout; p.bstride = dout_bstride; p.col0 = out_col0; p.batch = batch; p.L = n_pts; p.dim = dim; p.n_head = n_head;
        p.mesh_batch = mesh_batch; p.dst = g16; p.rowstat = rowstat;
        const long total = (long)batch * n_head * n_pts * (dim / 4);
        if (!g16_ready) {
            hipLaunchKernelGGL(satt_prep_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, s, p, 1);
            PIT_CHECK_LAUNCH();
        }
        g.b16 = g16; g.d_values = d_values; g.ld_dv = ld_dvalues; g.dv_bstride = dvalues_bstride; g.add_residual = add_residual;
        g.e_in = e_tiles;
        if (merged) {
            SattArgs g2 = g;
            g2.b16 = x16; g2.dscale = dscale;
            pit_detail::DwPair w = pit_detail::DwPair();
            // (carried where the launch's workgroups are four wavefronts: NACA 1.027 -> 1.007 ms; in the eight-wavefront launches of two heads
            // x hid 256 the tiles' workgroups idle half their waves and compete for the L2 stream: Elasticity 1.153 -> 1.189, Vorticity
            // 0.731 -> 0.742 - there the reductions stay a launch of their own)
            const bool riding = rider && !(n_head == 2 && dim == 256) && pit_detail::plan_rr_rider(*rider, &w, 256);
            if (!riding) w = pit_detail::DwPair();
            dispatch_satt_bwd(g, g2, metric != PIT_METRIC_EUCLID, w, s);
            PIT_CHECK_LAUNCH();
            return (rider && !riding) ? run_rider(rider, stream) : 0;
        }
        if (rider) { if (int rc = run_rider(rider, stream)) return rc; rider = nullptr; }
        dispatch_satt<1>(g, metric != PIT_METRIC_EUCLID, s);
        PIT_CHECK_LAUNCH();
    }
    if (rider) { if (int rc = run_rider(rider, stream)) return rc; }
    if (dscale) {
        g.b16 = x16; g.dscale = dscale; g.e_in = e_tiles;
        dispatch_satt<2>(g, metric != PIT_METRIC_EUCLID, s);
        PIT_CHECK_LAUNCH();
    }
    return 0;
}
