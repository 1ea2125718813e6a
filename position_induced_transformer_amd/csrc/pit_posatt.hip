// Fused position-attention kernels: distance -> head scale -> locality mask -> softmax
// -> weighted-value reduction, forward and backward, for gfx950.
//
// The reference materialises the (b,H,N,J) attention tensor and runs ~8 elementwise
// passes plus a row sort over it (pit.py:46-57, 133-144, 190-200, 248-258).  Here the
// weights live only in registers: a wavefront forms a 32(row) x 2(key) slab of
// P = exp(S_min - S)[S <= T] per v_mfma_f32_32x32x2_f32 and feeds it straight in as the
// A operand; the value rows are the B operand, read coalesced (32 consecutive channels
// per half-wave).  Thresholds T come from the order statistics of pit_select_fwd, the
// row maximum of the logits is -c*m_min, so no online-softmax rescale is needed and the
// sum is normalised once in the epilogue.
//
//   rows kernel  (MODE 0 forward, MODE 1 d(scale)): a workgroup owns 32 output rows of
//       one head and a group of 32*CT value columns; its W waves split the key range
//       and are reduced through LDS.  For batch-free meshes the batch is folded into
//       the column axis, so P is formed once for all samples.
//   cols kernel  (d values): a workgroup owns 32 KEYS and a column group; its waves
//       split the (head, row) range: dU[j,:] = sum_{h,n} P[h,n,j] dO[n,h,:].
//
// d(scale) uses  dc_h = -sum_{n,col} dO[n,col] * sum_j P[n,j] (m[n,j]-mbar_n) U[j,col]
// (SURVEY appendix B rearranged so that it is column-separable and centred:
// mbar_n = sum_j P m is saved by the forward), accumulated in fp64.
#include "pit_common.h"
#include "pit_block_dev.h"      // block_weights_body: the processor's weights as a rider of the down-projection launch
#include "pit_gemm_rd.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct AttArgs {
    const float* mesh_out; const float* mesh_in;
    int mesh_batch, n_out, n_in, sdim, periodic, coords_used;
    float period;
    const float* values; int batch, dim; long ld_values, values_bstride;
    const float* head; int n_head, head_is_scale;
    const float* stats; float rank_w; int masked;
    float* out; long ld_out, out_bstride; int out_col0, copy_inputs;
    float* rowstat; float* scale_out;
    const float* d_out; long ld_dout, dout_bstride;
    float* d_values; long ld_dvalues, dvalues_bstride; int add_residual;
    double* dscale_acc;
    float* d_head; const float* dhead_src; int dhead_is_scale, accumulate_head, nslots;   // d(scale) finish
    int ncols, colgroups, tiles_per_wg;
    int bf16;                             // PIT_MATH_BF16 for the forward and d(values) contractions
    unsigned values_bytes, dout_bytes;    // extents for the raw-buffer descriptors
    unsigned dim_magic;                   // floor(2^32 / dim): column -> (sample, channel) without an integer division
    int no_fast_loads;                    // PIT_NO_FAST_LOADS=1: checked loads everywhere (tests the >= 2 GiB path)
    int coord_dims;                       // > 0: value channels [0, coord_dims) are the key coordinates themselves (sparse kernels)
    int out16, dout16;                    // PIT_IO_OUT_BF16 / PIT_IO_DOUT_BF16: `out` / `d_out` hold bf16 elements (candidate-list kernels)
    unsigned dout_elems;                  // elements of d_out (dout_load's out-of-range offset)
    // round 4, large-regime kernels on PRECOMPUTED weights (pit_block_weights; batch-free self-attention, nothing masked):
    // pre_w (n_head, n_in, n_out) row-major over the CONTRACTED index - forward: E (symmetric), d(values): E, d(scale): Q
    const float* pre_w;
    int pre_swap;                         // d(scale) with the operands swapped: B rows = d_out head columns, the dot runs against values
};

// one element of d_out through its buffer descriptor: fp32, or bf16 widened exactly; offsets in ELEMENTS, `oob` = any
// element index beyond the tensor (the descriptor's range check then returns 0)
__device__ __forceinline__ float dout_load(__amdgpu_buffer_rsrc_t r, unsigned elem_off, int is16) {
    if (is16) return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)(elem_off * 2u), 0, 0) << 16);
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(elem_off * 4u), 0, 0));
}


// Folded column index -> (sample, channel).  For batch-free meshes the batch is folded into the
// column axis (col = sample*dim + channel); a runtime integer division costs ~35 VALU
// instructions on CDNA, this is a multiply-high plus one correction (exact for col < 2^31).
__device__ __forceinline__ void col_split(const AttArgs& a, int col, int mb, int& sample, int& chan) {
    if (a.mesh_batch != 1) { sample = mb; chan = col; return; }
    unsigned q = __umulhi((unsigned)col, a.dim_magic);
    unsigned r = (unsigned)col - q * (unsigned)a.dim;
    if (r >= (unsigned)a.dim) { ++q; r -= (unsigned)a.dim; }
    sample = (int)q; chan = (int)r;
}
// ceil(len / d) for a power-of-two d
__device__ __forceinline__ int ceil_div_pow2(int len, int d) { return (len + d - 1) >> (31 - __clz(d)); }

// d(scale): every wave adds its fp64 partial to one of `nslots` accumulators per head (spreading
// the atomics over addresses); a small finishing kernel (one workgroup per head) drains them with
// atomic exchanges - leaving them zero for the next call, so no memset is ever needed - applies
// d c / d lmda and writes or accumulates d_head.  (Folding the drain into the last workgroup of the
// main kernel via a ticket was measured slower: 0.479 vs 0.469 ms/step on Darcy b=8.)
__device__ __forceinline__ void dscale_add(double* slot, double v, bool doit) {
    if (doit) atomicAdd(slot, v);
}

__device__ __forceinline__ void dscale_drain_head(const AttArgs& a, int h, double* s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    double g = 0.0;
    for (int sl = threadIdx.x; sl < a.nslots; sl += blockDim.x) {      // only the slots this launch used
        unsigned long long* w = reinterpret_cast<unsigned long long*>(a.dscale_acc + (long)h * PIT_DSCALE_SLOTS + sl);
        g += __longlong_as_double((long long)atomicExch(w, 0ull));
    }
    g = wave_sum_d(g);
    __syncthreads();
    if (lane == 0) s_red[wave] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < nwaves; ++w) tot += s_red[w];
        if (!a.dhead_is_scale) {
            const float lm = a.dhead_src[h];
            tot *= head_scale_grad(lm, a.head_is_scale ? a.head[h] : head_scale_from_lmda(lm));
        }
        a.d_head[h] = a.accumulate_head ? a.d_head[h] + (float)tot : (float)tot;
    }
}

__global__ __launch_bounds__(256) void posatt_dhead_finish(AttArgs a) {
    __shared__ double s_red[10];
    dscale_drain_head(a, blockIdx.x, s_red);
}

// The same for several layers in ONE launch (pit_posatt_dhead_finish): the backward calls ran with
// PIT_HEAD_DEFER and left their accumulators loaded; one workgroup per (layer, head) drains all
// PIT_DSCALE_SLOTS slots.
constexpr int FINISH_MAX_LAYERS = 32;
struct FinishBatch {
    int n;
    double* ws[FINISH_MAX_LAYERS];
    float* d_head[FINISH_MAX_LAYERS];
    const float* head[FINISH_MAX_LAYERS];
    const float* scale[FINISH_MAX_LAYERS];
    int n_head[FINISH_MAX_LAYERS];
    int flags[FINISH_MAX_LAYERS];            // PIT_HEAD_ACCUMULATE | PIT_HEAD_IS_SCALE
    int wg_base[FINISH_MAX_LAYERS + 1];
};
__device__ __forceinline__ void finish_batch_body(const FinishBatch& fb, double* s_red) {
    int l = 0;
    while (l + 1 < fb.n && (int)blockIdx.x >= fb.wg_base[l + 1]) ++l;
    AttArgs a = AttArgs();
    a.dscale_acc = fb.ws[l];
    a.nslots = PIT_DSCALE_SLOTS;
    a.d_head = fb.d_head[l];
    a.dhead_src = fb.head[l];
    a.dhead_is_scale = (fb.flags[l] & PIT_HEAD_IS_SCALE) ? 1 : 0;
    a.accumulate_head = (fb.flags[l] & PIT_HEAD_ACCUMULATE) ? 1 : 0;
    a.head = fb.scale[l] ? fb.scale[l] : fb.head[l];
    a.head_is_scale = (fb.scale[l] || a.dhead_is_scale) ? 1 : 0;
    dscale_drain_head(a, (int)blockIdx.x - fb.wg_base[l], s_red);
}
__global__ __launch_bounds__(256) void posatt_dhead_finish_batch(FinishBatch fb) {
    __shared__ double s_red[10];
    finish_batch_body(fb, s_red);
}
// the same launch carrying a postponed MLP's weight-gradient reductions (round 5: the encoder MLP's, whose fused backward launch
// is the pass's last one - nothing else is left to carry them): workgroups [0, n_fin) drain, the rest reduce
__global__ __launch_bounds__(256) void posatt_dhead_finish_dw(FinishBatch fb, int n_fin, pit_detail::DwPair w) {
    __shared__ double s_red[10];
    if ((int)blockIdx.x < n_fin) { finish_batch_body(fb, s_red); return; }
    dw_pair_body(w, (int)blockIdx.x - n_fin, pit_dyn_smem());
}

constexpr int KEY_CHUNK = 2048;   // keys staged in LDS per pass (float4 each = 32 KiB)
constexpr int ROW_CHUNK = 1024;   // row records staged per pass in the cols kernel (32 KiB)

// Point `idx` of a contiguous (n, sdim) mesh as (x, y, z, 0): coordinates beyond `used` read
// as 0 through the buffer's range check (three independent loads, no branches).
__device__ __forceinline__ float4 load_point4(__amdgpu_buffer_rsrc_t r, unsigned bytes, long idx, int sdim, int used) {
    const unsigned base = (unsigned)(idx * sdim) * 4u;
    float4 v;
    v.x = buf_load(r, base);
    v.y = buf_load(r, used > 1 ? base + 4u : bytes);
    v.z = buf_load(r, used > 2 ? base + 8u : bytes);
    v.w = 0.0f;
    return v;
}

// Cross-wave reduction as a reduce-scatter through LDS: every wave parks its partial tile (and NX
// per-lane scalars), one barrier, then wave w sums and owns accumulator registers
// [w*16*CT/nwaves, ...) - so the epilogue (normalise, stores, residual loads) is spread over all
// waves instead of being serialised on wave 0.  Scalars are summed by every wave (all need them).
// `part` receives this wave's share (up to 16*CT registers, index q = t*16 + r).
#ifdef PIT_STAMPS
// diagnostic build only (tools/stamp_tiles.py): shader-clock stamps of one wave, never read by the kernels
__device__ unsigned long long pit_dbg_stamps[64];
#define PIT_STAMP(i_) do { if (blockIdx.x == 7 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) \
                               pit_dbg_stamps[i_] = __builtin_amdgcn_s_memtime(); } while (0)
#define PIT_STAMP_B(i_) do { if (bx == 1 && by == 0 && bz == 3 && threadIdx.x == 0) \
                                 pit_dbg_stamps[i_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PIT_STAMP(i_) do { } while (0)
#define PIT_STAMP_B(i_) do { } while (0)
#endif

// IL (interleaved column tiles, see posatt_rows_body): the parked order is register-major (q = i*CT + t), so that
// four consecutive q are one row's four adjacent columns.
template <int CT, int NX, bool IL = false>
__device__ __forceinline__ void park_tiles(const f32x16 (&acc)[CT], const float (&extra)[NX > 0 ? NX : 1], float* red,
                                           int wave, int lane) {
    constexpr int SLOT = (CT * 16 + NX) * 64;
    float* dst = red + (long)wave * SLOT + lane;
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(IL ? i * CT + t : t * 16 + i) * 64] = acc[t][i];
#pragma unroll
    for (int x = 0; x < NX; ++x) dst[(CT * 16 + x) * 64] = extra[x];
}
template <int CT, int NX>
__device__ __forceinline__ float summed(const float* red, int nwaves, int q, int lane) {
    constexpr int SLOT = (CT * 16 + NX) * 64;
    float v = 0.0f;
    for (int w = 0; w < nwaves; ++w) v += red[(long)w * SLOT + q * 64 + lane];
    return v;
}

// ------------------------------------------------------------------------------------
// rows kernel
// ------------------------------------------------------------------------------------
// IL ("interleaved", CT == 4 only): column tile t of a workgroup is the columns {4 l + t : l = 0..31} of its 128
// instead of [32 t, 32 t + 32): a lane's four B operands of a key are then ADJACENT in memory - one 16-B load per
// key instead of four 4-B loads (bare-loop measurements, tools/micro/mfma_mix.hip: a dword load per fp32 MFMA caps
// the matrix pipe at ~100 of 154 TF/s whatever the occupancy, a dwordx4 per four MFMAs at ~125) - and its four
// results of a row are adjacent too (16-B stores).  Needs 16-B aligned rows and dim % 4 == 0 (launch_rows checks).
template <int CT, int MODE, bool MASKED, bool BF, int NPX = 0, bool IL = false>
__device__ __forceinline__ void posatt_rows_body(const AttArgs& a, const int bx, const int by, const int bz) {
    static_assert(!IL || CT == 4, "interleaved tiles are the CT == 4 layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_xi = reinterpret_cast<float4*>(smem);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // bx = blockIdx.x = (sample, column group) so that workgroups dealt round-robin to the 8 XCDs
    // (id % 8) share value columns per XCD: each L2 then fetches only its own column slices
    const int n0 = bz * 32;
    const int h = by;
    const int mb = bx / a.colgroups, cg = bx % a.colgroups;
    const long rows_total = (long)a.mesh_batch * a.n_out;

    PIT_STAMP_B(32);
    const float c = a.head_is_scale ? a.head[h] : head_scale_from_lmda(a.head[h]);

    // ---- per-lane row constants (row = n0 + lane&31, both half-waves hold the same row)
    const int n = n0 + l31;
    const bool nvalid = n < a.n_out;
    const long rowid = (long)mb * a.n_out + (nvalid ? n : a.n_out - 1);
    const unsigned mo_bytes = (unsigned)((long)a.mesh_batch * a.n_out * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, mi_bytes);
    const float4 xo = load_point4(rmo, mo_bytes, rowid, a.sdim, a.coords_used);
    float T = __builtin_inff(), s_min = 0.0f, inv_l = 0.0f, mbar = 0.0f;
    if (MODE == 0) {
        if (MASKED) {
            const float sa = __fmul_rn(c, a.stats[rowid]);
            const float sb = __fmul_rn(c, a.stats[rows_total + rowid]);
            T = quantile_lerp(sa, sb, a.rank_w);
        }
        if (a.stats) s_min = __fmul_rn(c, a.stats[2 * rows_total + rowid]);
    } else {
        // backward: row constants saved by the forward {T, S_min, 1/rowsum, mbar}
        const float4 rs4 = *reinterpret_cast<const float4*>(
            a.rowstat + (((long)mb * a.n_head + h) * a.n_out + (nvalid ? n : a.n_out - 1)) * 4);
        T = rs4.x; s_min = rs4.y; inv_l = rs4.z; mbar = rs4.w;
    }

    // ---- per-lane column constants
    const __amdgpu_buffer_rsrc_t rvals = make_rsrc(a.values, a.values_bytes);
    const unsigned ld4 = (unsigned)a.ld_values * 4u;
    unsigned uoff[CT];                     // byte offset of this lane's column inside `values`
    bool cvalid[CT];
    int cb[CT], cd[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int col = IL ? cg * CT * 32 + l31 * CT + t : (cg * CT + t) * 32 + l31;
        cvalid[t] = col < a.ncols;
        const int cc = cvalid[t] ? col : 0;
        col_split(a, cc, mb, cb[t], cd[t]);
        uoff[t] = (unsigned)(((long)cb[t] * a.values_bstride + cd[t]) * 4);
    }

    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float rsum = 0.0f, qsum = 0.0f;
    const bool per = a.periodic != 0;
    constexpr bool bf = BF;                        // math mode (d(scale), MODE 1, is always exact fp32)
    int kpos[4];                                   // key position inside a group of 8 for slot u (math mode)
#pragma unroll
    for (int u = 0; u < 4; ++u) kpos[u] = group_pos(bf, u, half);

    // d(scale): the d_out tile this workgroup contracts with at the end; with few column tiles it is
    // fetched up front (16*CT registers) so its latency hides behind the whole key loop
    constexpr bool DOV_EARLY = (MODE == 1) && (CT <= 2);
    float dov[MODE == 1 ? CT : 1][16];
    auto load_dov = [&]() {
        const __amdgpu_buffer_rsrc_t rdo = make_rsrc(a.d_out, a.dout_bytes);
        if (IL && MODE == 1) {                             // a lane's four columns are adjacent: one 16-B load per row
            const unsigned cbase = (unsigned)(((long)cb[0] * a.dout_bstride + a.out_col0 + (long)h * a.dim + cd[0]) * 4);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int nr = n0 + acc_row(i, half);
                float q[4];
                buf_load4(rdo, (cvalid[0] && nr < a.n_out) ? cbase + (unsigned)nr * (unsigned)a.ld_dout * 4u : a.dout_bytes, q);
#pragma unroll
                for (int t = 0; t < (MODE == 1 ? CT : 0); ++t) dov[t][i] = q[t & 3];
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < (MODE == 1 ? CT : 0); ++t) {
            const unsigned cbase = (unsigned)(((long)cb[t] * a.dout_bstride + a.out_col0 + (long)h * a.dim + cd[t]) * 4);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int nr = n0 + acc_row(i, half);
                dov[t][i] = buf_load(rdo, (cvalid[t] && nr < a.n_out) ? cbase + (unsigned)nr * (unsigned)a.ld_dout * 4u
                                                                      : a.dout_bytes);
            }
        }
    };
    if (DOV_EARLY) load_dov();

    // Main loop.  A step covers 2*NP keys (NP per half-wave); the value rows of the NEXT step are in
    // flight while the current step forms its weights and issues its MFMAs (ping-pong register
    // buffers, two steps per trip).  The loop is issue-bound at PiT sizes (the weights cost ~15 VALU
    // instructions per element against one 64-cycle MFMA per CT elements), so everything that is
    // loop-invariant or wave-uniform is kept off the vector ALU:
    //   * value-row loads: per-lane offset = column + this half-wave's key (fixed), the row enters as
    //     the instruction's SCALAR offset - no per-load address arithmetic;
    //   * steps that are entirely inside the wave's key slice ("full": all but possibly the last)
    //     carry no range checks; only the tail step uses the checked variant;
    //   * the periodic wrap is a compile-time variant of the step, chosen once per launch;
    //   * rows beyond n_out evaluate the clamped duplicate of the last row (results never stored).
    constexpr int NP = NPX ? NPX : ((CT == 4) ? 4 : 8);
    constexpr int HS = BF ? 4 : 1;                       // key distance between the half-waves in a group of 8
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int hk = half * HS;
    const bool fast_ok = a.values_bytes < 0x80000000u && !a.no_fast_loads;   // scalar + vector offset cannot wrap
    unsigned voff[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) voff[t] = cvalid[t] ? uoff[t] + (unsigned)hk * ld4 : a.values_bytes;

    auto run = [&](auto per_tag) {
        constexpr bool PER = decltype(per_tag)::value;
        for (int jc0 = 0; jc0 < a.n_in; jc0 += KEY_CHUNK) {
            const int len = min(KEY_CHUNK, a.n_in - jc0);
            // this wave's slice of the chunk (multiple of 2*NP keys)
            const int per_wave = ceil_div_pow2(len, nwaves * 2 * NP) * 2 * NP;
            const int jb = wave_u * per_wave;
            const int je = min(len, jb + per_wave);
            const int nkeys = max(je - jb, 0);
            const int nfull = nkeys / (2 * NP);                       // steps without any range check
            const int nsteps = (nkeys + 2 * NP - 1) / (2 * NP);

            auto load_step = [&](float (&dst)[NP][CT], int st) {
                const int jj = jb + st * 2 * NP;
                if (st < nfull && fast_ok) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        constexpr int dummy = 0; (void)dummy;
                        const int koff = 8 * (u / 4) + (BF ? (u % 4) : 2 * (u % 4));
                        const int soff = (jc0 + jj + koff) * (int)ld4;  // wave-uniform: scalar offset operand
                        if (IL) {
                            const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rvals, (int)voff[0], soff, 0);
                            dst[u][0] = __int_as_float(q.x); dst[u][1 % CT] = __int_as_float(q.y);
                            dst[u][2 % CT] = __int_as_float(q.z); dst[u][3 % CT] = __int_as_float(q.w);
                            continue;
                        }
#pragma unroll
                        for (int t = 0; t < CT; ++t)
                            dst[u][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rvals, (int)voff[t], soff, 0));
                    }
                } else if (st < nsteps) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const int jl = jj + 8 * (u / 4) + kpos[u % 4];
                        const bool jv = jl < je;
                        const unsigned rowoff = (unsigned)(jc0 + jl) * ld4;
                        if (IL) {
                            float q[4];
                            buf_load4(rvals, (jv && cvalid[0]) ? uoff[0] + rowoff : a.values_bytes, q);
#pragma unroll
                            for (int t = 0; t < CT; ++t) dst[u][t] = q[t & 3];
                            continue;
                        }
#pragma unroll
                        for (int t = 0; t < CT; ++t)     // out-of-range offset -> hardware returns 0, no branch
                            dst[u][t] = buf_load(rvals, (jv && cvalid[t]) ? uoff[t] + rowoff : a.values_bytes);
                    }
                }
            };
            // weights of one step (VALU) ...
            auto weights = [&](float (&pw)[NP], bool (&anyk)[NP / 4], int st, auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                const int jj = jb + st * 2 * NP;
                const float4* xk = s_xi + hk;                          // this half-wave's keys: s_xi[jj + koff + hk]
#pragma unroll
                for (int q = 0; q < NP / 4; ++q) anyk[q] = false;
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int koff = 8 * (u / 4) + (BF ? (u % 4) : 2 * (u % 4));
                    bool jv = true;
                    float4 xi;
                    if (FULL) {
                        xi = xk[jj + koff];
                    } else {
                        const int jl = jj + koff + hk;
                        jv = jl < je;
                        xi = s_xi[jv ? jl : jb];
                    }
                    const float m = sq_dist3t<PER>(xo.x, xo.y, xo.z, xi.x, xi.y, xi.z, a.period);
                    const float sv = __fmul_rn(m, c);
                    bool keep = jv;
                    if (MASKED) keep = keep && (sv <= T);
                    float p = __expf(s_min - sv);
                    if (MASKED || !FULL) p = keep ? p : 0.0f;
                    if (MODE == 0) {
                        rsum += p;
                        qsum += p * m;
                        pw[u] = p;
                    } else {
                        pw[u] = p * (m - mbar) * inv_l;
                    }
                    if (MASKED) anyk[u / 4] |= keep;
                }
            };
            // ... and its contraction with the value rows (MFMA)
            auto contract = [&](const float (&pw)[NP], const bool (&anyk)[NP / 4], const float (&src)[NP][CT]) {
#pragma unroll
                for (int q = 0; q < NP / 4; ++q) {
                    if (MASKED && __builtin_amdgcn_ballot_w64(anyk[q]) == 0ull) continue;   // wave-uniform skip
                    if (bf) {
                        const bf16x4 ap = pack_bf16(pw[4 * q], pw[4 * q + 1], pw[4 * q + 2], pw[4 * q + 3]);
#pragma unroll
                        for (int t = 0; t < CT; ++t)
                            acc[t] = mfma_32x32x8_bf16(ap, pack_bf16(src[4 * q][t], src[4 * q + 1][t], src[4 * q + 2][t], src[4 * q + 3][t]), acc[t]);
                        continue;
                    }
#pragma unroll
                    for (int u = 4 * q; u < 4 * q + 4; ++u)
#pragma unroll
                        for (int t = 0; t < CT; ++t) acc[t] = mfma_32x32x2(pw[u], src[u][t], acc[t]);
                }
            };
            auto load_full = [&](float (&dst)[NP][CT], int st) {          // branch-free: steady state only
                const int jj = jb + st * 2 * NP;
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int koff = 8 * (u / 4) + (BF ? (u % 4) : 2 * (u % 4));
                    const int soff = (jc0 + jj + koff) * (int)ld4;
                    if (IL) {
                        const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rvals, (int)voff[0], soff, 0);
                        dst[u][0] = __int_as_float(q.x); dst[u][1 % CT] = __int_as_float(q.y);
                        dst[u][2 % CT] = __int_as_float(q.z); dst[u][3 % CT] = __int_as_float(q.w);
                        continue;
                    }
#pragma unroll
                    for (int t = 0; t < CT; ++t)
                        dst[u][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rvals, (int)voff[t], soff, 0));
                }
            };

            float b0[NP][CT], b1[NP][CT];
            float p0[NP], p1[NP];
            bool k0[NP / 4], k1[NP / 4];
            load_step(b0, 0);                                  // in flight during staging / barrier
            __syncthreads();
#pragma unroll 4
            for (int idx = threadIdx.x; idx < len; idx += blockDim.x)
                s_xi[idx] = load_point4(rmi, mi_bytes, (long)mb * a.n_in + jc0 + idx, a.sdim, a.coords_used);
            __syncthreads();
            // Software pipeline: while the MFMAs of step st run, the vector ALU already forms the weights
            // of step st+1 (independent work in ONE basic block, so the scheduler interleaves them) and the
            // value rows of step st+1 / st+2 are in flight.
            int st = 0;
            if (nfull > 0 && fast_ok) {
                weights(p0, k0, 0, std::true_type{});
                // two steps per trip with the register buffers swapping roles: as a one-step loop the ping-pong was
                // 20 register moves per step (a fifth of the loop's vector instructions, each in front of an MFMA
                // that reads the moved register)
                for (; CT >= 2 && st + 2 < nfull; st += 2) {    // (CT = 1: the latency regime, two or three steps per wave)
                    load_full(b1, st + 1);
                    weights(p1, k1, st + 1, std::true_type{});
                    contract(p0, k0, b0);
                    load_full(b0, st + 2);
                    weights(p0, k0, st + 2, std::true_type{});
                    contract(p1, k1, b1);
                }
                for (; st + 1 < nfull; ++st) {                 // steady state: steps st and st+1 are both full
                    load_full(b1, st + 1);
                    weights(p1, k1, st + 1, std::true_type{});
                    contract(p0, k0, b0);
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        p0[u] = p1[u];
#pragma unroll
                        for (int t = 0; t < CT; ++t) b0[u][t] = b1[u][t];
                    }
#pragma unroll
                    for (int q = 0; q < NP / 4; ++q) k0[q] = k1[q];
                }
                // p0 / b0 hold the last full step; then possibly the checked tail step
                load_step(b1, st + 1);
                contract(p0, k0, b0);
                st += 1;
                if (st < nsteps) { weights(p1, k1, st, std::false_type{}); contract(p1, k1, b1); }
            } else {                                           // short slices / >= 2 GiB tensors: plain loop
                for (; st < nsteps; ++st) {
#pragma unroll
                    for (int u = 0; u < NP; ++u)
#pragma unroll
                        for (int t = 0; t < CT; ++t) b1[u][t] = b0[u][t];
                    load_step(b0, st + 1);
                    if (st < nfull) weights(p0, k0, st, std::true_type{});
                    else weights(p0, k0, st, std::false_type{});
                    contract(p0, k0, b1);
                }
            }
        }
    };
    PIT_STAMP_B(33);
    if (per) run(std::true_type{}); else run(std::false_type{});
    PIT_STAMP_B(34);

    if (MODE == 1) {
        // dc_h -= sum acc[n,col] * dO[n,col]   (fp64 accumulation)
        if (!DOV_EARLY) load_dov();
        double part = 0.0;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) part += (double)acc[t][i] * (double)dov[t][i];
        part = wave_sum_d(part);
        // one fp64 atomic per workgroup (waves combine through LDS first), spread over
        // PIT_DSCALE_SLOTS accumulators per head to keep them off a single address
        double* wred = reinterpret_cast<double*>(smem);
        __syncthreads();                                   // staging region is free
        if (lane == 0) wred[wave] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int w = 0; w < nwaves; ++w) tot += wred[w];
            const int slot = (int)((bz + 131u * bx) & (a.nslots - 1));
            dscale_add(a.dscale_acc + h * PIT_DSCALE_SLOTS + slot, -tot, true);
        }
        return;
    }

    // ---- forward epilogue: reduce-scatter over the waves, normalise, store
    float extra[2];
    extra[0] = rsum + __shfl_xor(rsum, 32);
    extra[1] = qsum + __shfl_xor(qsum, 32);
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();                                   // staging region is free
    park_tiles<CT, 2, IL>(acc, extra, red, wave, lane);
    __syncthreads();
    PIT_STAMP_B(35);
    const float rs_tot = summed<CT, 2>(red, nwaves, CT * 16, lane);       // row = lane & 31
    const float qs_tot = summed<CT, 2>(red, nwaves, CT * 16 + 1, lane);
    const float inv = rs_tot > 0.0f ? 1.0f / rs_tot : 0.0f;
    PIT_STAMP_B(37);
    const int share = CT * 16 / nwaves;               // nwaves in {1,2,4,8}
    const int q0 = wave * share;
    // U accumulator registers per trip with ALL their LDS reads (U x nwaves) issued before the first add, and the
    // input-copy loads of the trip before its stores: one register per trip with a runtime wave loop was a chain
    // of dependent LDS round trips (measured with in-kernel stamps: 18.8 k of 105 k cycles at D = 256)
    auto finish = [&](auto u_tag, auto nw_tag, int qfirst) {
        constexpr int U = decltype(u_tag)::value, NW = decltype(nw_tag)::value;
        constexpr int SLOT = (CT * 16 + 2) * 64;
        float part[U][NW];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int w = 0; w < NW; ++w) part[u][w] = red[(long)w * SLOT + (qfirst + u) * 64 + lane];
        if constexpr (IL && U == 4) {
            // q = i*CT + t: the four registers of a trip are row i's four adjacent columns - 16-B accesses
            const int i = qfirst >> 2;
            const int nr = n0 + acc_row(i, half);
            const int col0 = cg * CT * 32 + l31 * CT;
            const bool cv = col0 < a.ncols;
            int bb, dd;
            col_split(a, cv ? col0 : 0, mb, bb, dd);
            const bool okr = cv && nr < a.n_out;
            float iv4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (a.copy_inputs && h == 0)                       // torch.cat((inputs, conv), -1), pit.py:44
                buf_load4(rvals, okr ? (unsigned)(((long)bb * a.values_bstride + dd) * 4) + (unsigned)nr * ld4 : a.values_bytes, iv4);
            const float rinv = __shfl(inv, acc_row(i, half));
            float4 v;
            float* vp = &v.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float sum = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += part[u][w];
                vp[u] = sum * rinv;
            }
            if (okr) {
                float* orow = a.out + (long)bb * a.out_bstride + (long)nr * a.ld_out + dd;
                *reinterpret_cast<float4*>(orow + a.out_col0 + (long)h * a.dim) = v;
                if (a.copy_inputs && h == 0) *reinterpret_cast<float4*>(orow) = make_float4(iv4[0], iv4[1], iv4[2], iv4[3]);
            }
            return;
        }
        long ooff[U];
        unsigned ioff[U];
        bool ok[U];
        float iv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = qfirst + u;
            const int t = IL ? q % CT : q >> 4, i = IL ? q / CT : q & 15;
            const int nr = n0 + acc_row(i, half);
            const int col = IL ? cg * CT * 32 + l31 * CT + t : (cg * CT + t) * 32 + l31;
            const bool cv = col < a.ncols;
            int bb, dd;
            col_split(a, cv ? col : 0, mb, bb, dd);
            ok[u] = cv && nr < a.n_out;
            ooff[u] = (long)bb * a.out_bstride + (long)nr * a.ld_out + dd;
            ioff[u] = (unsigned)(((long)bb * a.values_bstride + dd) * 4) + (unsigned)nr * ld4;
            iv[u] = 0.0f;
            if (a.copy_inputs && h == 0) iv[u] = buf_load(rvals, ok[u] ? ioff[u] : a.values_bytes);   // torch.cat((inputs, conv), -1), pit.py:44
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += part[u][w];
            v *= __shfl(inv, acc_row(IL ? (qfirst + u) / CT : (qfirst + u) & 15, half));
            if (ok[u]) a.out[ooff[u] + a.out_col0 + (long)h * a.dim] = v;
        }
        if (a.copy_inputs && h == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) a.out[ooff[u]] = iv[u];
        }
    };
    auto finish_all = [&](auto nw_tag) {
        int qq = 0;
        for (; qq + 4 <= share; qq += 4) { finish(std::integral_constant<int, 4>{}, nw_tag, q0 + qq); PIT_STAMP_B(38 + qq / 4); }
        for (; qq < share; ++qq) finish(std::integral_constant<int, 1>{}, nw_tag, q0 + qq);
    };
    switch (nwaves) {
        case 1: finish_all(std::integral_constant<int, 1>{}); break;
        case 2: finish_all(std::integral_constant<int, 2>{}); break;
        case 4: finish_all(std::integral_constant<int, 4>{}); break;
        default: finish_all(std::integral_constant<int, 8>{}); break;
    }
    if (wave == 0 && cg == 0 && half == 0 && nvalid) {
        float4 st; st.x = T; st.y = s_min; st.z = inv; st.w = qs_tot * inv;
        *reinterpret_cast<float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + n) * 4) = st;
    }
    PIT_STAMP_B(36);
    if (a.scale_out && wave == 0 && bx == 0 && bz == 0 && lane == 0) a.scale_out[h] = c;
}

template <int CT, int MODE, bool MASKED, bool BF, bool IL = false>
__global__ __launch_bounds__(512) void posatt_rows_kernel(AttArgs a) {
    posatt_rows_body<CT, MODE, MASKED, BF, 0, IL>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------------------------
// cols kernel: d values
// ------------------------------------------------------------------------------------
template <int CT, bool MASKED, bool BF, int NPX = 0, bool IL = false>
__device__ __forceinline__ void posatt_cols_body(const AttArgs& a, const int bx, const int by, const int bz) {
    static_assert(!IL || CT == 4, "interleaved tiles are the CT == 4 layout (see posatt_rows_body)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_rec = reinterpret_cast<float4*>(smem);                   // [ROW_CHUNK] {xo.xyz, T}
    float2* s_nrm = reinterpret_cast<float2*>(smem + ROW_CHUNK * sizeof(float4));   // [ROW_CHUNK] {S_min, 1/L}

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int j0 = by * 32;      // bx = blockIdx.x = column group: XCD-local reuse of the d_out columns
    const int cg = bx;
    const int mb = bz;

    const int j = j0 + l31;
    const bool jvalid = j < a.n_in;
    const unsigned mo_bytes = (unsigned)((long)a.mesh_batch * a.n_out * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, mi_bytes);
    const float4 xi = load_point4(rmi, mi_bytes, (long)mb * a.n_in + (jvalid ? j : a.n_in - 1), a.sdim, a.coords_used);

    const __amdgpu_buffer_rsrc_t rdout = make_rsrc(a.d_out, a.dout_bytes);
    const unsigned ldd4 = (unsigned)a.ld_dout * 4u;
    unsigned doff[CT];
    bool cvalid[CT];
    int cb[CT], cd[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int col = IL ? cg * CT * 32 + l31 * CT + t : (cg * CT + t) * 32 + l31;
        cvalid[t] = col < a.ncols;
        const int cc = cvalid[t] ? col : 0;
        col_split(a, cc, mb, cb[t], cd[t]);
        doff[t] = (unsigned)(((long)cb[t] * a.dout_bstride + a.out_col0 + cd[t]) * 4);
    }
    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    const bool per = a.periodic != 0;
    constexpr bool bf = BF;
    int kpos[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) kpos[u] = group_pos(bf, u, half);

    // Main loop: same organisation as the rows kernel (scalar-offset d_out loads, unchecked full
    // steps + checked tail, compile-time periodic variant, ping-pong prefetch); the reduced axis is
    // (head, row), the per-row constants {xo, T} / {S_min, 1/L} come from LDS.
    constexpr int NP = NPX ? NPX : ((CT == 4) ? 4 : 8);      // row pairs per step (NPX: override for the merged launch)
    constexpr int HS = BF ? 4 : 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int hk = half * HS;
    const bool fast_ok = a.dout_bytes < 0x80000000u && !a.no_fast_loads;
    unsigned voff[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) voff[t] = cvalid[t] ? doff[t] + (unsigned)hk * ldd4 : a.dout_bytes;

    auto run = [&](auto per_tag) {
        constexpr bool PER = decltype(per_tag)::value;
        for (int h = 0; h < a.n_head; ++h) {
            const float c = a.head_is_scale ? a.head[h] : head_scale_from_lmda(a.head[h]);
            const unsigned hoff4 = (unsigned)h * (unsigned)a.dim * 4u;
            for (int nc0 = 0; nc0 < a.n_out; nc0 += ROW_CHUNK) {
                const int len = min(ROW_CHUNK, a.n_out - nc0);
                const int per_wave = ceil_div_pow2(len, nwaves * 2 * NP) * 2 * NP;
                const int nb = wave_u * per_wave;
                const int ne = min(len, nb + per_wave);
                const int nrows = max(ne - nb, 0);
                const int nfull = nrows / (2 * NP);
                const int nsteps = (nrows + 2 * NP - 1) / (2 * NP);

                auto load_step = [&](float (&dst)[NP][CT], int st) {
                    const int nn = nb + st * 2 * NP;
                    if (st < nfull && fast_ok) {
#pragma unroll
                        for (int u = 0; u < NP; ++u) {
                            const int koff = 8 * (u / 4) + (BF ? (u % 4) : 2 * (u % 4));
                            const int soff = (nc0 + nn + koff) * (int)ldd4 + (int)hoff4;   // wave-uniform
                            if (IL) {
                                const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rdout, (int)voff[0], soff, 0);
                                dst[u][0] = __int_as_float(q.x); dst[u][1 % CT] = __int_as_float(q.y);
                                dst[u][2 % CT] = __int_as_float(q.z); dst[u][3 % CT] = __int_as_float(q.w);
                                continue;
                            }
#pragma unroll
                            for (int t = 0; t < CT; ++t)
                                dst[u][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdout, (int)voff[t], soff, 0));
                        }
                    } else if (st < nsteps) {
#pragma unroll
                        for (int u = 0; u < NP; ++u) {
                            const int nl = nn + 8 * (u / 4) + kpos[u % 4];
                            const bool nv = nl < ne;
                            const unsigned rowoff = (unsigned)(nc0 + nl) * ldd4 + hoff4;
                            if (IL) {
                                float q[4];
                                buf_load4(rdout, (nv && cvalid[0]) ? doff[0] + rowoff : a.dout_bytes, q);
#pragma unroll
                                for (int t = 0; t < CT; ++t) dst[u][t] = q[t & 3];
                                continue;
                            }
#pragma unroll
                            for (int t = 0; t < CT; ++t)
                                dst[u][t] = buf_load(rdout, (nv && cvalid[t]) ? doff[t] + rowoff : a.dout_bytes);
                        }
                    }
                };
                auto weights = [&](float (&pw)[NP], bool (&anyk)[NP / 4], int st, auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    const int nn = nb + st * 2 * NP;
                    const float4* rk = s_rec + hk;
                    const float2* nk = s_nrm + hk;
#pragma unroll
                    for (int q = 0; q < NP / 4; ++q) anyk[q] = false;
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const int koff = 8 * (u / 4) + (BF ? (u % 4) : 2 * (u % 4));
                        bool nv = true;
                        float4 r0;
                        float2 r1;
                        if (FULL) {
                            r0 = rk[nn + koff];
                            r1 = nk[nn + koff];
                        } else {
                            const int nl = nn + koff + hk;
                            nv = nl < ne;
                            r0 = s_rec[nv ? nl : nb];
                            r1 = s_nrm[nv ? nl : nb];
                        }
                        const float m = sq_dist3t<PER>(r0.x, r0.y, r0.z, xi.x, xi.y, xi.z, a.period);
                        const float sv = __fmul_rn(m, c);
                        bool keep = nv;
                        if (MASKED) keep = keep && (sv <= r0.w);
                        float p = __expf(r1.x - sv) * r1.y;
                        if (MASKED || !FULL) p = keep ? p : 0.0f;
                        pw[u] = p;
                        if (MASKED) anyk[u / 4] |= keep;
                    }
                };
                auto contract = [&](const float (&pw)[NP], const bool (&anyk)[NP / 4], const float (&src)[NP][CT]) {
#pragma unroll
                    for (int q = 0; q < NP / 4; ++q) {
                        if (MASKED && __builtin_amdgcn_ballot_w64(anyk[q]) == 0ull) continue;
                        if (bf) {
                            const bf16x4 ap = pack_bf16(pw[4 * q], pw[4 * q + 1], pw[4 * q + 2], pw[4 * q + 3]);
#pragma unroll
                            for (int t = 0; t < CT; ++t)
                                acc[t] = mfma_32x32x8_bf16(ap, pack_bf16(src[4 * q][t], src[4 * q + 1][t], src[4 * q + 2][t], src[4 * q + 3][t]), acc[t]);
                            continue;
                        }
#pragma unroll
                        for (int u = 4 * q; u < 4 * q + 4; ++u)
#pragma unroll
                            for (int t = 0; t < CT; ++t) acc[t] = mfma_32x32x2(pw[u], src[u][t], acc[t]);
                    }
                };
                auto load_full = [&](float (&dst)[NP][CT], int st) {          // branch-free: steady state only
                    const int nn = nb + st * 2 * NP;
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const int koff = 8 * (u / 4) + (BF ? (u % 4) : 2 * (u % 4));
                        const int soff = (nc0 + nn + koff) * (int)ldd4 + (int)hoff4;
                        if (IL) {
                            const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rdout, (int)voff[0], soff, 0);
                            dst[u][0] = __int_as_float(q.x); dst[u][1 % CT] = __int_as_float(q.y);
                            dst[u][2 % CT] = __int_as_float(q.z); dst[u][3 % CT] = __int_as_float(q.w);
                            continue;
                        }
#pragma unroll
                        for (int t = 0; t < CT; ++t)
                            dst[u][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdout, (int)voff[t], soff, 0));
                    }
                };

                float b0[NP][CT], b1[NP][CT];
                float p0[NP], p1[NP];
                bool k0[NP / 4], k1[NP / 4];
                load_step(b0, 0);
                __syncthreads();
#pragma unroll 2
                for (int idx = threadIdx.x; idx < len; idx += blockDim.x) {
                    const long rowid = (long)mb * a.n_out + nc0 + idx;
                    const float4 xo = load_point4(rmo, mo_bytes, rowid, a.sdim, a.coords_used);
                    const float4 rs4 = *reinterpret_cast<const float4*>(
                        a.rowstat + (((long)mb * a.n_head + h) * a.n_out + nc0 + idx) * 4);
                    float4 r0; r0.x = xo.x; r0.y = xo.y; r0.z = xo.z; r0.w = rs4.x;
                    s_rec[idx] = r0;
                    s_nrm[idx] = make_float2(rs4.y, rs4.z);
                }
                __syncthreads();
                int st = 0;
                if (NPX == 0 && nfull > 0 && fast_ok) {        // software pipeline as in the rows kernel (not in the
                                                               // register-capped merged launch)
                    weights(p0, k0, 0, std::true_type{});
                    for (; CT >= 2 && st + 2 < nfull; st += 2) {   // (two steps per trip: see the rows kernel)
                        load_full(b1, st + 1);
                        weights(p1, k1, st + 1, std::true_type{});
                        contract(p0, k0, b0);
                        load_full(b0, st + 2);
                        weights(p0, k0, st + 2, std::true_type{});
                        contract(p1, k1, b1);
                    }
                    for (; st + 1 < nfull; ++st) {
                        load_full(b1, st + 1);
                        weights(p1, k1, st + 1, std::true_type{});
                        contract(p0, k0, b0);
#pragma unroll
                        for (int u = 0; u < NP; ++u) {
                            p0[u] = p1[u];
#pragma unroll
                            for (int t = 0; t < CT; ++t) b0[u][t] = b1[u][t];
                        }
#pragma unroll
                        for (int q = 0; q < NP / 4; ++q) k0[q] = k1[q];
                    }
                    load_step(b1, st + 1);
                    contract(p0, k0, b0);
                    st += 1;
                    if (st < nsteps) { weights(p1, k1, st, std::false_type{}); contract(p1, k1, b1); }
                } else {
                    for (; st < nsteps; ++st) {
#pragma unroll
                        for (int u = 0; u < NP; ++u)
#pragma unroll
                            for (int t = 0; t < CT; ++t) b1[u][t] = b0[u][t];
                        load_step(b0, st + 1);
                        if (st < nfull) weights(p0, k0, st, std::true_type{});
                        else weights(p0, k0, st, std::false_type{});
                        contract(p0, k0, b1);
                    }
                }
            }
        }
    };
    if (per) run(std::true_type{}); else run(std::false_type{});

    float extra[1] = {0.0f};
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
    park_tiles<CT, 0, IL>(acc, extra, red, wave, lane);
    __syncthreads();
    const int share = CT * 16 / nwaves;
    const int q0 = wave * share;
    // four registers per trip: their LDS reads (4 x nwaves) and residual loads are all in flight before the first
    // add (one register per trip with a runtime wave loop was a chain of dependent LDS + memory round trips)
    auto finish = [&](auto u_tag, auto nw_tag, int qfirst) {
        constexpr int U = decltype(u_tag)::value, NW = decltype(nw_tag)::value;
        constexpr int SLOT = (CT * 16) * 64;
        float part[U][NW], res[U];
        long ooff[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int w = 0; w < NW; ++w) part[u][w] = red[(long)w * SLOT + (qfirst + u) * 64 + lane];
        if constexpr (IL && U == 4) {
            // q = i*CT + t: the four registers of a trip are key i's four adjacent columns - 16-B accesses
            const int i = qfirst >> 2;
            const int jr = j0 + acc_row(i, half);
            const int col0 = cg * CT * 32 + l31 * CT;
            const bool cv = col0 < a.ncols;
            int bb, dd;
            col_split(a, cv ? col0 : 0, mb, bb, dd);
            const bool okr = cv && jr < a.n_in;
            float r4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (a.add_residual)                                // self attention: d_out columns [0,dim) of the same row
                buf_load4(rdout, okr ? (unsigned)(((long)bb * a.dout_bstride + dd) * 4) + (unsigned)jr * ldd4 : a.dout_bytes, r4);
            float4 v;
            float* vp = &v.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float sum = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += part[u][w];
                vp[u] = sum + r4[u];
            }
            if (okr) *reinterpret_cast<float4*>(a.d_values + (long)bb * a.dvalues_bstride + (long)jr * a.ld_dvalues + dd) = v;
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = qfirst + u;
            const int t = IL ? q % CT : q >> 4, i = IL ? q / CT : q & 15;
            const int jr = j0 + acc_row(i, half);
            const int col = IL ? cg * CT * 32 + l31 * CT + t : (cg * CT + t) * 32 + l31;
            const bool cv = col < a.ncols;
            int bb, dd;
            col_split(a, cv ? col : 0, mb, bb, dd);
            ok[u] = cv && jr < a.n_in;
            // residual (self attention): d_out columns [0,dim) of the same row
            const unsigned roff = (unsigned)(((long)bb * a.dout_bstride + dd) * 4) + (unsigned)jr * ldd4;
            res[u] = buf_load(rdout, (a.add_residual && ok[u]) ? roff : a.dout_bytes);
            ooff[u] = (long)bb * a.dvalues_bstride + (long)jr * a.ld_dvalues + dd;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += part[u][w];
            if (ok[u]) a.d_values[ooff[u]] = v + res[u];
        }
    };
    auto finish_all = [&](auto nw_tag) {
        int qq = 0;
        for (; qq + 4 <= share; qq += 4) finish(std::integral_constant<int, 4>{}, nw_tag, q0 + qq);
        for (; qq < share; ++qq) finish(std::integral_constant<int, 1>{}, nw_tag, q0 + qq);
    };
    switch (nwaves) {
        case 1: finish_all(std::integral_constant<int, 1>{}); break;
        case 2: finish_all(std::integral_constant<int, 2>{}); break;
        case 4: finish_all(std::integral_constant<int, 4>{}); break;
        default: finish_all(std::integral_constant<int, 8>{}); break;
    }
}

template <int CT, bool MASKED, bool BF, bool IL = false>
__global__ __launch_bounds__(512) void posatt_cols_kernel(AttArgs a) {
    posatt_cols_body<CT, MASKED, BF, 0, IL>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// d(scale) and d(values) of one layer in ONE launch: the two are independent (both read d_out,
// the row constants and the values), each is a few hundred latency-bound workgroups, and merged
// they share the chip instead of running back to back.  Workgroups [0, n_cols_wgs) run the
// d(values) tiles (on the critical path of the backward pass: dispatched first), the rest the
// d(scale) tiles; `ar` / `ac` carry each part's own column grouping.
// (one column tile per workgroup only - the small, latency-bound regime; >= 4 waves per SIMD so that
// two 8-wave workgroups share a CU: all d(values) and d(scale) tiles are resident at once)
template <bool MASKED, bool BF>
__global__ __launch_bounds__(512, 4) void posatt_bwd_pair_kernel(AttArgs ar, AttArgs ac, int n_cols_wgs, int cgx, int cgy,
                                                                  int rgx, int rgy) {
    int id = blockIdx.x;
    if (id < n_cols_wgs) {
        posatt_cols_body<1, MASKED, BF, 4>(ac, id % cgx, (id / cgx) % cgy, id / (cgx * cgy));
    } else {
        id -= n_cols_wgs;
        posatt_rows_body<1, 1, MASKED, false, 4>(ar, id % rgx, (id / rgx) % rgy, id / (rgx * rgy));
    }
}

// ------------------------------------------------------------------------------------
// Large-regime kernels (>= 8 column tiles per row tile): weights in LDS, whole tiles per wave.
//
// The J-split kernels above recompute the 32 x J weight tile for every group of 128 columns and
// pay a cross-wave reduction plus a prologue per 128 columns - fine when the layer is tiny, the
// dominant cost at batch 256 (MFMA busy 24-30 %).  Here a workgroup (8 waves) forms the weight
// tile ONCE per 256-key chunk into LDS (k-major, so an A fragment is a conflict-free 32-lane
// read), and every wave owns up to 4 whole 32-column tiles of the output: no reduction between
// waves, the weights are reused by up to 32 column tiles, value rows are read once per tile.
constexpr int TK_CHUNK = 256;


// RT = 32-row tiles per workgroup: every value (B) fragment fetched from L2 feeds RT MFMAs, which
// is what bounds these kernels at scale (16*RT flop per operand byte).
template <int RT, int TPW, int MODE, bool MASKED, bool BF, bool PRE = false>
__global__ __launch_bounds__(512) void posatt_rows_tiles(AttArgs a) {
    static_assert(!(PRE && MASKED), "precomputed weights: unmasked layers only");
    constexpr int KC = TK_CHUNK / RT;                     // keys per LDS pass: Ps is always 32 KiB
    constexpr int GW = KC / 64;                           // 8-key groups filled per wave per pass
    __shared__ float Ps[KC * 32 * RT];
    __shared__ float4 s_xi[KC];
    __shared__ int s_flag[KC / 8];
    __shared__ float s_rs[2][16][32 * RT];
    PIT_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.z * 32 * RT;
    const int h = blockIdx.y;
    const int mb = blockIdx.x / a.colgroups, cs = blockIdx.x % a.colgroups;
    const long rows_total = (long)a.mesh_batch * a.n_out;
    const float c = PRE ? 0.0f : (a.head_is_scale ? a.head[h] : head_scale_from_lmda(a.head[h]));
    const unsigned mo_bytes = (unsigned)((long)a.mesh_batch * a.n_out * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, PRE ? 0u : mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, PRE ? 0u : mi_bytes);

    // per-lane row constants for each of the RT row tiles (row = n0 + rt*32 + lane&31)
    float4 xo[RT];
    float T[RT], s_min[RT], inv_l[RT], mbar[RT];
    bool nvalid[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int n = n0 + rt * 32 + l31;
        nvalid[rt] = n < a.n_out;
        const int nc = nvalid[rt] ? n : a.n_out - 1;
        const long rowid = (long)mb * a.n_out + nc;
        T[rt] = __builtin_inff(); s_min[rt] = 0.0f; inv_l[rt] = 0.0f; mbar[rt] = 0.0f;
        if (PRE) {                                          // (weights come from memory: only 1/rowsum of the forward's rows)
            xo[rt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 0) inv_l[rt] = a.rowstat[((long)h * a.n_out + nc) * 4 + 2];
            continue;
        }
        xo[rt] = load_point4(rmo, mo_bytes, rowid, a.sdim, a.coords_used);
        if (MODE == 0) {
            if (MASKED) T[rt] = quantile_lerp(__fmul_rn(c, a.stats[rowid]), __fmul_rn(c, a.stats[rows_total + rowid]), a.rank_w);
            if (a.stats) s_min[rt] = __fmul_rn(c, a.stats[2 * rows_total + rowid]);
        } else {
            const float4 rs4 = *reinterpret_cast<const float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + nc) * 4);
            T[rt] = rs4.x; s_min[rt] = rs4.y; inv_l[rt] = rs4.z; mbar[rt] = rs4.w;
        }
    }

    // this wave's column tiles: interleaved over the workgroup's tile range
    const __amdgpu_buffer_rsrc_t rvals = make_rsrc(a.values, a.values_bytes);
    const unsigned ld4 = (unsigned)a.ld_values * 4u;
    const int ntiles = (a.ncols + 31) / 32;
    const int tile_end = min(ntiles, (cs + 1) * a.tiles_per_wg);
    unsigned uoff[TPW];
    bool cvalid[TPW];
    int cb[TPW], cd[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = cs * a.tiles_per_wg + wave + 8 * t;
        const int col = tile * 32 + l31;
        cvalid[t] = tile < tile_end && col < a.ncols;
        const int cc = cvalid[t] ? col : 0;
        col_split(a, cc, mb, cb[t], cd[t]);
        uoff[t] = (unsigned)(((long)cb[t] * a.values_bstride + cd[t] + ((PRE && a.pre_swap) ? a.out_col0 + (long)h * a.dim : 0)) * 4);
    }
    f32x16 acc[RT][TPW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][t][i] = 0.0f;
    float rsum[RT], qsum[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { rsum[rt] = 0.0f; qsum[rt] = 0.0f; }
    const bool per = a.periodic != 0;
    constexpr bool bf = BF;
    int kpos[4];                                   // contraction-phase key position for slot u (math mode)
#pragma unroll
    for (int u = 0; u < 4; ++u) kpos[u] = group_pos(bf, u, half);

    // complete chunks: value rows through the instruction's SCALAR offset (row) + a fixed per-lane offset (column, this
    // half-wave's key inside a group) - no per-load address arithmetic or range select (as posatt_rows_body)
    const bool fast_ok = a.values_bytes < 0x80000000u && !a.no_fast_loads;
    unsigned voff[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) voff[t] = cvalid[t] ? uoff[t] + (unsigned)(half * (BF ? 4 : 1)) * ld4 : a.values_bytes;

    PIT_STAMP(1);
    // PRE: the weight tile of a pass = KC rows of pre_w (contracted index) x this workgroup's 32 RT columns (its output rows):
    // coalesced 128-B loads, requested one pass ahead
    const float* wt = PRE ? a.pre_w + (long)h * a.n_in * a.n_out : nullptr;
    float wnext[GW][4][RT];
    auto wload = [&](int c0n) {
#pragma unroll
        for (int it = 0; it < GW; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jl = (wave * GW + it) * 8 + 2 * u + half;
                const bool jv = c0n + jl < a.n_in;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    wnext[it][u][rt] = (jv && nvalid[rt]) ? wt[(long)(c0n + jl) * a.n_out + n0 + rt * 32 + l31] : 0.0f;
            }
    };
    if (PRE) wload(0);
    int pass_i = 0;
    for (int c0 = 0; c0 < a.n_in; c0 += KC, ++pass_i) {
        const int len = min(KC, a.n_in - c0);
        const int ngroups = (len + 7) / 8;
        float bnext[4][TPW];
        auto prefetch = [&](int g) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jl = g * 8 + kpos[u];
                const bool jv = g < ngroups && jl < len;
                const unsigned rowoff = (unsigned)(c0 + jl) * ld4;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    bnext[u][t] = buf_load(rvals, (jv && cvalid[t]) ? uoff[t] + rowoff : a.values_bytes);
            }
        };
        prefetch(0);                                          // value rows in flight during the weight phase
        __syncthreads();                                      // previous chunk fully consumed
        if (pass_i < 4) PIT_STAMP(2 + 4 * pass_i);
        if (!PRE) {
            for (int idx = tid; idx < len; idx += 512)
                s_xi[idx] = load_point4(rmi, mi_bytes, (long)mb * a.n_in + c0 + idx, a.sdim, a.coords_used);
            __syncthreads();
        }
        if (pass_i < 4) PIT_STAMP(3 + 4 * pass_i);
        // ---- weight phase: this wave fills keys [8*GW*wave, 8*GW*(wave+1)) of the chunk, all RT row tiles
        // (the periodic wrap is a compile-time variant: as a runtime flag it is if-converted into every distance)
        auto fill = [&](auto per_tag) {
            constexpr bool PER = decltype(per_tag)::value;
#pragma unroll
            for (int it = 0; it < GW; ++it) {
                const int g = wave * GW + it;
                bool anyk = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jl = g * 8 + 2 * u + half;
                    const bool jv = jl < len;
                    const float4 xi = s_xi[jv ? jl : 0];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float m = sq_dist3t<PER>(xo[rt].x, xo[rt].y, xo[rt].z, xi.x, xi.y, xi.z, a.period);
                        const float sv = __fmul_rn(m, c);
                        const bool keep = jv && nvalid[rt] && (sv <= T[rt]);
                        const float p = keep ? __expf(s_min[rt] - sv) : 0.0f;
                        float w = p;
                        if (MODE == 0) { rsum[rt] += p; qsum[rt] += p * m; }
                        else w = p * (m - mbar[rt]) * inv_l[rt];
                        Ps[(jl * RT + rt) * 32 + l31] = w;
                        anyk |= keep;
                    }
                }
                const bool flag = __builtin_amdgcn_ballot_w64(anyk) != 0ull;
                if (lane == 0) s_flag[g] = flag ? 1 : 0;
            }
        };
        if (PRE) {
#pragma unroll
            for (int it = 0; it < GW; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jl = (wave * GW + it) * 8 + 2 * u + half;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) Ps[(jl * RT + rt) * 32 + l31] = wnext[it][u][rt];
                }
            if (c0 + KC < a.n_in) wload(c0 + KC);               // the next pass's weights travel during this contraction
        } else if (per) fill(std::true_type{}); else fill(std::false_type{});
        if (pass_i < 4) PIT_STAMP(4 + 4 * pass_i);
        __syncthreads();
        if (pass_i < 4) PIT_STAMP(5 + 4 * pass_i);
        // ---- contraction phase: every wave walks the whole chunk for its own tiles
        auto contract = [&](int g, const float (&bcur)[4][TPW]) {
            if (MASKED && s_flag[g] == 0) return;
            if (bf) {
                bf16x4 bp[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) bp[t] = pack_bf16(bcur[0][t], bcur[1][t], bcur[2][t], bcur[3][t]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float af[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) af[u] = Ps[((g * 8 + kpos[u]) * RT + rt) * 32 + l31];
                    const bf16x4 ap = pack_bf16(af[0], af[1], af[2], af[3]);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) acc[rt][t] = mfma_32x32x8_bf16(ap, bp[t], acc[rt][t]);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float af[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) af[rt] = Ps[((g * 8 + 2 * u + half) * RT + rt) * 32 + l31];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) acc[rt][t] = mfma_32x32x2(af[rt], bcur[u][t], acc[rt][t]);
            }
        };
        if (len % 16 == 0 && fast_ok) {
            // (whole pairs of 8-key groups) two groups per trip, the register buffers swapping roles (no copies),
            // unchecked loads
            auto load_fast = [&](float (&dst)[4][TPW], int g) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int soff = (c0 + g * 8 + (BF ? u : 2 * u)) * (int)ld4;          // wave-uniform
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        dst[u][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rvals, (int)voff[t], soff, 0));
                }
            };
            for (int g = 0; g < ngroups; g += 2) {
                float b1[4][TPW];
                load_fast(b1, g + 1);
                contract(g, bnext);
                if (g + 2 < ngroups) load_fast(bnext, g + 2);
                contract(g + 1, b1);
            }
        } else {
            for (int g = 0; g < ngroups; ++g) {
                float bcur[4][TPW];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) bcur[u][t] = bnext[u][t];
                prefetch(g + 1);
                contract(g, bcur);
            }
        }
    }

    PIT_STAMP(20);
    if (MODE == 1) {
        const __amdgpu_buffer_rsrc_t rdo = make_rsrc(a.d_out, a.dout_bytes);
        double part = 0.0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const unsigned cbase = (unsigned)(((long)cb[t] * a.dout_bstride + ((PRE && a.pre_swap) ? 0 : a.out_col0 + (long)h * a.dim) + cd[t]) * 4);
                float dov[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int nr = n0 + rt * 32 + acc_row(i, half);
                    dov[i] = buf_load(rdo, (cvalid[t] && nr < a.n_out) ? cbase + (unsigned)nr * (unsigned)a.ld_dout * 4u
                                                                       : a.dout_bytes);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) part += (double)acc[rt][t][i] * (double)dov[i];
            }
        part = wave_sum_d(part);
        const int slot = (int)((blockIdx.z + 131u * blockIdx.x + 977u * wave) & (a.nslots - 1));
        dscale_add(a.dscale_acc + h * PIT_DSCALE_SLOTS + slot, -part, lane == 0);
        return;
    }
    // ---- forward epilogue: row sums over the 16 (wave, half) partials, normalise, store
    if (!PRE) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            s_rs[0][wave * 2 + half][rt * 32 + l31] = rsum[rt];
            s_rs[1][wave * 2 + half][rt * 32 + l31] = qsum[rt];
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float rs_tot = 0.0f, qs_tot = 0.0f;
        if (!PRE) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { rs_tot += s_rs[0][k][rt * 32 + l31]; qs_tot += s_rs[1][k][rt * 32 + l31]; }
        }
        const float inv = PRE ? inv_l[rt] : (rs_tot > 0.0f ? 1.0f / rs_tot : 0.0f);
        float inv_row[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) inv_row[i] = __shfl(inv, acc_row(i, half));
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            float* ocol = a.out + (long)cb[t] * a.out_bstride + a.out_col0 + (long)h * a.dim + cd[t];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int nr = n0 + rt * 32 + acc_row(i, half);
                if (cvalid[t] && nr < a.n_out) ocol[(long)nr * a.ld_out] = acc[rt][t][i] * inv_row[i];
            }
            if (a.copy_inputs && h == 0) {
                float iv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int nr = n0 + rt * 32 + acc_row(i, half);
                    iv[i] = buf_load(rvals, (cvalid[t] && nr < a.n_out) ? uoff[t] + (unsigned)nr * ld4 : a.values_bytes);
                }
                float* icol = a.out + (long)cb[t] * a.out_bstride + cd[t];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int nr = n0 + rt * 32 + acc_row(i, half);
                    if (cvalid[t] && nr < a.n_out) icol[(long)nr * a.ld_out] = iv[i];
                }
            }
        }
        if (!PRE && wave == 0 && cs == 0 && half == 0 && nvalid[rt]) {
            float4 st; st.x = T[rt]; st.y = s_min[rt]; st.z = inv; st.w = qs_tot * inv;
            *reinterpret_cast<float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + n0 + rt * 32 + l31) * 4) = st;
        }
    }
    PIT_STAMP(21);
    if (!PRE && a.scale_out && tid == 0 && blockIdx.x == 0 && blockIdx.z == 0) a.scale_out[h] = c;
}

template <int TPW, bool MASKED, bool BF, bool PRE = false>
__global__ __launch_bounds__(512) void posatt_cols_tiles(AttArgs a) {
    static_assert(!(PRE && MASKED), "precomputed weights: unmasked layers only");
    __shared__ float Ps[TK_CHUNK * 32];
    __shared__ float4 s_rec[TK_CHUNK * 2];
    __shared__ int s_flag[TK_CHUNK / 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int j0 = blockIdx.y * 32;
    const int cs = blockIdx.x;
    const int mb = blockIdx.z;
    const int j = j0 + l31;
    const bool jvalid = j < a.n_in;
    const unsigned mo_bytes = (unsigned)((long)a.mesh_batch * a.n_out * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, PRE ? 0u : mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, PRE ? 0u : mi_bytes);
    const float4 xi = PRE ? make_float4(0.f, 0.f, 0.f, 0.f)
                          : load_point4(rmi, mi_bytes, (long)mb * a.n_in + (jvalid ? j : a.n_in - 1), a.sdim, a.coords_used);
    const __amdgpu_buffer_rsrc_t rdout = make_rsrc(a.d_out, a.dout_bytes);
    const unsigned ldd4 = (unsigned)a.ld_dout * 4u;
    const int ntiles = (a.ncols + 31) / 32;
    const int tile_end = min(ntiles, (cs + 1) * a.tiles_per_wg);
    unsigned doff[TPW];
    bool cvalid[TPW];
    int cb[TPW], cd[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = cs * a.tiles_per_wg + wave + 8 * t;
        const int col = tile * 32 + l31;
        cvalid[t] = tile < tile_end && col < a.ncols;
        const int cc = cvalid[t] ? col : 0;
        col_split(a, cc, mb, cb[t], cd[t]);
        doff[t] = (unsigned)(((long)cb[t] * a.dout_bstride + a.out_col0 + cd[t]) * 4);
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    const bool per = a.periodic != 0;
    constexpr bool bf = BF;
    int kpos[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) kpos[u] = group_pos(bf, u, half);
    // complete chunks: d_out rows through the instruction's SCALAR offset (row) + a fixed per-lane offset (column, this
    // half-wave's row inside a group) - no per-load address arithmetic or range select (as posatt_cols_body)
    const bool fast_ok = a.dout_bytes < 0x80000000u && !a.no_fast_loads;
    unsigned voff[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) voff[t] = cvalid[t] ? doff[t] + (unsigned)(half * (BF ? 4 : 1)) * ldd4 : a.dout_bytes;

    // PRE: pre_w = E (n_head, n_out, n_in) row-major over the contracted row index; P[n][j] = E[n][j] / rowsum[n]
    float wnext[4][4];
    auto wload = [&](int h, int c0n) {
        const float* eh = a.pre_w + (long)h * a.n_out * a.n_in;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int nl = (wave * 4 + it) * 8 + 2 * u + half;
                wnext[it][u] = (c0n + nl < a.n_out && jvalid) ? eh[(long)(c0n + nl) * a.n_in + j] : 0.0f;
            }
    };
    if (PRE) wload(0, 0);
    for (int h = 0; h < a.n_head; ++h) {
        const float c = PRE ? 0.0f : (a.head_is_scale ? a.head[h] : head_scale_from_lmda(a.head[h]));
        const unsigned hoff = (unsigned)h * (unsigned)a.dim * 4u;
        for (int c0 = 0; c0 < a.n_out; c0 += TK_CHUNK) {
            const int len = min(TK_CHUNK, a.n_out - c0);
            const int ngroups = (len + 7) / 8;
            float bnext[4][TPW];
            auto prefetch = [&](int g) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int nl = g * 8 + kpos[u];
                    const bool nv = g < ngroups && nl < len;
                    const unsigned rowoff = (unsigned)(c0 + nl) * ldd4 + hoff;
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        bnext[u][t] = buf_load(rdout, (nv && cvalid[t]) ? doff[t] + rowoff : a.dout_bytes);
                }
            };
            prefetch(0);
            __syncthreads();
            for (int idx = tid; idx < len; idx += 512) {
                const long rowid = (long)mb * a.n_out + c0 + idx;
                const float4 xo = PRE ? make_float4(0.f, 0.f, 0.f, 0.f) : load_point4(rmo, mo_bytes, rowid, a.sdim, a.coords_used);
                const float4 rs4 = *reinterpret_cast<const float4*>(
                    a.rowstat + (((long)mb * a.n_head + h) * a.n_out + c0 + idx) * 4);
                float4 r0; r0.x = xo.x; r0.y = xo.y; r0.z = xo.z; r0.w = rs4.x;
                float4 r1; r1.x = rs4.y; r1.y = rs4.z; r1.z = 0.0f; r1.w = 0.0f;
                s_rec[2 * idx] = r0;
                s_rec[2 * idx + 1] = r1;
            }
            __syncthreads();
            auto fill = [&](auto per_tag) {
                constexpr bool PER = decltype(per_tag)::value;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int g = wave * 4 + it;
                    bool anyk = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int nl = g * 8 + 2 * u + half;
                        const bool nv = nl < len;
                        const float4 r0 = s_rec[2 * (nv ? nl : 0)];
                        const float4 r1 = s_rec[2 * (nv ? nl : 0) + 1];
                        const float m = sq_dist3t<PER>(r0.x, r0.y, r0.z, xi.x, xi.y, xi.z, a.period);
                        const float sv = __fmul_rn(m, c);
                        const bool keep = nv && jvalid && (sv <= r0.w);
                        Ps[nl * 32 + l31] = keep ? __expf(r1.x - sv) * r1.y : 0.0f;
                        anyk |= keep;
                    }
                    const bool flag = __builtin_amdgcn_ballot_w64(anyk) != 0ull;
                    if (lane == 0) s_flag[g] = flag ? 1 : 0;
                }
            };
            if (PRE) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int nl = (wave * 4 + it) * 8 + 2 * u + half;
                        Ps[nl * 32 + l31] = (nl < len) ? wnext[it][u] * s_rec[2 * nl + 1].y : 0.0f;       // E[n][j] / rowsum[n]
                    }
                // the next chunk's (or head's) weights travel during this contraction
                if (c0 + TK_CHUNK < a.n_out) wload(h, c0 + TK_CHUNK);
                else if (h + 1 < a.n_head) wload(h + 1, 0);
            } else if (per) fill(std::true_type{}); else fill(std::false_type{});
            __syncthreads();
            auto contract = [&](int g, const float (&bcur)[4][TPW]) {
                if (MASKED && s_flag[g] == 0) return;
                if (bf) {
                    float af[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) af[u] = Ps[(g * 8 + kpos[u]) * 32 + l31];
                    const bf16x4 ap = pack_bf16(af[0], af[1], af[2], af[3]);
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        acc[t] = mfma_32x32x8_bf16(ap, pack_bf16(bcur[0][t], bcur[1][t], bcur[2][t], bcur[3][t]), acc[t]);
                    return;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float af = Ps[(g * 8 + 2 * u + half) * 32 + l31];
#pragma unroll
                    for (int t = 0; t < TPW; ++t) acc[t] = mfma_32x32x2(af, bcur[u][t], acc[t]);
                }
            };
            if (len % 16 == 0 && fast_ok) {
                // (whole pairs of 8-row groups) two groups per trip, the register buffers swapping roles (no copies),
                // unchecked loads
                auto load_fast = [&](float (&dst)[4][TPW], int g) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int soff = (c0 + g * 8 + (BF ? u : 2 * u)) * (int)ldd4 + (int)hoff;   // wave-uniform
#pragma unroll
                        for (int t = 0; t < TPW; ++t)
                            dst[u][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdout, (int)voff[t], soff, 0));
                    }
                };
                for (int g = 0; g < ngroups; g += 2) {
                    float b1[4][TPW];
                    load_fast(b1, g + 1);
                    contract(g, bnext);
                    if (g + 2 < ngroups) load_fast(bnext, g + 2);
                    contract(g + 1, b1);
                }
            } else {
                for (int g = 0; g < ngroups; ++g) {
                    float bcur[4][TPW];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < TPW; ++t) bcur[u][t] = bnext[u][t];
                    prefetch(g + 1);
                    contract(g, bcur);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const unsigned rbase = (unsigned)(((long)cb[t] * a.dout_bstride + cd[t]) * 4);
        float rv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int jr = j0 + acc_row(i, half);
            rv[i] = buf_load(rdout, (a.add_residual && cvalid[t] && jr < a.n_in) ? rbase + (unsigned)jr * ldd4
                                                                                 : a.dout_bytes);
        }
        float* gcol = a.d_values + (long)cb[t] * a.dvalues_bstride + cd[t];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int jr = j0 + acc_row(i, half);
            if (cvalid[t] && jr < a.n_in) gcol[(long)jr * a.ld_dvalues] = acc[t][i] + rv[i];
        }
    }
}

// the same merge for 2 / 4 column tiles per workgroup (bigger layers: Elasticity, Vorticity), without
// the register cap - these bodies need their registers, one workgroup per CU
// (round 6: ... and, behind the attention workgroups, the weight-gradient reductions of the MLP that produced d_out as gemm_rr_tile
// tiles - `w` from plan_rr_rider; they used to be a launch of their own between that MLP's backward and this one, which does not
// depend on them; the first four waves of such a workgroup run the tile, in the math mode of the kernel)
template <int CT, bool MASKED, bool BF, bool IL = false>
__global__ __launch_bounds__(512) void posatt_bwd_pair_wide_kernel(AttArgs ar, AttArgs ac, int n_cols_wgs, int cgx, int cgy,
                                                                    int rgx, int rgy, int n_att, pit_detail::DwPair w) {
    int id = blockIdx.x;
    if (id >= n_att) {
        if (threadIdx.x >= 256) return;
        id -= n_att;
        const bool second = id >= w.n1;
        if (second) id -= w.n1;
        const GemmArgs& g = second ? w.g2 : w.g1;
        const int tx = second ? w.tx2 : w.tx1, tiles = second ? w.tiles2 : w.tiles1, slabs = second ? w.slabs2 : w.slabs1;
        const int slab = id / tiles, tile = id % tiles;
        const int kbeg = (int)((long)slab * w.nchunks / slabs) * pit_detail::RR_BK;
        const int kend = min(g.K, (int)((long)(slab + 1) * w.nchunks / slabs) * pit_detail::RR_BK);
        float* smem = pit_dyn_smem();
        gemm_rr_tile<1, 1, pit_detail::RR_BK, BF>(g, tile % tx, tile / tx, kbeg, kend, smem, smem + 2 * pit_detail::RR_BK * 64);
        return;
    }
    if (id < n_cols_wgs) {
        posatt_cols_body<CT, MASKED, BF, 0, IL>(ac, id % cgx, (id / cgx) % cgy, id / (cgx * cgy));
    } else {
        id -= n_cols_wgs;
        posatt_rows_body<CT, 1, MASKED, false, 0, IL>(ar, id % rgx, (id / rgx) % rgy, id / (rgx * rgy));
    }
}

// tuning overrides for experiments (tools/microbench.py): PIT_FORCE_CT, PIT_FORCE_WAVES, PIT_NO_TILES_KERNEL
int env_int(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

// tiles per workgroup for the large-regime kernels (0 = use the J-split kernels): they pay off
// only when the launch has real work (MFMA-instruction count), small layers stay latency-bound
int tiles_per_wg_for(int ncols, long units, long work_mfma, int mesh_batch) {
    if (env_int("PIT_NO_TILES_KERNEL")) return 0;
    const int ntiles = (ncols + 31) / 32;
    if (ntiles < 8) return 0;
    // measured (MI355X): wins for batch-free meshes with the batch folded into >= 32 column tiles
    // (Darcy b=256: 106 -> 55 us per processor layer); per-sample meshes with 8 tiles are faster
    // on the J-split kernels
    if (!env_int("PIT_FORCE_TILES") && (mesh_batch != 1 || ntiles < 32 || work_mfma < (1L << 19))) return 0;
    int tpw = 32;                                              // 4 tiles per wave
    while (tpw > 8 && (ntiles < tpw || units * ((ntiles + tpw - 1) / tpw) < 512)) tpw >>= 1;
    return tpw;
}

int pow2_floor(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }



// column-tile count per workgroup: least padding, then enough workgroups to cover the chip
int choose_ct(int ncols, long other_wgs) {
    int best = 4;
    long best_pad = -1;
    for (int ct = 4; ct >= 1; ct >>= 1) {
        const long pad = ((ncols + 32 * ct - 1) / (32 * ct)) * 32L * ct;
        if (best_pad < 0 || pad < best_pad) { best_pad = pad; best = ct; }
    }
    while (best > 1 && other_wgs * ((ncols + 32 * best - 1) / (32 * best)) < 256) best >>= 1;
    return best;
}

size_t rows_smem(int ct, int nwaves, int n_in) {
    const size_t stage = (size_t)min(KEY_CHUNK, n_in) * sizeof(float4);
    const size_t red = (size_t)nwaves * (ct * 16 + 2) * 64 * sizeof(float);
    return stage > red ? stage : red;
}
size_t cols_smem(int ct, int nwaves, int n_out) {
    (void)n_out;
    const size_t stage = (size_t)ROW_CHUNK * (sizeof(float4) + sizeof(float2));     // {xo,T} and {S_min,1/L} arrays
    const size_t red = (size_t)nwaves * (ct * 16) * 64 * sizeof(float);
    return stage > red ? stage : red;
}

// interleaved column tiles (16-B operand loads / result stores): every row a lane touches must be 16-B aligned and
// a lane's four columns must belong to one sample.  mode 0: forward (values, out), 1: d(scale) (values, d_out),
// 2: d(values) (d_out, d_values)
bool interleave_ok(const AttArgs& a, int mode) {
    static const bool off = getenv("PIT_NO_INTERLEAVE") != nullptr;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (off || a.dim % 4 != 0 || a.ncols % 4 != 0) return false;
    const bool vals = al(a.values) && a.ld_values % 4 == 0 && a.values_bstride % 4 == 0;
    const bool dout = al(a.d_out) && a.ld_dout % 4 == 0 && a.dout_bstride % 4 == 0 && a.out_col0 % 4 == 0;
    if (mode == 0) return vals && al(a.out) && a.ld_out % 4 == 0 && a.out_bstride % 4 == 0 && a.out_col0 % 4 == 0;
    if (mode == 1) return vals && dout;
    return dout && al(a.d_values) && a.ld_dvalues % 4 == 0 && a.dvalues_bstride % 4 == 0;
}

template <int MODE>
void launch_rows(const AttArgs& a0, hipStream_t s) {
    AttArgs a = a0;
    const bool bf = (MODE == 0) && a.bf16 != 0;
    const int n_tiles = (a.n_out + 31) / 32;
    const long work = (long)n_tiles * a.n_head * a.mesh_batch * ((a.ncols + 31) / 32) * ((a.n_in + 1) / 2);
    // row tiles per workgroup: more reuse of every value fragment, as long as the grid still fills the chip
    int rt = 4;
    while (rt > 1 && (n_tiles < rt || (long)((n_tiles + rt - 1) / rt) * a.n_head * a.mesh_batch * ((a.ncols + 255) / 256) < 256)) rt >>= 1;
    if (int f = env_int("PIT_FORCE_RT")) rt = f;
    if (const int tpwg0 = tiles_per_wg_for(a.ncols, (long)((n_tiles + rt - 1) / rt) * a.n_head * a.mesh_batch, work, a.mesh_batch)) {
        int tpwg = tpwg0;
        if (rt == 4) tpwg = 8;                                 // registers: RT*TPW accumulator tiles <= 4
        else if (rt == 2 && tpwg > 16) tpwg = 16;
        a.tiles_per_wg = tpwg;
        a.colgroups = ((a.ncols + 31) / 32 + tpwg - 1) / tpwg;
        dim3 grid(a.mesh_batch * a.colgroups, a.n_head, (n_tiles + rt - 1) / rt), block(512);
#define PIT_RT_BF(RT_, TPW_, BF_)                                                                                 \
        do {                                                                                                      \
            if (a.masked) hipLaunchKernelGGL((posatt_rows_tiles<RT_, TPW_, MODE, true, BF_>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((posatt_rows_tiles<RT_, TPW_, MODE, false, BF_>), grid, block, 0, s, a);      \
        } while (0)
#define PIT_RT(RT_, TPW_) do { if (bf) PIT_RT_BF(RT_, TPW_, (MODE == 0)); else PIT_RT_BF(RT_, TPW_, false); } while (0)
        if (rt == 4) PIT_RT(4, 1);
        else if (rt == 2) { if (tpwg == 16) PIT_RT(2, 2); else PIT_RT(2, 1); }
        else { if (tpwg == 32) PIT_RT(1, 4); else if (tpwg == 16) PIT_RT(1, 2); else PIT_RT(1, 1); }
#undef PIT_RT
#undef PIT_RT_BF
        return;
    }
    int ct = choose_ct(a.ncols, (long)n_tiles * a.n_head * a.mesh_batch);
    if (int f = env_int("PIT_FORCE_CT")) ct = f;
    a.colgroups = (a.ncols + 32 * ct - 1) / (32 * ct);
    const int wmax = (ct == 4) ? 4 : 8;                                  // parked tiles must fit 96 KiB of LDS
    int nwaves = max(1, min(wmax, pow2_floor(a.n_in / 32)));
    if (int f = env_int("PIT_FORCE_WAVES")) nwaves = pow2_floor(max(1, min(f, wmax)));     // (the epilogue is specialised for 1/2/4/8)
    dim3 grid(a.mesh_batch * a.colgroups, a.n_head, n_tiles), block(64 * nwaves);
    const size_t sm = rows_smem(ct, nwaves, a.n_in);
#define PIT_ROWS_BF(CT_, BF_)                                                                               \
    do {                                                                                                    \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_rows_kernel<CT_, MODE, true, BF_>,      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304),         \
                            (void)hipFuncSetAttribute((const void*)posatt_rows_kernel<CT_, MODE, false, BF_>,     \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);  \
        (void)once;                                                                                         \
        if (a.masked) hipLaunchKernelGGL((posatt_rows_kernel<CT_, MODE, true, BF_>), grid, block, sm, s, a); \
        else hipLaunchKernelGGL((posatt_rows_kernel<CT_, MODE, false, BF_>), grid, block, sm, s, a);        \
    } while (0)
#define PIT_ROWS(CT_) do { if (bf) PIT_ROWS_BF(CT_, (MODE == 0)); else PIT_ROWS_BF(CT_, false); } while (0)
#define PIT_ROWS_IL_BF(BF_)                                                                                 \
    do {                                                                                                    \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_rows_kernel<4, MODE, true, BF_, true>,  \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304),         \
                            (void)hipFuncSetAttribute((const void*)posatt_rows_kernel<4, MODE, false, BF_, true>, \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);  \
        (void)once;                                                                                         \
        if (a.masked) hipLaunchKernelGGL((posatt_rows_kernel<4, MODE, true, BF_, true>), grid, block, sm, s, a); \
        else hipLaunchKernelGGL((posatt_rows_kernel<4, MODE, false, BF_, true>), grid, block, sm, s, a);    \
    } while (0)
    if (ct == 4 && interleave_ok(a, MODE)) { if (bf) PIT_ROWS_IL_BF((MODE == 0)); else PIT_ROWS_IL_BF(false); }
    else if (ct == 4) PIT_ROWS(4);
    else if (ct == 2) PIT_ROWS(2);
    else PIT_ROWS(1);
#undef PIT_ROWS_IL_BF
#undef PIT_ROWS
#undef PIT_ROWS_BF
}

void launch_cols(const AttArgs& a0, hipStream_t s) {
    AttArgs a = a0;
    const bool bf = a.bf16 != 0;
    const int j_tiles = (a.n_in + 31) / 32;
    const long work = (long)j_tiles * a.mesh_batch * ((a.ncols + 31) / 32) * ((a.n_out + 1) / 2) * a.n_head;
    if (const int tpwg = tiles_per_wg_for(a.ncols, (long)j_tiles * a.mesh_batch, work, a.mesh_batch)) {
        a.tiles_per_wg = tpwg;
        a.colgroups = ((a.ncols + 31) / 32 + tpwg - 1) / tpwg;
        dim3 grid(a.colgroups, j_tiles, a.mesh_batch), block(512);
#define PIT_CT_BF(TPW_, BF_)                                                                             \
        do {                                                                                             \
            if (a.masked) hipLaunchKernelGGL((posatt_cols_tiles<TPW_, true, BF_>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((posatt_cols_tiles<TPW_, false, BF_>), grid, block, 0, s, a);        \
        } while (0)
#define PIT_CT(TPW_) do { if (bf) PIT_CT_BF(TPW_, true); else PIT_CT_BF(TPW_, false); } while (0)
        if (tpwg == 32) PIT_CT(4); else if (tpwg == 16) PIT_CT(2); else PIT_CT(1);
#undef PIT_CT
#undef PIT_CT_BF
        return;
    }
    int ct = choose_ct(a.ncols, (long)j_tiles * a.mesh_batch);
    if (int f = env_int("PIT_FORCE_CT")) ct = f;
    a.colgroups = (a.ncols + 32 * ct - 1) / (32 * ct);
    const int wmax = (ct == 4) ? 4 : 8;
    int nwaves = max(1, min(wmax, pow2_floor(a.n_out * a.n_head / 32)));
    if (int f = env_int("PIT_FORCE_WAVES")) nwaves = pow2_floor(max(1, min(f, wmax)));
    dim3 grid(a.colgroups, j_tiles, a.mesh_batch), block(64 * nwaves);
    const size_t sm = cols_smem(ct, nwaves, a.n_out);
#define PIT_COLS_BF(CT_, BF_)                                                                          \
    do {                                                                                               \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_cols_kernel<CT_, true, BF_>,       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304),    \
                            (void)hipFuncSetAttribute((const void*)posatt_cols_kernel<CT_, false, BF_>,      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true); \
        (void)once;                                                                                    \
        if (a.masked) hipLaunchKernelGGL((posatt_cols_kernel<CT_, true, BF_>), grid, block, sm, s, a);  \
        else hipLaunchKernelGGL((posatt_cols_kernel<CT_, false, BF_>), grid, block, sm, s, a);         \
    } while (0)
#define PIT_COLS(CT_) do { if (bf) PIT_COLS_BF(CT_, true); else PIT_COLS_BF(CT_, false); } while (0)
#define PIT_COLS_IL_BF(BF_)                                                                            \
    do {                                                                                               \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_cols_kernel<4, true, BF_, true>,   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304),    \
                            (void)hipFuncSetAttribute((const void*)posatt_cols_kernel<4, false, BF_, true>,  \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true); \
        (void)once;                                                                                    \
        if (a.masked) hipLaunchKernelGGL((posatt_cols_kernel<4, true, BF_, true>), grid, block, sm, s, a); \
        else hipLaunchKernelGGL((posatt_cols_kernel<4, false, BF_, true>), grid, block, sm, s, a);     \
    } while (0)
    if (ct == 4 && interleave_ok(a, 2)) { if (bf) PIT_COLS_IL_BF(true); else PIT_COLS_IL_BF(false); }
    else if (ct == 4) PIT_COLS(4);
    else if (ct == 2) PIT_COLS(2);
    else PIT_COLS(1);
#undef PIT_COLS_IL_BF
#undef PIT_COLS
#undef PIT_COLS_BF
}

// d(scale) + d(values) in one launch (posatt_bwd_pair_kernel) when both are in the small regime
// with the same column-tile count; returns false when the pair does not apply (caller launches
// the two kernels separately).

// the pair above plus the two weight-gradient reductions of the MLP that produced d_out (pit_hip.h: `rider`):
// three small latency-bound grids in one launch; the attention workgroups (the longer ones) come first
template <bool MASKED, bool BF>
__global__ __launch_bounds__(512, 4) void posatt_bwd_pair_dw_kernel(AttArgs ar, AttArgs ac, int n_cols_wgs, int cgx, int cgy,
                                                                     int rgx, int rgy, int n_att, pit_detail::DwPair w) {
    int id = blockIdx.x;
    if (id >= n_att) {
        dw_pair_body(w, id - n_att, pit_dyn_smem());
        return;
    }
    if (id < n_cols_wgs) {
        posatt_cols_body<1, MASKED, BF, 4>(ac, id % cgx, (id / cgx) % cgy, id / (cgx * cgy));
    } else {
        id -= n_cols_wgs;
        posatt_rows_body<1, 1, MASKED, false, 4>(ar, id % rgx, (id / rgx) % rgy, id / (rgx * rgy));
    }
}

// `rider` (may be null) is carried along when the narrow-tile kernel is the one chosen; *rider_done says so
bool launch_bwd_pair(const AttArgs& a0, hipStream_t s, const pit_mlp_params_job* job = nullptr, bool* rider_done = nullptr) {
    if (env_int("PIT_NO_BWD_PAIR") || env_int("PIT_FORCE_CT") || env_int("PIT_FORCE_WAVES")) return false;
    const int n_tiles = (a0.n_out + 31) / 32, j_tiles = (a0.n_in + 31) / 32;
    {   // either part would take the large-regime kernels: keep them separate
        const long work_r = (long)n_tiles * a0.n_head * a0.mesh_batch * ((a0.ncols + 31) / 32) * ((a0.n_in + 1) / 2);
        const long work_c = (long)j_tiles * a0.mesh_batch * ((a0.ncols + 31) / 32) * ((a0.n_out + 1) / 2) * a0.n_head;
        int rt = 4;
        while (rt > 1 && (n_tiles < rt || (long)((n_tiles + rt - 1) / rt) * a0.n_head * a0.mesh_batch * ((a0.ncols + 255) / 256) < 256)) rt >>= 1;
        if (tiles_per_wg_for(a0.ncols, (long)((n_tiles + rt - 1) / rt) * a0.n_head * a0.mesh_batch, work_r, a0.mesh_batch)) return false;
        if (tiles_per_wg_for(a0.ncols, (long)j_tiles * a0.mesh_batch, work_c, a0.mesh_batch)) return false;
    }
    const int ct = choose_ct(a0.ncols, (long)n_tiles * a0.n_head * a0.mesh_batch);
    if (ct != choose_ct(a0.ncols, (long)j_tiles * a0.mesh_batch)) return false;
    AttArgs ar = a0, ac = a0;
    ar.colgroups = ac.colgroups = (a0.ncols + 32 * ct - 1) / (32 * ct);
    const int wmax = (ct == 4) ? 4 : 8;
    const int nw_r = max(1, min(wmax, pow2_floor(a0.n_in / 32)));
    const int nw_c = max(1, min(wmax, pow2_floor(a0.n_out * a0.n_head / 32)));
    const int nwaves = min(nw_r, nw_c);
    const long rows_wgs = (long)a0.mesh_batch * ar.colgroups * a0.n_head * n_tiles;
    const long cols_wgs = (long)ac.colgroups * j_tiles * a0.mesh_batch;
    // merged launches pay off while the parts are latency-bound; big grids gain nothing
    if (rows_wgs + cols_wgs > 4096 || rows_wgs + cols_wgs > 0x7fffffffL) return false;
    const size_t sm = std::max(rows_smem(ct, nwaves, a0.n_in), cols_smem(ct, nwaves, a0.n_out));
    dim3 grid((unsigned)(rows_wgs + cols_wgs)), block(64 * nwaves);
    pit_detail::DwPair dw;
    const pit_detail::DwPair* rider = &dw;
    if (job && ct == 1 && nwaves >= 2 && pit_detail::plan_dw_pair(*job, nwaves, &dw)) {
        const int n_att = (int)(rows_wgs + cols_wgs);
        const size_t smw = std::max(sm, (size_t)nwaves * 16 * 64 * sizeof(float));
        dim3 gridw((unsigned)(n_att + rider->n1 + rider->n2));
#define PIT_PAIR_DW(M_, BF_)                                                                                  \
    do {                                                                                                      \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_bwd_pair_dw_kernel<M_, BF_>,        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);    \
        (void)once;                                                                                           \
        hipLaunchKernelGGL((posatt_bwd_pair_dw_kernel<M_, BF_>), gridw, block, smw, s, ar, ac, (int)cols_wgs, \
                           ac.colgroups, j_tiles, a0.mesh_batch * ar.colgroups, a0.n_head, n_att, *rider);    \
    } while (0)
        if (a0.masked) { if (a0.bf16) PIT_PAIR_DW(true, true); else PIT_PAIR_DW(true, false); }
        else { if (a0.bf16) PIT_PAIR_DW(false, true); else PIT_PAIR_DW(false, false); }
#undef PIT_PAIR_DW
        *rider_done = true;
        return true;
    }
#define PIT_PAIR_K(M_, BF_)                                                                                   \
    do {                                                                                                      \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_bwd_pair_kernel<M_, BF_>,                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);    \
        (void)once;                                                                                           \
        hipLaunchKernelGGL((posatt_bwd_pair_kernel<M_, BF_>), grid, block, sm, s, ar, ac, (int)cols_wgs,      \
                           ac.colgroups, j_tiles, a0.mesh_batch * ar.colgroups, a0.n_head);                   \
    } while (0)
#define PIT_PAIR_W(CT_, M_, BF_)                                                                              \
    do {                                                                                                      \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_bwd_pair_wide_kernel<CT_, M_, BF_>,       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);    \
        (void)once;                                                                                           \
        hipLaunchKernelGGL((posatt_bwd_pair_wide_kernel<CT_, M_, BF_>), wgrid, block, wsm, s, ar, ac, (int)cols_wgs, \
                           ac.colgroups, j_tiles, a0.mesh_batch * ar.colgroups, a0.n_head, (int)(rows_wgs + cols_wgs), wdw); \
    } while (0)
#define PIT_PAIR_IL(M_, BF_)                                                                                  \
    do {                                                                                                      \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_bwd_pair_wide_kernel<4, M_, BF_, true>,   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);    \
        (void)once;                                                                                           \
        hipLaunchKernelGGL((posatt_bwd_pair_wide_kernel<4, M_, BF_, true>), wgrid, block, wsm, s, ar, ac, (int)cols_wgs, \
                           ac.colgroups, j_tiles, a0.mesh_batch * ar.colgroups, a0.n_head, (int)(rows_wgs + cols_wgs), wdw); \
    } while (0)
    const bool il = ct == 4 && interleave_ok(a0, 1) && interleave_ok(a0, 2);
    // the wide kernels (four or eight waves, 96 KiB of LDS allowed) carry the job as gemm_rr_tile tiles whatever its size; the tiles contract in
    // the kernel's math mode, which is the job's
    pit_detail::DwPair wdw = pit_detail::DwPair();
    dim3 wgrid = grid;
    size_t wsm = sm;
    static const bool no_wide_rider = getenv("PIT_NO_WIDE_DW_RIDER") != nullptr;
    // (... while the attention itself is latency-bound: same-box A/B per fp32 step, carried / own launch - NACA 728 points x hid 128
    // 1.419 / 1.465 ms, Vorticity 256 x 256 x 2 heads 1.219 / 1.231, Elasticity 972 x 256 x 2 2.460 / 2.393: there the attention
    // workgroups run the fp32 matrix pipe at 62 % and the tiles compete for it)
    const double att_work = (double)a0.n_out * a0.n_in * a0.ncols * a0.n_head * a0.mesh_batch;
    if (job && rider_done && ct != 1 && nwaves >= 4 && !no_wide_rider && att_work <= 2.5e9 && ((job->math_mode & 0xff) == PIT_MATH_BF16) == (a0.bf16 != 0) &&
        pit_detail::plan_rr_rider(*job, &wdw, 256)) {
        wgrid = dim3((unsigned)(rows_wgs + cols_wgs + wdw.n1 + wdw.n2));
        wsm = std::max(sm, (size_t)65536);
        *rider_done = true;
    } else {
        wdw = pit_detail::DwPair();
    }
#define PIT_PAIR_CT(M_, BF_) do { if (ct == 1) PIT_PAIR_K(M_, BF_); else if (ct == 2) PIT_PAIR_W(2, M_, BF_); else if (il) PIT_PAIR_IL(M_, BF_); else PIT_PAIR_W(4, M_, BF_); } while (0)
    if (a0.masked) { if (a0.bf16) PIT_PAIR_CT(true, true); else PIT_PAIR_CT(true, false); }
    else { if (a0.bf16) PIT_PAIR_CT(false, true); else PIT_PAIR_CT(false, false); }
#undef PIT_PAIR_CT
#undef PIT_PAIR_IL
#undef PIT_PAIR_W
#undef PIT_PAIR_K
    return true;
}

// ------------------------------------------------------------------------------------
// Sparse path for the masked (locality < 1) layers: O(N*k) instead of O(N*J).
//
// pit_neighbors_fwd gives every row a candidate list that is a superset of its kept set for any
// head scale (k+2 keys plus ties).  One wavefront owns one output row: lanes evaluate the exact
// mask / weights on the candidates (lane = candidate), then the kept candidates are walked in
// groups of G with lane = value column (64*CR columns per wave, all loads of a group in flight),
// accumulating with plain FMAs - with ~10-40 keys per row there is no contraction worth an MFMA.
// Rows whose list overflowed (count > cap: heavy ties / duplicate points) scan all keys instead.
// d(values) walks the transposed lists (key -> rows); overflowed rows are added with atomics.
struct SparseArgs {
    const int* nbr_idx; const int* nbr_cnt; int cap;
    const int* rev_ptr; const int* rev_row; long rev_stride;
};

template <int NH, int CR, int MODE>
__device__ __forceinline__ void sparse_rows_body(const AttArgs& a, const SparseArgs& sp, const int bx, const int by, const int bz) {
    // keys gathered per batch: G x CR value registers in flight; at 8 columns per lane two keys keep the rows kernels
    // at 5 waves per SIMD (93 registers; four keys: 128) - these kernels hide gather latency with occupancy
    constexpr int G = (CR >= 8) ? 2 : ((CR >= 4) ? 4 : 8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows_total = (long)a.mesh_batch * a.n_out;
    const long row_raw = (long)bx * 4 + wave;
    const bool active = row_raw < rows_total;          // tail waves recompute the last row, write nothing
    const long row = active ? row_raw : rows_total - 1;
    const int mb = (int)(row / a.n_out), n = (int)(row - (long)mb * a.n_out);
    const int cblk = by, h0 = bz * NH;

    const unsigned mo_bytes = (unsigned)((long)a.mesh_batch * a.n_out * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, mi_bytes);
    const __amdgpu_buffer_rsrc_t rvals = make_rsrc(a.values, a.values_bytes);
    const unsigned ld4 = (unsigned)a.ld_values * 4u;
    const float4 xo = load_point4(rmo, mo_bytes, row, a.sdim, a.coords_used);

    float c[NH], T[NH], smin[NH], invl[NH], mbar[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        c[h] = a.head_is_scale ? a.head[h0 + h] : head_scale_from_lmda(a.head[h0 + h]);
        if (MODE == 0) {
            T[h] = quantile_lerp(__fmul_rn(c[h], a.stats[row]), __fmul_rn(c[h], a.stats[rows_total + row]), a.rank_w);
            smin[h] = __fmul_rn(c[h], a.stats[2 * rows_total + row]);
            invl[h] = 0.0f; mbar[h] = 0.0f;
        } else {
            const float4 rs4 = *reinterpret_cast<const float4*>(a.rowstat + (((long)mb * a.n_head + h0 + h) * a.n_out + n) * 4);
            T[h] = rs4.x; smin[h] = rs4.y; invl[h] = rs4.z; mbar[h] = rs4.w;
        }
    }
    // coordinate channels (coord_dims > 0: the torch.cat((mesh_in, func_in), -1) of the task forwards, e.g.
    // train_darcy.py:51-55, is never materialised): channel cd < coord_dims of a value row IS coordinate cd of that
    // key, read from mesh_in; the remaining channels come from `values`, which then holds dim - coord_dims channels
    const int kd = a.coord_dims;
    // (nothing of the coordinate path may stay live across the key loop when kd == 0: two more per-column arrays
    // took the 8-columns-per-lane instantiation from 4 to 3 waves per SIMD - measured +27 % on the batch-256
    // up-projection - so the coordinate offsets are re-derived inside the `if (kd)` block)
    unsigned uoff[CR];
    bool cvalid[CR], vload[CR];
    int cb[CR], cd[CR];
#pragma unroll
    for (int r = 0; r < CR; ++r) {
        const int col = cblk * 64 * CR + r * 64 + lane;
        cvalid[r] = col < a.ncols;
        const int cc = cvalid[r] ? col : 0;
        col_split(a, cc, mb, cb[r], cd[r]);
        vload[r] = cvalid[r] && cd[r] >= kd;                 // this column is read from `values`
        uoff[r] = (unsigned)(((long)cb[r] * a.values_bstride + (cd[r] - kd)) * 4);
    }
    float acc[NH][CR];
    float rsum[NH], qsum[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        rsum[h] = 0.0f; qsum[h] = 0.0f;
#pragma unroll
        for (int r = 0; r < CR; ++r) acc[h][r] = 0.0f;
    }
    const bool per = a.periodic != 0;
    const int* list = sp.nbr_idx + row * sp.cap;
    // the first 64 list entries are requested TOGETHER with the count (the list has cap slots whatever the count says): one
    // dependent round trip less in front of the coordinates (round 4: these launches are chains of ~5 round trips)
    const int j_first = list[lane < sp.cap ? lane : sp.cap - 1];
    const int cnt = sp.nbr_cnt[row];
    const bool scan_all = cnt > sp.cap;
    const int total = scan_all ? a.n_in : cnt;

    for (int base = 0; base < total; base += 64) {
        const int i = base + lane;
        const bool valid = i < total;
        const int j = valid ? (scan_all ? i : (base == 0 ? j_first : list[i])) : 0;
        const float4 xi = load_point4(rmi, mi_bytes, (long)mb * a.n_in + j, a.sdim, a.coords_used);
        const float m = sq_dist3(xo.x, xo.y, xo.z, xi.x, xi.y, xi.z, per, a.period);
        float p[NH];
        bool any = false;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float sv = __fmul_rn(m, c[h]);
            const bool keep = valid && (sv <= T[h]);
            float pv = keep ? __expf(smin[h] - sv) : 0.0f;
            if (MODE == 0) { rsum[h] += pv; qsum[h] += pv * m; }
            else pv = pv * (m - mbar[h]) * invl[h];
            p[h] = pv;
            any |= keep;
        }
        unsigned long long mask = __builtin_amdgcn_ballot_w64(any);
        while (mask) {
            int ji[G];
            float pi[NH][G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool has = mask != 0ull;
                const int src = has ? __builtin_ctzll(mask) : 0;
                if (has) mask &= mask - 1ull;
                ji[g] = has ? __builtin_amdgcn_readlane(j, src) : -1;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float pv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[h]), src));
                    pi[h][g] = has ? pv : 0.0f;
                }
            }
            float v[G][CR];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < CR; ++r)
                    v[g][r] = buf_load(rvals, (ji[g] >= 0 && vload[r]) ? uoff[r] + (unsigned)ji[g] * ld4 : a.values_bytes);
            if (kd) {                              // wave-uniform: the coordinate channels of these keys (an out-of-range load adds 0)
#pragma unroll
                for (int r = 0; r < CR; ++r) {
                    const bool cc = cvalid[r] && cd[r] < kd;
                    const unsigned coff = (unsigned)(((long)mb * a.n_in * a.sdim + cd[r]) * 4);
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        v[g][r] += buf_load(rmi, (ji[g] >= 0 && cc) ? coff + (unsigned)ji[g] * (unsigned)a.sdim * 4u : mi_bytes);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int r = 0; r < CR; ++r) acc[h][r] += pi[h][g] * v[g][r];
        }
    }

    if (MODE == 1) {
        const __amdgpu_buffer_rsrc_t rdo = make_rsrc(a.d_out, a.dout_bytes);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            double part = 0.0;
#pragma unroll
            for (int r = 0; r < CR; ++r) {
                const unsigned off = (unsigned)((long)cb[r] * a.dout_bstride + a.out_col0 + (long)(h0 + h) * a.dim + cd[r]) +
                                     (unsigned)n * (unsigned)a.ld_dout;
                part += (double)acc[h][r] * (double)dout_load(rdo, cvalid[r] ? off : a.dout_elems, a.dout16);
            }
            part = wave_sum_d(part);
            const int slot = (int)((bx + 131u * by + 977u * wave) & (a.nslots - 1));
            dscale_add(a.dscale_acc + (h0 + h) * PIT_DSCALE_SLOTS + slot, -part, lane == 0 && active);
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float rs = wave_sum(rsum[h]);
        const float qs = wave_sum(qsum[h]);
        const float inv = rs > 0.0f ? 1.0f / rs : 0.0f;
#pragma unroll
        for (int r = 0; r < CR; ++r)
            if (cvalid[r] && active) {
                const long o = (long)cb[r] * a.out_bstride + (long)n * a.ld_out + a.out_col0 + (long)(h0 + h) * a.dim + cd[r];
                if (a.out16) reinterpret_cast<unsigned short*>(a.out)[o] = f_to_bf16(acc[h][r] * inv);
                else a.out[o] = acc[h][r] * inv;
            }
        if (cblk == 0 && lane == 0 && active) {
            float4 st; st.x = T[h]; st.y = smin[h]; st.z = inv; st.w = qs * inv;
            *reinterpret_cast<float4*>(a.rowstat + (((long)mb * a.n_head + h0 + h) * a.n_out + n) * 4) = st;
            if (a.scale_out && row == 0) a.scale_out[h0 + h] = c[h];
        }
    }
    if (a.copy_inputs && h0 == 0) {
#pragma unroll
        for (int r = 0; r < CR; ++r) {
            const float iv = buf_load(rvals, cvalid[r] ? uoff[r] + (unsigned)n * ld4 : a.values_bytes);
            if (cvalid[r] && active) a.out[(long)cb[r] * a.out_bstride + (long)n * a.ld_out + cd[r]] = iv;
        }
    }
}

template <int NH, int CR, int MODE>
__global__ __launch_bounds__(256, 4) void posatt_sparse_rows(AttArgs a, SparseArgs sp) {
    sparse_rows_body<NH, CR, MODE>(a, sp, blockIdx.x, blockIdx.y, blockIdx.z);
}

// XCD-aware launch order for many column blocks (batch folded into the columns): workgroups are dealt
// round-robin to the 8 XCDs by their linear id, so id % 8 selects the column block residue - every
// XCD's L2 then holds only its own column slices of the gathered rows instead of all of them.
// 1-D grid of 8 * gx * ceil(gy/8) * gz workgroups; (gx, gy) = row groups, column blocks.
__device__ __forceinline__ bool xcd_remap(int id, int gx, int gy, int& bx, int& by, int& bz) {
    const int gyc = (gy + 7) / 8;
    const int xcd = id & 7, t = id >> 3;
    bx = t % gx;
    const int u = t / gx;
    by = (u % gyc) * 8 + xcd;
    bz = u / gyc;
    return by < gy;
}
template <int NH, int CR, int MODE>
__global__ __launch_bounds__(256, 4) void posatt_sparse_rows_x(AttArgs a, SparseArgs sp, int gx, int gy) {
    int bx, by, bz;
    if (!xcd_remap((int)blockIdx.x, gx, gy, bx, by, bz)) return;
    sparse_rows_body<NH, CR, MODE>(a, sp, bx, by, bz);
}

// D16 (PIT_IO_DOUT_BF16): d_out holds bf16 - a lane then owns CR/2 PAIRS of adjacent columns {128 q + 2 lane, +1} and one
// 4-B load fetches both (2-B loads per lane made the launch 3x slower than the fp32 one: 615 vs 220 us on the Vorticity
// decoder); needs CR, dim and out_col0 even (the launch helper checks), cross attention only (no residual).
template <int CR, bool D16 = false>
__device__ __forceinline__ void sparse_cols_body(const AttArgs& a, const SparseArgs& sp, const int bx, const int by) {
    static_assert(!D16 || CR % 2 == 0, "bf16 d_out: column pairs");
    constexpr int G = (CR >= 8) ? 2 : ((CR >= 4) ? 4 : 8);      // (see sparse_rows_body)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long keys_total = (long)a.mesh_batch * a.n_in;
    const long kid = (long)bx * 4 + wave;
    if (kid >= keys_total) return;
    const int mb = (int)(kid / a.n_in), j = (int)(kid - (long)mb * a.n_in);
    const int cblk = by;
    const unsigned mo_bytes = (unsigned)((long)a.mesh_batch * a.n_out * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, mi_bytes);
    const __amdgpu_buffer_rsrc_t rdout = make_rsrc(a.d_out, a.dout_bytes);
    constexpr unsigned ES = D16 ? 2u : 4u;                   // bytes per d_out element
    const unsigned ldd4 = (unsigned)a.ld_dout * ES;
    const float4 xi = load_point4(rmi, mi_bytes, kid, a.sdim, a.coords_used);
    unsigned doff[CR];
    bool cvalid[CR];
    int cb[CR], cd[CR];
#pragma unroll
    for (int r = 0; r < CR; ++r) {
        const int col = D16 ? cblk * 64 * CR + (r >> 1) * 128 + 2 * lane + (r & 1) : cblk * 64 * CR + r * 64 + lane;
        cvalid[r] = col < a.ncols;
        const int cc = cvalid[r] ? col : 0;
        col_split(a, cc, mb, cb[r], cd[r]);
        doff[r] = (unsigned)(((long)cb[r] * a.dout_bstride + a.out_col0 + cd[r]) * ES);
    }
    float acc[CR];
#pragma unroll
    for (int r = 0; r < CR; ++r) acc[r] = 0.0f;
    const bool per = a.periodic != 0;
    const int beg = sp.rev_ptr[(long)mb * (a.n_in + 1) + j], end = sp.rev_ptr[(long)mb * (a.n_in + 1) + j + 1];
    const int* rrow = sp.rev_row + (long)mb * sp.rev_stride;

    for (int h = 0; h < a.n_head; ++h) {
        const float c = a.head_is_scale ? a.head[h] : head_scale_from_lmda(a.head[h]);
        const unsigned hoff = (unsigned)h * (unsigned)a.dim * ES;
        for (int base = beg; base < end; base += 64) {
            const int e = base + lane;
            int nrow = (e < end) ? rrow[e] : -1;
            const bool valid = nrow >= 0;                   // -1: slot of a row that overflowed its list
            nrow = valid ? nrow : 0;
            const float4 xo = load_point4(rmo, mo_bytes, (long)mb * a.n_out + nrow, a.sdim, a.coords_used);
            const float4 rs4 = *reinterpret_cast<const float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + nrow) * 4);
            const float m = sq_dist3(xo.x, xo.y, xo.z, xi.x, xi.y, xi.z, per, a.period);
            const float sv = __fmul_rn(m, c);
            const bool keep = valid && (sv <= rs4.x);
            const float p = keep ? __expf(rs4.y - sv) * rs4.z : 0.0f;
            unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
            while (mask) {
                int ni[G];
                float pi[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const bool has = mask != 0ull;
                    const int src = has ? __builtin_ctzll(mask) : 0;
                    if (has) mask &= mask - 1ull;
                    ni[g] = has ? __builtin_amdgcn_readlane(nrow, src) : -1;
                    const float pv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), src));
                    pi[g] = has ? pv : 0.0f;
                }
                float v[G][CR];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (D16) {
#pragma unroll
                        for (int q = 0; q < CR / 2; ++q) {
                            const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(
                                rdout, (int)((ni[g] >= 0 && cvalid[2 * q]) ? doff[2 * q] + hoff + (unsigned)ni[g] * ldd4 : a.dout_bytes), 0, 0);
                            v[g][2 * q] = __uint_as_float(w << 16);
                            v[g][2 * q + (CR > 1 ? 1 : 0)] = __uint_as_float(w & 0xffff0000u);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < CR; ++r)
                            v[g][r] = buf_load(rdout, (ni[g] >= 0 && cvalid[r]) ? doff[r] + hoff + (unsigned)ni[g] * ldd4 : a.dout_bytes);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int r = 0; r < CR; ++r) acc[r] += pi[g] * v[g][r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < CR; ++r) {
        float res = 0.0f;
        if (!D16) {
            const unsigned roff = (unsigned)(((long)cb[r] * a.dout_bstride + cd[r]) * 4) + (unsigned)j * ldd4;
            res = buf_load(rdout, (a.add_residual && cvalid[r]) ? roff : a.dout_bytes);
        }
        // (coordinate channels carry no gradient: the meshes are data; d_values holds the dim - coord_dims others)
        if (cvalid[r] && cd[r] >= a.coord_dims)
            a.d_values[(long)cb[r] * a.dvalues_bstride + (long)j * a.ld_dvalues + (cd[r] - a.coord_dims)] = acc[r] + res;
    }
}

template <int CR, bool D16 = false>
__global__ __launch_bounds__(256) void posatt_sparse_cols(AttArgs a, SparseArgs sp) {
    sparse_cols_body<CR, D16>(a, sp, blockIdx.x, blockIdx.y);
}
template <int CR, bool D16 = false>
__global__ __launch_bounds__(256) void posatt_sparse_cols_x(AttArgs a, SparseArgs sp, int gx, int gy) {
    int bx, by, bz;
    if (!xcd_remap((int)blockIdx.x, gx, gy, bx, by, bz)) return;
    sparse_cols_body<CR, D16>(a, sp, bx, by);
}

// rows whose candidate list overflowed are not in the transposed lists: add their contribution
// to d(values) with atomics (rare: duplicated points / massive ties).
__device__ __forceinline__ void sparse_overflow_body(const AttArgs& a, const SparseArgs& sp, const int bx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows_total = (long)a.mesh_batch * a.n_out;
    const long row = (long)bx * 4 + wave;
    if (row >= rows_total) return;
    if (sp.nbr_cnt[row] <= sp.cap) return;
    const int mb = (int)(row / a.n_out), n = (int)(row - (long)mb * a.n_out);
    const unsigned mo_bytes = (unsigned)((long)a.mesh_batch * a.n_out * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, mi_bytes);
    const float4 xo = load_point4(rmo, mo_bytes, row, a.sdim, a.coords_used);
    const bool per = a.periodic != 0;
    for (int h = 0; h < a.n_head; ++h) {
        const float c = a.head_is_scale ? a.head[h] : head_scale_from_lmda(a.head[h]);
        const float4 rs4 = *reinterpret_cast<const float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + n) * 4);
        for (int base = 0; base < a.n_in; base += 64) {
            const int j = base + lane;
            const bool valid = j < a.n_in;
            const float4 xi = load_point4(rmi, mi_bytes, (long)mb * a.n_in + (valid ? j : 0), a.sdim, a.coords_used);
            const float m = sq_dist3(xo.x, xo.y, xo.z, xi.x, xi.y, xi.z, per, a.period);
            const float sv = __fmul_rn(m, c);
            const bool keep = valid && (sv <= rs4.x);
            const float p = keep ? __expf(rs4.y - sv) * rs4.z : 0.0f;
            unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                const int jj = __builtin_amdgcn_readlane(j, src);
                const float pv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), src));
                for (int col = lane; col < a.ncols; col += 64) {
                    int bb, dd;
                    col_split(a, col, mb, bb, dd);
                    if (dd < a.coord_dims) continue;
                    const long go = (long)bb * a.dout_bstride + (long)n * a.ld_dout + a.out_col0 + (long)h * a.dim + dd;
                    const float g = a.dout16 ? bf16_to_f(reinterpret_cast<const unsigned short*>(a.d_out)[go]) : a.d_out[go];
                    atomicAdd(a.d_values + (long)bb * a.dvalues_bstride + (long)jj * a.ld_dvalues + (dd - a.coord_dims), pv * g);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void posatt_sparse_overflow_cols(AttArgs a, SparseArgs sp) {
    sparse_overflow_body(a, sp, blockIdx.x);
}

// d(values) over the transposed lists and d(scale) over the rows of a sparse layer in ONE launch:
// workgroups [0, n_cols) walk the transposed lists (critical path first), the rest the d(scale)
// rows.  The overflow pass stays a separate, later launch: it ADDS to d(values) with atomics and
// must come after the plain stores of the key-owning waves.
template <int NH, int CRR, int CRC, bool D16 = false>
__global__ __launch_bounds__(256) void posatt_sparse_bwd_kernel(AttArgs a, SparseArgs sp, int n_cols, int cgx,
                                                                 int rgx, int rgy) {
    int id = blockIdx.x;
    if (id < n_cols) {
        sparse_cols_body<CRC, D16>(a, sp, id % cgx, id / cgx);
    } else {
        id -= n_cols;
        sparse_rows_body<NH, CRR, 1>(a, sp, id % rgx, (id / rgx) % rgy, id / (rgx * rgy));
    }
}

// The same launch carrying a postponed MLP's weight-gradient reductions (pit_hip.h: rider) - a kernel of its own,
// instantiated for the small layers only (<= 4 columns per lane): inlined into the kernel above, the reductions'
// body cost its 8-columns-per-lane instantiation 28 % (Vorticity's up-projection 168 -> 232 us).  The reductions
// take the FIRST n_dw workgroup ids: the launch has more workgroups than the chip holds at once.
template <int NH, int CRR, int CRC>
__global__ __launch_bounds__(256) void posatt_sparse_bwd_dw_kernel(AttArgs a, SparseArgs sp, int n_cols, int cgx,
                                                                    int rgx, int rgy, int n_dw, pit_detail::DwPair w) {
    int id = blockIdx.x;
    if (id < n_dw) {
        dw_pair_body(w, id, pit_dyn_smem());
        return;
    }
    id -= n_dw;
    if (id < n_cols) {
        sparse_cols_body<CRC>(a, sp, id % cgx, id / cgx);
    } else {
        id -= n_cols;
        sparse_rows_body<NH, CRR, 1>(a, sp, id % rgx, (id / rgx) % rgy, id / (rgx * rgy));
    }
}

// ------------------------------------------------------------------------------------
// Round 4: masked layers as UNION-TILE contractions (coherent row orderings: grids, body-fitted meshes).
//
// The candidate-list kernels above give a wavefront to every output row and gather the ~k value rows of its kept keys:
// neighbouring rows of a mesh fetch the same few keys over and over (NACA decoder: 225 k rows x 16 keys x 512 B = 1.85 GB
// of L2 gathers per launch, three launches per step).  Here ONE WAVEFRONT takes 16 CONSECUTIVE rows (the M of
// v_mfma_f32_16x16x4_f32), four lanes per row, and works entirely on its own - no workgroup barrier anywhere:
//   1. lane q of a row holds the row's candidates i = q, q + 4, ... and their squared distances in registers;
//   2. the union of the 16 lists: an LDS bitmap over the keys, popcount scan, compacted to a sorted list (U keys, typically
//      25-60 for 16-key rows); a candidate's position in the union = word base + popcount of the lower bits;
//   3. the exact weights of every (row, candidate) pair - the same distance, threshold test and exp as sparse_rows_body -
//      scattered into a dense 16 x U tile in LDS (zero where a key is not a row's candidate);
//   4. MODE 0 / 1: out tile (16 rows x 128 columns per pass) = P (16 x U) V_union (U x 128), every union key's value row
//      fetched ONCE per tile as 32-byte pieces per lane (column c of MFMA tile t = lane column * 8 + t, so a lane's eight
//      accumulators are eight consecutive columns: float4 loads and stores);
//      MODE 2: d(values)[union keys] += P^T (U x 16) dO (16 x 128), added to memory with fp32 atomics (the caller zeroes
//      d_values; run-to-run the sums differ in the last bits - the host keeps the transposed-list kernel when
//      torch.use_deterministic_algorithms is on).
// Unions beyond UW_CH keys (incoherent orderings, overflowed lists) are walked in chunks: correct for any input, fast only
// when neighbouring rows share their keys - the host decides per kind of mesh plan (ops.MeshPlan.union_tiles).
// MFMA work = rows x U x columns instead of rows x k x columns of VALU FMAs, U / k ~ 2-3 for coherent meshes (a 64-row
// tile's union is ~130 keys: the first version of this kernel spent 47 us of a 140 us launch in zero-padding).
constexpr int UW_ROWS = 16, UW_CH = 64, UW_CB = 128;
#ifndef UW_OCC
#define UW_OCC 4      // wavefronts per SIMD the one-head kernels are compiled for (measured: 4 with 2-step fetch groups > 3 with 4)
#endif
#ifndef UW_FG
#define UW_FG 2       // MFMA steps (of 4 keys) per fetch group; two groups in flight
#endif
typedef float f32x4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4u mfma_16x16x4_u(float a, float b, f32x4u c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// LDS written by some lanes of a wavefront, read by others: the LDS pipeline is in order per wavefront, the compiler
// must be told not to move accesses across
#define PIT_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                                 __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// One wavefront's 16-row tile: the registers and the wave-private LDS of steps 1-3 above.
// EPL: candidates per lane (four lanes per row): 8 for capacities <= 32, 16 for <= 64
template <int NH, int MODE, int EPL>
struct UnionTile {
    static constexpr int PS = 68;                                     // P row stride (floats): conflict-free operand reads
    const AttArgs& a; const SparseArgs& sp;
    float* P;                                                         // [NH][16][PS]
    unsigned* bm;                                                     // [128] key bitmap, then [128] word bases
    unsigned short* ulist;                                            // [n_in] union keys (sorted)
    int lane, r, q, mb, r0, U;
    bool rvalid, ovf;
    unsigned mi_bytes;
    __amdgpu_buffer_rsrc_t rmi;
    float4 xo;
    float c[NH], rc[NH][4];                                           // scale; per row {T, S_min, 1/rowsum, mbar}
    int jr[EPL];                                                      // candidate keys, then their union positions (-1: none)
    float mr[EPL];                                                    // their squared distances

    __device__ __forceinline__ UnionTile(const AttArgs& a_, const SparseArgs& sp_, char* wave_lds, int lane_, int mb_, int r0_)
        : a(a_), sp(sp_), lane(lane_), r(lane_ & 15), q(lane_ >> 4), mb(mb_), r0(r0_) {
        P = reinterpret_cast<float*>(wave_lds);
        bm = reinterpret_cast<unsigned*>(P + NH * UW_ROWS * PS);
        ulist = reinterpret_cast<unsigned short*>(bm + 256);
        mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
        rmi = make_rsrc(a.mesh_in, mi_bytes);
#pragma unroll
        for (int h = 0; h < NH; ++h) c[h] = a.head[h];                // (the host passes the scale itself)
    }
    __device__ __forceinline__ float dist_to(int j) const {
        const float4 xi = load_point4(rmi, mi_bytes, (long)mb * a.n_in + j, a.sdim, a.coords_used);
        return sq_dist3(xo.x, xo.y, xo.z, xi.x, xi.y, xi.z, a.periodic != 0, a.period);
    }
    // one (row, key) weight per head from the squared distance m
    __device__ __forceinline__ float weigh(float m, int h) const {
        const float sv = __fmul_rn(m, c[h]);
        return sv <= rc[h][0] ? __expf(rc[h][1] - sv) : 0.0f;
    }
    // ---- the row's constants, its candidates and their distances: registers for the rest of the tile
    __device__ __forceinline__ void load() {
        const long rows_total = (long)a.mesh_batch * a.n_out;
        const unsigned mo_bytes = (unsigned)(rows_total * a.sdim * 4);
        const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, mo_bytes);
        rvalid = r0 + r < a.n_out;
        const long grow = (long)mb * a.n_out + (rvalid ? r0 + r : a.n_out - 1);
        {
            // (the list slots are read before the count arrives - every slot below the capacity is addressable - and
            // masked afterwards: one memory round trip instead of two)
            const int* lst = sp.nbr_idx + grow * sp.cap;
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                const int i = q + 4 * t;
                jr[t] = i < sp.cap ? lst[i] : -1;
            }
        }
        const int cnt = rvalid ? sp.nbr_cnt[grow] : 0;
        ovf = cnt > sp.cap;                                           // (an overflowed row scans every key)
        xo = load_point4(rmo, mo_bytes, grow, a.sdim, a.coords_used);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if (MODE == 0) {
                rc[h][0] = quantile_lerp(__fmul_rn(c[h], a.stats[grow]), __fmul_rn(c[h], a.stats[rows_total + grow]), a.rank_w);
                rc[h][1] = __fmul_rn(c[h], a.stats[2 * rows_total + grow]);
                rc[h][2] = 0.0f; rc[h][3] = 0.0f;
            } else {
                const float4 rs4 = *reinterpret_cast<const float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + (rvalid ? r0 + r : a.n_out - 1)) * 4);
                rc[h][0] = rs4.x; rc[h][1] = rs4.y; rc[h][2] = rs4.z; rc[h][3] = rs4.w;
            }
        }
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const unsigned j = (unsigned)jr[t] < (unsigned)a.n_in ? (unsigned)jr[t] : 0u;   // (slots past the count hold anything)
            mr[t] = dist_to((int)j);
        }
#pragma unroll
        for (int t = 0; t < EPL; ++t)
            if (ovf || q + 4 * t >= cnt) jr[t] = -1;
    }
    // ---- union of the 16 candidate lists; the candidates' keys become their positions in it
    __device__ __forceinline__ void unite() {
        const int nwords = (a.n_in + 31) / 32;
        const bool any_ovf = __builtin_amdgcn_ballot_w64(ovf) != 0ull;
        PIT_WAVE_LDS_SYNC();                                          // (a previous tile's reads of bm / ulist are done)
        bm[lane] = 0u; bm[lane + 64] = 0u;
        PIT_WAVE_LDS_SYNC();
        if (any_ovf) {
#pragma unroll
            for (int w = lane; w < 128; w += 64) {
                const int rem = a.n_in - 32 * w;
                if (rem > 0) bm[w] = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
            }
        } else {
#pragma unroll
            for (int t = 0; t < EPL; ++t)
                if (jr[t] >= 0) atomicOr(&bm[jr[t] >> 5], 1u << (jr[t] & 31));
        }
        PIT_WAVE_LDS_SYNC();
        {                                                             // word bases: exclusive scan of the popcounts (<= 128 words)
            const unsigned w0 = lane < nwords ? bm[lane] : 0u, w1 = lane + 64 < nwords ? bm[lane + 64] : 0u;
            const int v0 = __popc(w0), v1 = __popc(w1);
            int s0 = v0, s1 = v1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t0 = __shfl_up(s0, o, 64), t1 = __shfl_up(s1, o, 64);
                if (lane >= o) { s0 += t0; s1 += t1; }
            }
            const int tot0 = __shfl(s0, 63, 64), tot1 = __shfl(s1, 63, 64);
            U = tot0 + tot1;
            int b0 = s0 - v0, b1 = tot0 + s1 - v1;
            bm[128 + lane] = (unsigned)b0; bm[192 + lane] = (unsigned)b1;
            unsigned bits = w0;
            while (bits) { const int t = __builtin_ctz(bits); bits &= bits - 1u; ulist[b0++] = (unsigned short)(32 * lane + t); }
            bits = w1;
            while (bits) { const int t = __builtin_ctz(bits); bits &= bits - 1u; ulist[b1++] = (unsigned short)(32 * (lane + 64) + t); }
        }
        PIT_WAVE_LDS_SYNC();
#pragma unroll
        for (int t = 0; t < EPL; ++t)
            if (jr[t] >= 0) jr[t] = (int)bm[128 + (jr[t] >> 5)] + __popc(bm[jr[t] >> 5] & ((1u << (jr[t] & 31)) - 1u));
    }
    // ---- the weight tile of union positions [c0, c0 + UW_CH)
    __device__ __forceinline__ void put(float m, int pos) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float pv = weigh(m, h);
            if (MODE == 1) pv = pv * (m - rc[h][3]) * rc[h][2];
            else if (MODE == 2) pv = pv * rc[h][2];
            P[(h * UW_ROWS + r) * PS + pos] = pv;
        }
    }
    __device__ __forceinline__ void build(int c0) {
        PIT_WAVE_LDS_SYNC();                                          // (the previous chunk's operand reads are done)
        for (int e = lane; e < NH * UW_ROWS * PS / 4; e += 64) reinterpret_cast<float4*>(P)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        PIT_WAVE_LDS_SYNC();
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const int pos = jr[t] - c0;
            if (jr[t] >= 0 && pos >= 0 && pos < UW_CH) put(mr[t], pos);
        }
        if (ovf) {
            const int uc = min(UW_CH, U - c0);
            for (int pos = q; pos < uc; pos += 4) put(dist_to((int)ulist[c0 + pos]), pos);
        }
        PIT_WAVE_LDS_SYNC();
    }
};

// MODE 0: forward; MODE 1: d(scale)
template <int NH, int MODE, int EPL>
__global__ __launch_bounds__(256, NH == 1 ? UW_OCC : 1) void posatt_union_kernel(AttArgs a, SparseArgs sp, int cb_per_wave, int wave_lds_bytes) {
    typedef UnionTile<NH, MODE, EPL> Tile;
    constexpr int PS = Tile::PS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mb = blockIdx.z, r0 = (blockIdx.y * 4 + wave) * UW_ROWS;
    if (r0 >= a.n_out) return;                                        // (no workgroup barrier below)
    Tile T(a, sp, reinterpret_cast<char*>(lds) + (size_t)wave * wave_lds_bytes, lane, mb, r0);
    const int r = T.r, q = T.q;
    T.load();
    T.unite();
    const int U = T.U;
    float inv[NH];                                                    // MODE 0: 1 / row sum, on all four lanes of the row
    if (MODE == 0) {
        // ---- row sums (and sum p m for the backward's mbar): private partial sums of the row's four lanes, combined in a
        //      fixed order - the same bits on every run
        float ps[NH], pm[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) { ps[h] = 0.0f; pm[h] = 0.0f; }
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            if (T.jr[t] < 0) continue;
#pragma unroll
            for (int h = 0; h < NH; ++h) { const float pv = T.weigh(T.mr[t], h); ps[h] += pv; pm[h] += pv * T.mr[t]; }
        }
        if (T.ovf)
            for (int pos = q; pos < U; pos += 4) {
                const float m = T.dist_to((int)T.ulist[pos]);
#pragma unroll
                for (int h = 0; h < NH; ++h) { const float pv = T.weigh(m, h); ps[h] += pv; pm[h] += pv * m; }
            }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            ps[h] += __shfl_xor(ps[h], 16, 64); ps[h] += __shfl_xor(ps[h], 32, 64);
            pm[h] += __shfl_xor(pm[h], 16, 64); pm[h] += __shfl_xor(pm[h], 32, 64);
            inv[h] = ps[h] > 0.0f ? 1.0f / ps[h] : 0.0f;
            if (q == 0 && blockIdx.x == 0 && T.rvalid)
                *reinterpret_cast<float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + r0 + r) * 4) =
                    make_float4(T.rc[h][0], T.rc[h][1], inv[h], pm[h] * inv[h]);
        }
        if (a.scale_out && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && tid < NH)
            a.scale_out[tid] = a.head[tid];                           // the backward reads the scale from here
    }
    const bool single = U <= UW_CH;
    if (single) T.build(0);
    // ---- out tile = P V_union, the value rows of the union fetched two groups of keys ahead of the MFMAs
    const float* P = T.P;
    const unsigned short* ulist = T.ulist;
    const __amdgpu_buffer_rsrc_t rvals = make_rsrc(a.values, a.values_bytes);
    const unsigned ld4 = (unsigned)a.ld_values * 4u;
    for (int cbi = 0; cbi < cb_per_wave; ++cbi) {
        const int col0 = (blockIdx.x * cb_per_wave + cbi) * UW_CB;
        if (col0 >= a.ncols) break;
        const int col = col0 + 8 * r;                                 // this lane's eight consecutive columns (dim % 8 == 0)
        const bool cvalid = col < a.ncols;
        int cb, cd;
        col_split(a, cvalid ? col : 0, mb, cb, cd);
        const unsigned uoff = (unsigned)(((long)cb * a.values_bstride + cd) * 4);
        f32x4u acc[NH][8];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[h][t] = f32x4u{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < U; c0 += UW_CH) {
            if (!single) T.build(c0);
            const int uc = min(UW_CH, U - c0);
            const int nst = (uc + 3) >> 2;                            // steps of 4 keys
            auto fetch = [&](int s0, float (&bv)[UW_FG][8]) {
#pragma unroll
                for (int g = 0; g < UW_FG; ++g) {
                    const int k = 4 * (s0 + g) + q;
                    const unsigned key = (unsigned)ulist[c0 + (k < uc ? k : 0)];
                    const unsigned off = (cvalid && k < uc) ? uoff + key * ld4 : a.values_bytes;
                    float lo[4], hi[4];
                    buf_load4(rvals, off, lo);
                    buf_load4(rvals, (cvalid && k < uc) ? off + 16u : a.values_bytes, hi);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { bv[g][t] = lo[t]; bv[g][4 + t] = hi[t]; }
                }
            };
            auto mma = [&](int s0, const float (&bv)[UW_FG][8]) {
#pragma unroll
                for (int g = 0; g < UW_FG; ++g) {
                    if (s0 + g >= nst) break;
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const float av = P[(h * UW_ROWS + r) * PS + 4 * (s0 + g) + q];
#pragma unroll
                        for (int t = 0; t < 8; ++t) acc[h][t] = mfma_16x16x4_u(av, bv[g][t], acc[h][t]);
                    }
                }
            };
            float b0[UW_FG][8], b1[UW_FG][8];
            fetch(0, b0);
            for (int s0 = 0; s0 < nst; s0 += 2 * UW_FG) {
                if (s0 + UW_FG < nst) fetch(s0 + UW_FG, b1);
                mma(s0, b0);
                if (s0 + UW_FG < nst) {
                    if (s0 + 2 * UW_FG < nst) fetch(s0 + 2 * UW_FG, b0);
                    mma(s0 + UW_FG, b1);
                }
            }
        }
        // accumulator i of every tile = row 4 q + i, columns cd .. cd + 7 of head h
        if (MODE == 1) {
            const __amdgpu_buffer_rsrc_t rdo = make_rsrc(a.d_out, a.dout_bytes);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                double part = 0.0;
                const unsigned dbase = (unsigned)((long)cb * a.dout_bstride + a.out_col0 + (long)h * a.dim + cd);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = r0 + 4 * q + i;
                    const bool ok = cvalid && n < a.n_out;
                    const unsigned e = dbase + (unsigned)n * (unsigned)a.ld_dout;
                    float dv[8];
                    if (a.dout16) {
                        const i32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rdo, ok ? (int)(e * 2u) : (int)a.dout_bytes, 0, 0);
                        const unsigned ww[4] = {(unsigned)w.x, (unsigned)w.y, (unsigned)w.z, (unsigned)w.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t) { dv[2 * t] = __uint_as_float(ww[t] << 16); dv[2 * t + 1] = __uint_as_float(ww[t] & 0xFFFF0000u); }
                    } else {
                        float lo[4], hi[4];
                        buf_load4(rdo, ok ? e * 4u : a.dout_bytes, lo);
                        buf_load4(rdo, ok ? e * 4u + 16u : a.dout_bytes, hi);
#pragma unroll
                        for (int t = 0; t < 4; ++t) { dv[t] = lo[t]; dv[4 + t] = hi[t]; }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) part += (double)acc[h][t][i] * (double)dv[t];
                }
                part = wave_sum_d(part);
                const int slot = (int)((blockIdx.y * 4u + wave + 131u * blockIdx.x + 31u * cbi) & (a.nslots - 1));
                dscale_add(a.dscale_acc + h * PIT_DSCALE_SLOTS + slot, -part, lane == 0);
            }
            continue;
        }
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = r0 + 4 * q + i;
                const float rinv = __shfl(inv[h], 4 * q + i, 64);       // (lane `row` is the row's q = 0 lane)
                if (!cvalid || n >= a.n_out) continue;
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = acc[h][t][i] * rinv;
                const long o = (long)cb * a.out_bstride + (long)n * a.ld_out + a.out_col0 + (long)h * a.dim + cd;
                if (a.out16) {
                    uint4 w;
                    w.x = (unsigned)f_to_bf16(v[0]) | ((unsigned)f_to_bf16(v[1]) << 16);
                    w.y = (unsigned)f_to_bf16(v[2]) | ((unsigned)f_to_bf16(v[3]) << 16);
                    w.z = (unsigned)f_to_bf16(v[4]) | ((unsigned)f_to_bf16(v[5]) << 16);
                    w.w = (unsigned)f_to_bf16(v[6]) | ((unsigned)f_to_bf16(v[7]) << 16);
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.out) + o) = w;
                } else {
                    *reinterpret_cast<float4*>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(a.out + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
    }
}

// d(values)[key] += sum_h sum_rows P_h[row][key] dO_h[row] WITHOUT transposed lists: a workgroup of 8 wavefronts owns
// DV_RB = 256 CONSECUTIVE rows.  The keys its rows touch (a bitmap in LDS, compacted: ~100-200 for a coherent ordering) are
// the M dimension of a transposed contraction  D[slot][column] = sum over the block's rows of  P^T[slot][row] dO[row][column]:
// P^T is built 64 rows at a time in LDS ([DV_KC slots][64 rows], four lanes per row scatter the row's weights), wavefront w
// owns columns [16 w, 16 w + 16) and keeps its D tiles (DV_KC / 16 MFMA tiles of 16 slots) in registers over all the rows -
// the accumulation IS the MFMA's, no adds in LDS (measured: ds_add_f32 sustains one lane per ~3.5 clocks per CU, 490 us for
// this layer) - skipping the 16-slot x 4-row operand blocks that hold only zeros (most: a row touches ~16 of the slots).
// One pass at the end adds the block's D to memory with fp32 atomics (the caller zeroes d_values; ~rows / 256 x slots x
// columns adds instead of rows / 16 x ~45 x columns: 86 M -> ~20 M for a NACA decoder layer).  More than DV_KC keys in a
// block (incoherent orderings, overflowed lists): the slots are walked in ranges of DV_KC - correct for any input.
constexpr int DV_WAVES = 8, DV_RB = 256, DV_KC = 192, DV_PS = 68;
template <int NH, int EPL>
__global__ __launch_bounds__(512, (NH == 1 && EPL == 8) ? 4 : 2) void posatt_union_dv_kernel(AttArgs a, SparseArgs sp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* PT = lds;                                                  // [DV_KC][DV_PS]: P^T of 64 rows
    unsigned* wbm = reinterpret_cast<unsigned*>(PT + DV_KC * DV_PS);  // [128] bitmap of the block's keys, [128] word bases
    int* wflag = reinterpret_cast<int*>(wbm + 256);                   // [0] = keys in the bitmap, [1] = some list overflowed
    unsigned short* wkeys = reinterpret_cast<unsigned short*>(wflag + 4);   // [n_in] slot -> key
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int mb = blockIdx.z, rb0 = blockIdx.y * DV_RB;
    const int nwords = (a.n_in + 31) / 32;
    const long rows_total = (long)a.mesh_batch * a.n_out;
    const unsigned mo_bytes = (unsigned)(rows_total * a.sdim * 4);
    const unsigned mi_bytes = (unsigned)((long)a.mesh_batch * a.n_in * a.sdim * 4);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(a.mesh_out, mo_bytes);
    const __amdgpu_buffer_rsrc_t rmi = make_rsrc(a.mesh_in, mi_bytes);
    const bool per = a.periodic != 0;
    float c[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) c[h] = a.head[h];                    // (the host passes the scale itself)
    // ---- four lanes per row, two rows per thread (rows rb0 + p * 128 + tid / 4): candidates, distances, row constants
    const int rq = tid & 3;
    int js[2][EPL];                                                   // candidate keys, then their slots (-1: none)
    float ms[2][EPL];
    float rT[2][NH], rS[2][NH], rI[2][NH];                            // threshold, S_min, 1 / row sum
    float4 xo[2];
    bool ovf[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int n = rb0 + p * 128 + (tid >> 2);
        const bool rvalid = n < a.n_out;
        const long grow = (long)mb * a.n_out + (rvalid ? n : a.n_out - 1);
        const int* lst = sp.nbr_idx + grow * sp.cap;
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const int i = rq + 4 * t;
            js[p][t] = i < sp.cap ? lst[i] : -1;                      // (read before the count arrives, masked below)
        }
        const int cnt = rvalid ? sp.nbr_cnt[grow] : 0;
        ovf[p] = cnt > sp.cap;
        xo[p] = load_point4(rmo, mo_bytes, grow, a.sdim, a.coords_used);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float4 rs4 = *reinterpret_cast<const float4*>(a.rowstat + (((long)mb * a.n_head + h) * a.n_out + (rvalid ? n : a.n_out - 1)) * 4);
            rT[p][h] = rs4.x; rS[p][h] = rs4.y; rI[p][h] = rvalid ? rs4.z : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const unsigned j = (unsigned)js[p][t] < (unsigned)a.n_in ? (unsigned)js[p][t] : 0u;
            const float4 xi = load_point4(rmi, mi_bytes, (long)mb * a.n_in + j, a.sdim, a.coords_used);
            ms[p][t] = sq_dist3(xo[p].x, xo[p].y, xo[p].z, xi.x, xi.y, xi.z, per, a.period);
        }
#pragma unroll
        for (int t = 0; t < EPL; ++t)
            if (ovf[p] || rq + 4 * t >= cnt) js[p][t] = -1;
    }
    // ---- the block's key set
    if (tid < 256) wbm[tid] = 0u;
    if (tid == 0) wflag[1] = 0;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (ovf[p]) wflag[1] = 1;
#pragma unroll
        for (int t = 0; t < EPL; ++t)
            if (js[p][t] >= 0) atomicOr(&wbm[js[p][t] >> 5], 1u << (js[p][t] & 31));
    }
    __syncthreads();
    if (wflag[1]) {                                                   // a row whose list overflowed scans ALL keys
        if (tid < 128) { const int rem = a.n_in - 32 * tid; wbm[tid] = rem <= 0 ? 0u : (rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u)); }
        __syncthreads();
    }
    if (wave == 0) {
        const unsigned w0 = lane < nwords ? wbm[lane] : 0u, w1 = lane + 64 < nwords ? wbm[lane + 64] : 0u;
        const int v0 = __popc(w0), v1 = __popc(w1);
        int s0 = v0, s1 = v1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t0 = __shfl_up(s0, o, 64), t1 = __shfl_up(s1, o, 64);
            if (lane >= o) { s0 += t0; s1 += t1; }
        }
        const int tot0 = __shfl(s0, 63, 64), tot1 = __shfl(s1, 63, 64);
        int b0 = s0 - v0, b1 = tot0 + s1 - v1;
        wbm[128 + lane] = (unsigned)b0; wbm[192 + lane] = (unsigned)b1;
        if (lane == 0) wflag[0] = tot0 + tot1;
        unsigned bits = w0;
        while (bits) { const int t = __builtin_ctz(bits); bits &= bits - 1u; wkeys[b0++] = (unsigned short)(32 * lane + t); }
        bits = w1;
        while (bits) { const int t = __builtin_ctz(bits); bits &= bits - 1u; wkeys[b1++] = (unsigned short)(32 * (lane + 64) + t); }
    }
    __syncthreads();
    const int UW = wflag[0];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < EPL; ++t)
            if (js[p][t] >= 0) js[p][t] = (int)wbm[128 + (js[p][t] >> 5)] + __popc(wbm[js[p][t] >> 5] & ((1u << (js[p][t] & 31)) - 1u));
    const __amdgpu_buffer_rsrc_t rdo = make_rsrc(a.d_out, a.dout_bytes);
    const int ncb = (a.ncols + UW_CB - 1) / UW_CB;
    for (int cbi = blockIdx.x; cbi < ncb; cbi += gridDim.x) {
        const int col = cbi * UW_CB + 16 * wave + r;                  // this lane's column
        const bool cvalid = col < a.ncols;
        int cb, cd;
        col_split(a, cvalid ? col : 0, mb, cb, cd);
        const unsigned dob = (unsigned)((long)cb * a.dout_bstride + a.out_col0 + cd);
        for (int sp0 = 0; sp0 < UW; sp0 += DV_KC) {
            const int kc = min(DV_KC, UW - sp0);
            const int nmt = (kc + 15) >> 4;
            f32x4u acc[DV_KC / 16];
#pragma unroll
            for (int mt = 0; mt < DV_KC / 16; ++mt) acc[mt] = f32x4u{0.f, 0.f, 0.f, 0.f};
            for (int ch = 0; ch < DV_RB / 64; ++ch) {
                if (rb0 + 64 * ch >= a.n_out) break;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    // d(out) of the chunk's rows 4 ks + q, this lane's column: in flight while P^T is built
                    float bv[16];
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const int n = rb0 + 64 * ch + 4 * ks + q;
                        bv[ks] = dout_load(rdo, (cvalid && n < a.n_out) ? dob + (unsigned)(h * a.dim) + (unsigned)n * (unsigned)a.ld_dout : a.dout_elems, a.dout16);
                    }
                    __syncthreads();                                  // (the previous chunk's operand reads are done)
                    for (int e = tid; e < nmt * 16 * DV_PS / 4; e += 512) reinterpret_cast<float4*>(PT)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                    __syncthreads();
                    if ((tid >> 8) == (ch & 1)) {                     // the threads that hold this chunk's rows
                        const int p = ch >> 1, row = (tid >> 2) & 63;
                        auto put = [&](float m, int slot) {
                            const float sv = __fmul_rn(m, c[h]);
                            if (sv <= (p ? rT[1][h] : rT[0][h]))
                                PT[slot * DV_PS + row] = __expf((p ? rS[1][h] : rS[0][h]) - sv) * (p ? rI[1][h] : rI[0][h]);
                        };
#pragma unroll
                        for (int t = 0; t < EPL; ++t) {
                            const int slot = (p ? js[1][t] : js[0][t]) - sp0;
                            if ((p ? js[1][t] : js[0][t]) >= 0 && slot >= 0 && slot < DV_KC) put(p ? ms[1][t] : ms[0][t], slot);
                        }
                        if (p ? ovf[1] : ovf[0]) {
                            const float4 x = p ? xo[1] : xo[0];
                            for (int sl = rq; sl < kc; sl += 4) {
                                const float4 xi = load_point4(rmi, mi_bytes, (long)mb * a.n_in + (int)wkeys[sp0 + sl], a.sdim, a.coords_used);
                                put(sq_dist3(x.x, x.y, x.z, xi.x, xi.y, xi.z, per, a.period), sl);
                            }
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int mt = 0; mt < DV_KC / 16; ++mt) {
                        if (mt < nmt) {
                            float av[16];                             // (all 16 operand reads in flight before the first test)
#pragma unroll
                            for (int ks = 0; ks < 16; ++ks) av[ks] = PT[(mt * 16 + r) * DV_PS + 4 * ks + q];
#pragma unroll
                            for (int ks = 0; ks < 16; ++ks)
                                if (__builtin_amdgcn_ballot_w64(av[ks] != 0.0f) != 0ull) acc[mt] = mfma_16x16x4_u(av[ks], bv[ks], acc[mt]);
                        }
                    }
                }
            }
            // ---- the block's sums to memory: accumulator i of tile mt = slot sp0 + 16 mt + 4 q + i, this lane's column.  (Round 6: the
            // same sums through LDS, 64 slots at a time, so that a wavefront's atomic instruction adds 256 contiguous bytes of one key row
            // - what made pit_fold.hip's epilogue - measured 163 against 145 us here: six more barriers per range, and the adds no longer
            // trickle out while other waves still contract.)
            if (cvalid) {
                float* gcol = a.d_values + (long)cb * a.dvalues_bstride + cd;
#pragma unroll
                for (int mt = 0; mt < DV_KC / 16; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int sl = 16 * mt + 4 * q + i;
                        if (sl < kc) atomicAdd(gcol + (long)wkeys[sp0 + sl] * a.ld_dvalues, acc[mt][i]);
                    }
            }
        }
    }
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (long k = i; k < n; ++k) p[k] = 0.0f;
    }
}

// LDS bytes of one wavefront of posatt_union_kernel
int union_wave_lds(int nh, int n_in) {
    return (int)((nh * UW_ROWS * 80 * sizeof(float) + 256 * sizeof(unsigned) + (size_t)n_in * sizeof(unsigned short) + 15) & ~(size_t)15);
}
// does the shape qualify?  (the caller made sure a.head holds the scales: head_is_scale)  Eight consecutive columns per lane:
// 16-byte loads and stores of the value / output rows.
bool union_ok(const AttArgs& a, const SparseArgs& sp) {
    return a.masked && sp.nbr_idx && sp.nbr_cnt && a.n_in <= 4096 && a.n_head <= 2 && a.coord_dims == 0 && sp.cap <= 64 &&
           a.n_out >= 16 && a.dim % 8 == 0 && a.ld_values % 4 == 0 && a.values_bstride % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(a.values) & 15) == 0 && !env_int("PIT_NO_UNION_TILES");
}
bool aligned_rows(const void* p, long ld, long bstride, int col0, int elems16) {   // 16-byte pieces of `elems16` elements
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % elems16 == 0 && bstride % elems16 == 0 && col0 % elems16 == 0;
}
template <int MODE>
void launch_union(const AttArgs& a, const SparseArgs& sp, hipStream_t s) {
    const int ncb = (a.ncols + UW_CB - 1) / UW_CB;
    const int rtiles = (a.n_out + UW_ROWS - 1) / UW_ROWS;
    // column passes per wavefront: the tile's weights are built once per wavefront - as many passes as keep >= 2048 wavefronts
    // (meshes shared by the batch, PIT_UNION_TILES=1: Vorticity b=20 with 1 / 2 / 4 / 8 / 16 / 40 passes 1.842 / 1.821 / 1.816 / 1.813 /
    // 1.863 / 2.077 ms against 1.835 on the per-row kernels; Darcy b=256 with 2 / 4 / 16: 1.858 / 1.833 / 1.867 against 1.859)
    int per = 1;
    while (per < ncb && (long)rtiles * a.mesh_batch * ((ncb + 2 * per - 1) / (2 * per)) >= 2048) per *= 2;
    dim3 grid((unsigned)((ncb + per - 1) / per), (unsigned)((rtiles + 3) / 4), (unsigned)a.mesh_batch), block(256);
    const int wl = union_wave_lds(a.n_head, a.n_in);
#define PIT_UN(NH_, EPL_) do {                                                                                             \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_union_kernel<NH_, MODE, EPL_>,                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);                 \
        (void)once;                                                                                                        \
        hipLaunchKernelGGL((posatt_union_kernel<NH_, MODE, EPL_>), grid, block, (size_t)4 * wl, s, a, sp, per, wl); } while (0)
    if (a.n_head == 2) { if (sp.cap <= 32) PIT_UN(2, 8); else PIT_UN(2, 16); }
    else               { if (sp.cap <= 32) PIT_UN(1, 8); else PIT_UN(1, 16); }
#undef PIT_UN
}
// d(values) of the union-tile form (posatt_union_dv_kernel); the caller zeroed d_values
void launch_union_dv(const AttArgs& a, const SparseArgs& sp, hipStream_t s) {
    const int ncb = (a.ncols + UW_CB - 1) / UW_CB;
    const size_t sm = (size_t)DV_KC * DV_PS * sizeof(float) + 256 * sizeof(unsigned) + 16 + (size_t)a.n_in * sizeof(unsigned short) + 16;
    dim3 grid((unsigned)std::min(ncb, 64), (unsigned)((a.n_out + DV_RB - 1) / DV_RB), (unsigned)a.mesh_batch), block(512);
#define PIT_UNDV(NH_, EPL_) do {                                                                                           \
        static bool once = ((void)hipFuncSetAttribute((const void*)posatt_union_dv_kernel<NH_, EPL_>,                      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);                 \
        (void)once;                                                                                                        \
        hipLaunchKernelGGL((posatt_union_dv_kernel<NH_, EPL_>), grid, block, sm, s, a, sp); } while (0)
    if (a.n_head == 2) { if (sp.cap <= 32) PIT_UNDV(2, 8); else PIT_UNDV(2, 16); }
    else               { if (sp.cap <= 32) PIT_UNDV(1, 8); else PIT_UNDV(1, 16); }
#undef PIT_UNDV
}

// columns per lane: as many as the column count allows, fewer when the launch would otherwise
// have too few wavefronts to hide the gather latency (one wave = one row / key)
int cr_for(int ncols, long units) {
    int cr = ncols > 256 ? 8 : (ncols > 128 ? 4 : (ncols > 64 ? 2 : 1));
    while (cr > 1 && units * ((ncols + 64 * cr - 1) / (64 * cr)) < 2048) cr >>= 1;
    return cr;
}

// d(scale) of a candidate-list layer with a postponed MLP's weight-gradient reductions in the same launch
template <int NH, int CR>
__global__ __launch_bounds__(256, 4) void posatt_sparse_rows_dw(AttArgs a, SparseArgs sp, int gx, int gy, int n_dw,
                                                              pit_detail::DwPair w) {
    int id = blockIdx.x;
    if (id < n_dw) {                                     // (first: see posatt_sparse_bwd_kernel)
        dw_pair_body(w, id, pit_dyn_smem());
        return;
    }
    id -= n_dw;
    sparse_rows_body<NH, CR, 1>(a, sp, id % gx, (id / gx) % gy, id / (gx * gy));
}

constexpr size_t DW_SMEM_4WAVES = 4 * 16 * 64 * sizeof(float);   // gemm_rd_body's parking area, 256-thread workgroups

// forward of a (small) candidate-list layer with the processor's block weights formed by extra workgroups of the launch
// (pit_posatt_fwd_job: they depend on the latent mesh and the lmda's only - one launch less in the step's chain)
template <int NH, int CR>
__global__ __launch_bounds__(256, 4) void posatt_sparse_rows_w(AttArgs a, SparseArgs sp, int gx, int gy, int n_att, WeightsArgs w) {
    const int id = blockIdx.x;
    if (id >= n_att) {                                   // (last: the attention rows are the longer chains)
        block_weights_body(w, id - n_att);
        return;
    }
    sparse_rows_body<NH, CR, 0>(a, sp, id % gx, (id / gx) % gy, id / (gx * gy));
}

template <int MODE>
void launch_sparse_rows(const AttArgs& a, const SparseArgs& sp, hipStream_t s, const pit_mlp_params_job* job = nullptr,
                        bool* rider_done = nullptr, const WeightsArgs* wjob = nullptr) {
    const int nh = (a.n_head % 2 == 0) ? 2 : 1;
    const long rows = (long)a.mesh_batch * a.n_out;
    const int cr = cr_for(a.ncols, rows * (a.n_head / nh));
    dim3 grid((unsigned)((rows + 3) / 4), (a.ncols + 64 * cr - 1) / (64 * cr), a.n_head / nh), block(256);
    if (MODE == 0 && wjob && (long)grid.x * grid.y * grid.z <= 4096) {       // small launch: carry the processor's weights
        const int n_att = (int)(grid.x * grid.y * grid.z);
        dim3 gridw((unsigned)(n_att + block_weights_wgs(wjob->n_layers, wjob->n_head, wjob->L)));
#define PIT_SRW(NH_, CR_) hipLaunchKernelGGL((posatt_sparse_rows_w<NH_, CR_>), gridw, block, 0, s, a, sp, (int)grid.x, (int)grid.y, n_att, *wjob)
#define PIT_SRW_CR(NH_) do { if (cr == 8) PIT_SRW(NH_, 8); else if (cr == 4) PIT_SRW(NH_, 4); else if (cr == 2) PIT_SRW(NH_, 2); else PIT_SRW(NH_, 1); } while (0)
        if (nh == 2) PIT_SRW_CR(2); else PIT_SRW_CR(1);
#undef PIT_SRW_CR
#undef PIT_SRW
        *rider_done = true;
        return;
    }
    pit_detail::DwPair dw;
    const pit_detail::DwPair* rider = &dw;
    if (MODE == 1 && job && (long)grid.x * grid.y * grid.z <= 16384 && pit_detail::plan_dw_pair(*job, 4, &dw)) {   // small launch: carry the reductions
        const int n_att = (int)(grid.x * grid.y * grid.z);
        dim3 gridw((unsigned)(n_att + rider->n1 + rider->n2));
        // (gemm_rr tiles for this rider - the encoder MLP's 64 x 64 reduction - measured neutral: 0.1985 vs 0.1994 ms)
#define PIT_SRW(NH_, CR_) hipLaunchKernelGGL((posatt_sparse_rows_dw<NH_, CR_>), gridw, block, DW_SMEM_4WAVES, s, a, sp, \
                                             (int)grid.x, (int)grid.y, rider->n1 + rider->n2, *rider)
#define PIT_SRW_CR(NH_) do { if (cr == 8) PIT_SRW(NH_, 8); else if (cr == 4) PIT_SRW(NH_, 4); else if (cr == 2) PIT_SRW(NH_, 2); else PIT_SRW(NH_, 1); } while (0)
        if (nh == 2) PIT_SRW_CR(2); else PIT_SRW_CR(1);
#undef PIT_SRW_CR
#undef PIT_SRW
        *rider_done = true;
        return;
    }
    const long xtotal = 8L * grid.x * ((grid.y + 7) / 8) * grid.z;
    const bool remap = grid.y >= 16 && xtotal < 0x7fffffffL && !env_int("PIT_NO_XCD_REMAP");   // (few blocks: padding costs more than locality gains)
#define PIT_SR(NH_, CR_)                                                                                              \
    do {                                                                                                              \
        if (remap) hipLaunchKernelGGL((posatt_sparse_rows_x<NH_, CR_, MODE>), dim3((unsigned)xtotal), block, 0, s, a, sp, \
                                      (int)grid.x, (int)grid.y);                                                      \
        else hipLaunchKernelGGL((posatt_sparse_rows<NH_, CR_, MODE>), grid, block, 0, s, a, sp);                      \
    } while (0)
#define PIT_SR_CR(NH_) do { if (cr == 8) PIT_SR(NH_, 8); else if (cr == 4) PIT_SR(NH_, 4); else if (cr == 2) PIT_SR(NH_, 2); else PIT_SR(NH_, 1); } while (0)
    if (nh == 2) PIT_SR_CR(2); else PIT_SR_CR(1);
#undef PIT_SR_CR
#undef PIT_SR
}

// merged d(values) + d(scale) launch for a sparse layer; false when the launch would be large or
// the parts' columns-per-lane are a combination that is not instantiated (caller launches the
// parts separately)
bool launch_sparse_bwd_pair(const AttArgs& a, const SparseArgs& sp, bool complete, hipStream_t s,
                            const pit_mlp_params_job* job = nullptr, bool* rider_done = nullptr) {
    if (env_int("PIT_NO_BWD_PAIR")) return false;
    const int nh = (a.n_head % 2 == 0) ? 2 : 1;
    const long rows = (long)a.mesh_batch * a.n_out, keys = (long)a.mesh_batch * a.n_in;
    const int crr = cr_for(a.ncols, rows * (a.n_head / nh));
    int crc = cr_for(a.ncols, keys);
    if (crc != crr && crc != 1) crc = (crc > crr) ? crr : 1;     // instantiated: equal, or 1 column per lane for d(values)
    if (a.dout16) {                                               // bf16 d_out: the column-pair variant of d(values), no rider
        if (crr < 2 || a.dim % 2 || a.out_col0 % 2 || a.ncols % 2 || a.ld_dout % 2 || a.dout_bstride % 2) return false;
        crc = crr; job = nullptr;
    }
    const int rblocks = (a.ncols + 64 * crr - 1) / (64 * crr), cblocks = (a.ncols + 64 * crc - 1) / (64 * crc);
    const long rgx = (rows + 3) / 4, cgx = (keys + 3) / 4;
    const long n_rows = rgx * rblocks * (a.n_head / nh), n_cols = cgx * cblocks;
    if (n_rows + n_cols > 16384) return false;            // big launches gain nothing from merging
    dim3 grid((unsigned)(n_rows + n_cols)), block(256);
    pit_detail::DwPair dw;
    if (job && crr <= 4 && pit_detail::plan_dw_pair(*job, 4, &dw)) {
        const int n_dw = dw.n1 + dw.n2;
        dim3 gridw((unsigned)(n_rows + n_cols + n_dw));
#define PIT_SBW(NH_, CRR_, CRC_) hipLaunchKernelGGL((posatt_sparse_bwd_dw_kernel<NH_, CRR_, CRC_>), gridw, block, DW_SMEM_4WAVES, s, a, sp, (int)n_cols, (int)cgx, (int)rgx, rblocks, n_dw, dw)
#define PIT_SBW_C(NH_, CRR_) do { if (crc == CRR_) PIT_SBW(NH_, CRR_, CRR_); else PIT_SBW(NH_, CRR_, 1); } while (0)
#define PIT_SBW_CR(NH_) do { if (crr == 4) PIT_SBW_C(NH_, 4); else if (crr == 2) PIT_SBW_C(NH_, 2); else PIT_SBW(NH_, 1, 1); } while (0)
        if (nh == 2) PIT_SBW_CR(2); else PIT_SBW_CR(1);
#undef PIT_SBW_CR
#undef PIT_SBW_C
#undef PIT_SBW
        *rider_done = true;
        if (!complete) hipLaunchKernelGGL(posatt_sparse_overflow_cols, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a, sp);
        return true;
    }
#define PIT_SB(NH_, CRR_, CRC_) hipLaunchKernelGGL((posatt_sparse_bwd_kernel<NH_, CRR_, CRC_>), grid, block, 0, s, a, sp, (int)n_cols, (int)cgx, (int)rgx, rblocks)
#define PIT_SB16(NH_, CRR_) hipLaunchKernelGGL((posatt_sparse_bwd_kernel<NH_, CRR_, CRR_, true>), grid, block, 0, s, a, sp, (int)n_cols, (int)cgx, (int)rgx, rblocks)
#define PIT_SB_C(NH_, CRR_) do { if (a.dout16) PIT_SB16(NH_, CRR_); else if (crc == CRR_) PIT_SB(NH_, CRR_, CRR_); else PIT_SB(NH_, CRR_, 1); } while (0)
#define PIT_SB_CR(NH_) do { if (crr == 8) PIT_SB_C(NH_, 8); else if (crr == 4) PIT_SB_C(NH_, 4); else if (crr == 2) PIT_SB_C(NH_, 2); else PIT_SB(NH_, 1, 1); } while (0)
    if (nh == 2) PIT_SB_CR(2); else PIT_SB_CR(1);
#undef PIT_SB_CR
#undef PIT_SB_C
#undef PIT_SB16
#undef PIT_SB
    if (!complete) hipLaunchKernelGGL(posatt_sparse_overflow_cols, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a, sp);
    return true;
}

void launch_sparse_cols(const AttArgs& a, const SparseArgs& sp, bool complete, hipStream_t s) {
    const long keys = (long)a.mesh_batch * a.n_in;
    const int cr = a.dout16 ? std::max(2, cr_for(a.ncols, keys)) : cr_for(a.ncols, keys);
    dim3 grid((unsigned)((keys + 3) / 4), (a.ncols + 64 * cr - 1) / (64 * cr)), block(256);
    const long xtotal = 8L * grid.x * ((grid.y + 7) / 8);
    const bool remap = grid.y >= 16 && xtotal < 0x7fffffffL && !env_int("PIT_NO_XCD_REMAP");   // (few blocks: padding costs more than locality gains)
#define PIT_SC(CR_)                                                                                                   \
    do {                                                                                                              \
        if (remap) hipLaunchKernelGGL((posatt_sparse_cols_x<CR_>), dim3((unsigned)xtotal), block, 0, s, a, sp,        \
                                      (int)grid.x, (int)grid.y);                                                      \
        else hipLaunchKernelGGL((posatt_sparse_cols<CR_>), grid, block, 0, s, a, sp);                                 \
    } while (0)
#define PIT_SC16(CR_)                                                                                                 \
    do {                                                                                                              \
        if (remap) hipLaunchKernelGGL((posatt_sparse_cols_x<CR_, true>), dim3((unsigned)xtotal), block, 0, s, a, sp,  \
                                      (int)grid.x, (int)grid.y);                                                      \
        else hipLaunchKernelGGL((posatt_sparse_cols<CR_, true>), grid, block, 0, s, a, sp);                           \
    } while (0)
    if (a.dout16) { if (cr == 8) PIT_SC16(8); else if (cr == 4) PIT_SC16(4); else PIT_SC16(2); }
    else if (cr == 8) PIT_SC(8); else if (cr == 4) PIT_SC(4); else if (cr == 2) PIT_SC(2); else PIT_SC(1);
#undef PIT_SC16
#undef PIT_SC
    const long rows = (long)a.mesh_batch * a.n_out;
    if (!complete) hipLaunchKernelGGL(posatt_sparse_overflow_cols, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a, sp);
}

int fill_common(AttArgs& a, const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                int space_dim, int metric, float period, const float* values, int batch, int dim,
                long ld_values, long values_bstride, const float* head, int n_head, int head_is_scale,
                int coord_dims = 0) {
    if (!mesh_out || !mesh_in || !values || !head) return PIT_ERR_NULL;
    if (coord_dims < 0 || coord_dims >= dim || coord_dims > space_dim) return PIT_ERR_SIZE;
    if (mesh_batch <= 0 || n_out <= 0 || n_in <= 0 || batch <= 0 || dim <= 0 || n_head <= 0) return PIT_ERR_SIZE;
    if (space_dim < 1 || space_dim > 3) return PIT_ERR_SIZE;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    if (mesh_batch != 1 && mesh_batch != batch) return PIT_ERR_SIZE;
    if (n_head > 65535 || (long)batch * dim > 0x7fffffffL) return PIT_ERR_UNSUPPORTED;
    a = AttArgs();
    a.mesh_out = mesh_out; a.mesh_in = mesh_in; a.mesh_batch = mesh_batch; a.n_out = n_out; a.n_in = n_in;
    a.sdim = space_dim; a.periodic = (metric != PIT_METRIC_EUCLID);
    a.coords_used = (metric == PIT_METRIC_PERIODIC1D) ? 1 : space_dim;
    a.period = period;
    a.values = values; a.batch = batch; a.dim = dim; a.ld_values = ld_values; a.values_bstride = values_bstride;
    a.head = head; a.n_head = n_head; a.head_is_scale = head_is_scale;
    a.ncols = (mesh_batch == 1) ? batch * dim : dim;
    a.coord_dims = coord_dims;
    const unsigned long long vb = ((unsigned long long)(batch - 1) * values_bstride +
                                   (unsigned long long)(n_in - 1) * ld_values + (dim - coord_dims)) * 4ull;
    if (vb > PIT_MAX_BUFFER_BYTES) return PIT_ERR_UNSUPPORTED;
    a.values_bytes = (unsigned)vb;
    a.bf16 = (t_call_math == PIT_MATH_BF16);
    a.dim_magic = (dim == 1) ? 0xFFFFFFFFu : (unsigned)(0x100000000ull / (unsigned long long)dim);
    a.no_fast_loads = env_int("PIT_NO_FAST_LOADS");
    return 0;
}

// ---- large-regime launches on precomputed weights (round 4)
// one AttArgs for a batch-free self-attention layer whose weights are given: no meshes, no head scale
int fill_pre(AttArgs& a, int n_pts, int n_head, int dim, int batch, const float* bsrc, long ld_b, long b_bstride) {
    if (n_pts <= 0 || batch <= 0 || dim <= 0 || n_head <= 0 || n_head > 65535 || (long)batch * dim > 0x7fffffffL) return PIT_ERR_SIZE;
    a = AttArgs();
    a.mesh_batch = 1; a.n_out = n_pts; a.n_in = n_pts; a.sdim = 1; a.coords_used = 1;
    a.values = bsrc; a.batch = batch; a.dim = dim; a.ld_values = ld_b; a.values_bstride = b_bstride;
    a.n_head = n_head; a.head_is_scale = 1;
    a.ncols = batch * dim;
    a.bf16 = (t_call_math == PIT_MATH_BF16);
    a.dim_magic = (dim == 1) ? 0xFFFFFFFFu : (unsigned)(0x100000000ull / (unsigned long long)dim);
    a.no_fast_loads = env_int("PIT_NO_FAST_LOADS");
    return 0;
}
// grid of posatt_rows_tiles for this shape (as launch_rows chooses it); false = the shape is not in the tiles regime
template <int MODE>
bool launch_rows_pre(AttArgs& a, hipStream_t s) {
    const bool bf = (MODE == 0) && a.bf16 != 0;
    const int n_tiles = (a.n_out + 31) / 32;
    const long work = (long)n_tiles * a.n_head * ((a.ncols + 31) / 32) * ((a.n_in + 1) / 2);
    int rt = 4;
    while (rt > 1 && (n_tiles < rt || (long)((n_tiles + rt - 1) / rt) * a.n_head * ((a.ncols + 255) / 256) < 256)) rt >>= 1;
    const int tpwg0 = tiles_per_wg_for(a.ncols, (long)((n_tiles + rt - 1) / rt) * a.n_head, work, 1);
    if (!tpwg0) return false;
    int tpwg = tpwg0;
    if (rt == 4) tpwg = 8;
    else if (rt == 2 && tpwg > 16) tpwg = 16;
    a.tiles_per_wg = tpwg;
    a.colgroups = ((a.ncols + 31) / 32 + tpwg - 1) / tpwg;
    dim3 grid(a.colgroups, a.n_head, (n_tiles + rt - 1) / rt), block(512);
#define PIT_RTP(RT_, TPW_) do { if (bf) hipLaunchKernelGGL((posatt_rows_tiles<RT_, TPW_, MODE, false, (MODE == 0), true>), grid, block, 0, s, a); \
                                else hipLaunchKernelGGL((posatt_rows_tiles<RT_, TPW_, MODE, false, false, true>), grid, block, 0, s, a); } while (0)
    if (rt == 4) PIT_RTP(4, 1);
    else if (rt == 2) { if (tpwg == 16) PIT_RTP(2, 2); else PIT_RTP(2, 1); }
    else { if (tpwg == 32) PIT_RTP(1, 4); else if (tpwg == 16) PIT_RTP(1, 2); else PIT_RTP(1, 1); }
#undef PIT_RTP
    return true;
}
bool launch_cols_pre(AttArgs& a, hipStream_t s) {
    const bool bf = a.bf16 != 0;
    const int j_tiles = (a.n_in + 31) / 32;
    const long work = (long)j_tiles * ((a.ncols + 31) / 32) * ((a.n_out + 1) / 2) * a.n_head;
    const int tpwg = tiles_per_wg_for(a.ncols, (long)j_tiles, work, 1);
    if (!tpwg) return false;
    a.tiles_per_wg = tpwg;
    a.colgroups = ((a.ncols + 31) / 32 + tpwg - 1) / tpwg;
    dim3 grid(a.colgroups, j_tiles, 1), block(512);
#define PIT_CTP(TPW_) do { if (bf) hipLaunchKernelGGL((posatt_cols_tiles<TPW_, false, true, true>), grid, block, 0, s, a); \
                           else hipLaunchKernelGGL((posatt_cols_tiles<TPW_, false, false, true>), grid, block, 0, s, a); } while (0)
    if (tpwg == 32) PIT_CTP(4); else if (tpwg == 16) PIT_CTP(2); else PIT_CTP(1);
#undef PIT_CTP
    return true;
}

}  // namespace

// include/pit_hip.h: 1 when pit_posatt_pre_fwd / _bwd run this shape (the large regime of a batch-free self-attention layer)
extern "C" int pit_posatt_pre_supported(int n_pts, int n_head, int dim, int batch) {
    static const bool off = getenv("PIT_NO_PRE_WEIGHTS") != nullptr;        // (diagnostic switch, read once)
    if (n_pts <= 0 || n_head <= 0 || dim <= 0 || batch <= 0 || off) return 0;
    const long ncols = (long)batch * dim;
    if (ncols > 0x7fffffffL || n_pts % 4 != 0 || n_pts > 2048) return 0;          // (E and Q: n_head * n_pts^2 floats per layer)
    const int n_tiles = (n_pts + 31) / 32;
    int rt = 4;
    while (rt > 1 && (n_tiles < rt || (long)((n_tiles + rt - 1) / rt) * n_head * ((ncols + 255) / 256) < 256)) rt >>= 1;
    const long work = (long)n_tiles * n_head * ((ncols + 31) / 32) * ((n_pts + 1) / 2);
    return tiles_per_wg_for((int)ncols, (long)((n_tiles + rt - 1) / rt) * n_head, work, 1) != 0 &&
           tiles_per_wg_for((int)ncols, (long)n_tiles, work, 1) != 0;
}

extern "C" int pit_posatt_pre_fwd(const float* e, const float* rowstat, int n_pts, int n_head, int dim, int batch,
                                  const float* values, long ld_values, long values_bstride,
                                  float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                                  int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (!e || !rowstat || !values || !out) return PIT_ERR_NULL;
    if (math_mode & ~0xff) return PIT_ERR_UNSUPPORTED;
    AttArgs a;
    if (int rc = fill_pre(a, n_pts, n_head, dim, batch, values, ld_values, values_bstride)) return rc;
    const unsigned long long vb = ((unsigned long long)(batch - 1) * values_bstride + (unsigned long long)(n_pts - 1) * ld_values + dim) * 4ull;
    if (vb > PIT_MAX_BUFFER_BYTES) return PIT_ERR_UNSUPPORTED;
    a.values_bytes = (unsigned)vb;
    a.out = out; a.ld_out = ld_out; a.out_bstride = out_bstride; a.out_col0 = out_col0; a.copy_inputs = copy_inputs;
    a.rowstat = const_cast<float*>(rowstat); a.pre_w = e;
    if (!launch_rows_pre<0>(a, (hipStream_t)stream)) return PIT_ERR_UNSUPPORTED;
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_posatt_pre_bwd(const float* e, const float* q, const float* rowstat, int n_pts, int n_head, int dim, int batch,
                                  const float* values, long ld_values, long values_bstride,
                                  const float* d_out, long ld_dout, long dout_bstride, int out_col0,
                                  float* d_values, long ld_dvalues, long dvalues_bstride, int add_residual,
                                  double* workspace, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (!e || !rowstat || !d_out || !d_values) return PIT_ERR_NULL;
    if (workspace && (!q || !values)) return PIT_ERR_NULL;
    if (math_mode & ~0xff) return PIT_ERR_UNSUPPORTED;
    const unsigned long long db = ((unsigned long long)(batch - 1) * dout_bstride + (unsigned long long)(n_pts - 1) * ld_dout +
                                   out_col0 + (unsigned long long)n_head * dim) * 4ull;
    if (db > PIT_MAX_BUFFER_BYTES) return PIT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    {   // d(values)[j] = residual + sum_h sum_n (E_h[n][j] / rowsum_h[n]) dO_h[n]
        AttArgs a;
        if (int rc = fill_pre(a, n_pts, n_head, dim, batch, nullptr, 0, 0)) return rc;
        a.rowstat = const_cast<float*>(rowstat); a.pre_w = e;
        a.d_out = d_out; a.ld_dout = ld_dout; a.dout_bstride = dout_bstride; a.out_col0 = out_col0;
        a.dout_bytes = (unsigned)db; a.dout_elems = (unsigned)(db / 4);
        a.d_values = d_values; a.ld_dvalues = ld_dvalues; a.dvalues_bstride = dvalues_bstride; a.add_residual = add_residual;
        if (!launch_cols_pre(a, s)) return PIT_ERR_UNSUPPORTED;
        PIT_CHECK_LAUNCH();
    }
    if (workspace) {
        // d c_h -= sum_{j,col} U[j,col] * sum_n Q_h[n][j] dO_h[n,col]: the rows kernel with the operands swapped - its B rows are
        // the head's columns of d_out, the dot runs against the values
        const unsigned long long vb = ((unsigned long long)(batch - 1) * values_bstride + (unsigned long long)(n_pts - 1) * ld_values + dim) * 4ull;
        if (vb > PIT_MAX_BUFFER_BYTES) return PIT_ERR_UNSUPPORTED;
        AttArgs a;
        if (int rc = fill_pre(a, n_pts, n_head, dim, batch, d_out, ld_dout, dout_bstride)) return rc;
        a.values_bytes = (unsigned)db;
        a.out_col0 = out_col0; a.pre_swap = 1; a.pre_w = q; a.rowstat = const_cast<float*>(rowstat);
        a.d_out = values; a.ld_dout = ld_values; a.dout_bstride = values_bstride;
        a.dout_bytes = (unsigned)vb; a.dout_elems = (unsigned)(vb / 4);
        a.dscale_acc = workspace; a.nslots = PIT_DSCALE_SLOTS;
        if (!launch_rows_pre<1>(a, s)) return PIT_ERR_UNSUPPORTED;
        PIT_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int pit_posatt_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                              int space_dim, int metric, float period,
                              const float* values, int batch, int dim, long ld_values, long values_bstride,
                              const float* head, int n_head, int head_is_scale,
                              const float* stats, float rank_w, int masked, int self_attn,
                              float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                              float* rowstat, float* scale_out,
                              const int* nbr_idx, const int* nbr_cnt, int nbr_cap, int coord_dims, int math_mode,
                              void* stream) {
    return pit_posatt_fwd_job(mesh_out, mesh_in, mesh_batch, n_out, n_in, space_dim, metric, period, values, batch, dim, ld_values,
                              values_bstride, head, n_head, head_is_scale, stats, rank_w, masked, self_attn, out, ld_out,
                              out_bstride, out_col0, copy_inputs, rowstat, scale_out, nbr_idx, nbr_cnt, nbr_cap, coord_dims,
                              math_mode, stream, nullptr);
}

extern "C" int pit_posatt_fwd_job(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                                  int space_dim, int metric, float period,
                                  const float* values, int batch, int dim, long ld_values, long values_bstride,
                                  const float* head, int n_head, int head_is_scale,
                                  const float* stats, float rank_w, int masked, int self_attn,
                                  float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                                  float* rowstat, float* scale_out,
                                  const int* nbr_idx, const int* nbr_cnt, int nbr_cap, int coord_dims, int math_mode,
                                  void* stream, const pit_block_weights_job* job) {
    PIT_ENTER_MATH(math_mode);
    // the rider: inside the attention launch when that is the small candidate-list kernel, else a launch of its own after the
    // attention's (not at all when this call fails)
    WeightsArgs wargs;
    bool wdone = job == nullptr;
    if (job) {
        if (int rcw = fill_weights_args(wargs, job->mesh, job->n_pts, job->space_dim, job->metric, job->period, job->n_layers,
                                        job->heads, job->head_is_scale, job->n_head, job->e, job->q, job->inv, job->rowstat,
                                        job->scale_out)) return rcw;
    }
    auto finish = [&]() -> int {
        if (wdone) return 0;
        wdone = true;
        return pit_block_weights(job->mesh, job->n_pts, job->space_dim, job->metric, job->period, job->n_layers, job->heads,
                                 job->head_is_scale, job->n_head, job->e, job->q, job->inv, job->rowstat, job->scale_out, stream);
    };
    AttArgs a;
    int rc = fill_common(a, mesh_out, mesh_in, mesh_batch, n_out, n_in, space_dim, metric, period, values, batch,
                         dim, ld_values, values_bstride, head, n_head, head_is_scale, coord_dims);
    if (rc) return rc;
    if (!out || !rowstat) return PIT_ERR_NULL;
    if (coord_dims > 0 && (copy_inputs || !(nbr_idx && nbr_cnt && masked))) return PIT_ERR_UNSUPPORTED;   // candidate-list kernels only
    if (!stats && (masked || !self_attn)) return PIT_ERR_NULL;
    if (copy_inputs && n_out != n_in) return PIT_ERR_SIZE;
    a.stats = stats; a.rank_w = rank_w; a.masked = masked;
    a.out = out; a.ld_out = ld_out; a.out_bstride = out_bstride; a.out_col0 = out_col0; a.copy_inputs = copy_inputs;
    a.rowstat = rowstat; a.scale_out = scale_out;
    a.out16 = (math_mode & PIT_IO_OUT_BF16) ? 1 : 0;
    if (a.out16 && (copy_inputs || !(nbr_idx && nbr_cnt && masked))) return PIT_ERR_UNSUPPORTED;   // candidate-list kernels only
    if (math_mode & ~(0xff | PIT_IO_OUT_BF16 | PIT_ATT_UNION)) return PIT_ERR_UNSUPPORTED;
    if (nbr_idx && nbr_cnt && masked && (math_mode & PIT_ATT_UNION) && !copy_inputs && (head_is_scale || scale_out)) {
        SparseArgs spu{nbr_idx, nbr_cnt, nbr_cap, nullptr, nullptr, 0};
        if (union_ok(a, spu) && aligned_rows(out, ld_out, out_bstride, out_col0, a.out16 ? 8 : 4)) {   // coherent row ordering (the caller's plan says so): union tiles
            if (!head_is_scale) {
                if (int rc2 = pit_head_scale(head, n_head, scale_out, stream)) return rc2;
                a.head = scale_out; a.head_is_scale = 1;
                a.scale_out = nullptr;
            }
            launch_union<0>(a, spu, (hipStream_t)stream);
            PIT_CHECK_LAUNCH();
            return finish();
        }
    }
    if (nbr_idx && nbr_cnt && masked) {
        SparseArgs sp{nbr_idx, nbr_cnt, nbr_cap, nullptr, nullptr, 0};
        if (!head_is_scale && scale_out && (long)mesh_batch * n_out > 8192) {
            // one wave per row: evaluate c once up front instead of fp64 sin/tan in every wave
            if (int rc2 = pit_head_scale(head, n_head, scale_out, stream)) return rc2;
            a.head = scale_out; a.head_is_scale = 1;
        }
        launch_sparse_rows<0>(a, sp, (hipStream_t)stream, nullptr, &wdone, job ? &wargs : nullptr);
    } else {
        launch_rows<0>(a, (hipStream_t)stream);
    }
    PIT_CHECK_LAUNCH();
    return finish();
}

extern "C" int pit_posatt_bwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                              int space_dim, int metric, float period,
                              const float* values, int batch, int dim, long ld_values, long values_bstride,
                              const float* head, int n_head, int head_is_scale, const float* scale,
                              const float* rowstat, int masked,
                              const float* d_out, long ld_dout, long dout_bstride, int out_col0,
                              float* d_values, long ld_dvalues, long dvalues_bstride, int add_residual,
                              float* d_head, int accumulate_head, double* workspace,
                              const int* nbr_idx, const int* nbr_cnt, int nbr_cap, int nbr_complete,
                              const int* rev_ptr, const int* rev_row, const pit_mlp_params_job* rider,
                              int coord_dims, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    AttArgs a;
    int rc = fill_common(a, mesh_out, mesh_in, mesh_batch, n_out, n_in, space_dim, metric, period, values, batch,
                         dim, ld_values, values_bstride, head, n_head, head_is_scale, coord_dims);
    if (rc) return rc;
    // the postponed weight-gradient reductions: inside the attention launch when it can carry them, else by
    // the call the caller postponed (after the attention launches; not at all when this call fails)
    struct Rider {
        const pit_mlp_params_job* job; void* stream; bool done;
        int finish() {
            if (!job || done) return 0;
            done = true;
            return pit_mlp_bwd_params(job->x, job->ldx, job->rows, job->n0, job->n1, job->n2, job->h, job->out_gelu,
                                      job->d_y, job->ld_dy, job->d_w1, job->d_b1, job->d_w2, job->d_b2,
                                      job->accumulate, job->scratch, job->math_mode, stream);
        }
    } rd{rider, stream, false};
    if (coord_dims > 0 && (add_residual || !(nbr_idx && nbr_cnt && masked) || (d_values && !(rev_ptr && rev_row))))
        return PIT_ERR_UNSUPPORTED;                             // candidate-list kernels only
    if (!rowstat || !d_out || !workspace) return PIT_ERR_NULL;
    if (add_residual && n_out != n_in) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    if (scale) { a.head = scale; a.head_is_scale = 1; }     // c saved by the forward: no fp64 sin/tan here
    a.masked = masked; a.rowstat = const_cast<float*>(rowstat);
    a.d_out = d_out; a.ld_dout = ld_dout; a.dout_bstride = dout_bstride; a.out_col0 = out_col0;
    a.d_values = d_values; a.ld_dvalues = ld_dvalues; a.dvalues_bstride = dvalues_bstride;
    a.add_residual = add_residual; a.dscale_acc = workspace;
    if (math_mode & ~(0xff | PIT_IO_DOUT_BF16 | PIT_ATT_UNION)) return PIT_ERR_UNSUPPORTED;
    a.dout16 = (math_mode & PIT_IO_DOUT_BF16) ? 1 : 0;
    {
        const int width = out_col0 + n_head * dim;
        const unsigned long long de = (unsigned long long)(batch - 1) * dout_bstride + (unsigned long long)(n_out - 1) * ld_dout + width;
        const unsigned long long db = de * (a.dout16 ? 2ull : 4ull);
        if (db > PIT_MAX_BUFFER_BYTES) return PIT_ERR_UNSUPPORTED;
        a.dout_bytes = (unsigned)db;
        a.dout_elems = (unsigned)de;
    }
    const bool sparse = masked && nbr_idx && nbr_cnt;
    SparseArgs sp{nbr_idx, nbr_cnt, nbr_cap, rev_ptr, rev_row, (long)n_out * nbr_cap};
    const bool union_form = sparse && (math_mode & PIT_ATT_UNION) && !add_residual && a.head_is_scale && union_ok(a, sp) &&
                            aligned_rows(d_out, ld_dout, dout_bstride, out_col0, a.dout16 ? 8 : 4) &&
                            (!d_values || (ld_dvalues == dim && dvalues_bstride == (long)n_in * dim));
    if (a.dout16 && (!sparse || add_residual || (d_values && !(rev_ptr && rev_row) && !union_form) || dim % 2 || out_col0 % 2 || ld_dout % 2 ||
                     dout_bstride % 2 || (reinterpret_cast<uintptr_t>(d_out) & 3)))
        return PIT_ERR_UNSUPPORTED;                             // candidate-list kernels only, column pairs
    if (union_form) {
        // union tiles (coherent row ordering): d(scale) as a dense contraction over each 16-row tile's union of keys
        if (d_head) {
            a.d_head = d_head; a.dhead_src = head; a.dhead_is_scale = head_is_scale;
            a.accumulate_head = (accumulate_head & PIT_HEAD_ACCUMULATE) ? 1 : 0;
            a.nslots = PIT_DSCALE_SLOTS;
            launch_union<1>(a, sp, s);
            PIT_CHECK_LAUNCH();
            if (!(accumulate_head & PIT_HEAD_DEFER)) {
                hipLaunchKernelGGL(posatt_dhead_finish, dim3(n_head), dim3(256), 0, s, a);
                PIT_CHECK_LAUNCH();
            }
        }
        if (d_values && rev_ptr && rev_row) {
            // d(values) from the transposed lists when the plan carries them (the same bits on every run); the host builds
            // plans of union-tile kinds without them (ops.MeshPlan._wants_reverse_lists) and the tiles supply d(values)
            launch_sparse_cols(a, sp, nbr_complete != 0, s);
            PIT_CHECK_LAUNCH();
        } else if (d_values) {
            const long nv = (long)batch * n_in * dim;
            hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)std::min<long>((nv + 1023) / 1024, 2048L)), dim3(256), 0, s, d_values, nv);
            PIT_CHECK_LAUNCH();
            launch_union_dv(a, sp, s);
            PIT_CHECK_LAUNCH();
        }
        return rd.finish();
    }
    if (d_head) {
        a.d_head = d_head; a.dhead_src = head; a.dhead_is_scale = head_is_scale;
        a.accumulate_head = (accumulate_head & PIT_HEAD_ACCUMULATE) ? 1 : 0;
        const bool defer = (accumulate_head & PIT_HEAD_DEFER) != 0;     // caller drains later (pit_posatt_dhead_finish)
        const long approx_wgs = sparse ? ((long)mesh_batch * n_out + 3) / 4 * std::max(1, a.ncols / 512) * n_head
                                       : (long)((n_out + 31) / 32) * n_head * mesh_batch * std::max(1, a.ncols / 128);
        // accumulator slots in use: ~32 adds per slot keeps the fp64 atomics uncontended while the
        // drain (one exchange per slot and head) stays short for small launches
        int ns = 32;
        while (ns < PIT_DSCALE_SLOTS && (long)ns * 32 < approx_wgs * 4) ns <<= 1;
        a.nslots = ns;
        bool paired = false;
        if (d_values) {                                                // d(scale) + d(values) in one launch
            if (!sparse) paired = launch_bwd_pair(a, s, rider, &rd.done);
            else if (rev_ptr && rev_row)
                paired = launch_sparse_bwd_pair(a, sp, nbr_complete != 0, s, rider, &rd.done);
        }
        if (!paired) {
            if (sparse) launch_sparse_rows<1>(a, sp, s, rd.done ? nullptr : rider, &rd.done);
            else launch_rows<1>(a, s);
        }
        PIT_CHECK_LAUNCH();
        if (!defer) {
            hipLaunchKernelGGL(posatt_dhead_finish, dim3(n_head), dim3(256), 0, s, a);
            PIT_CHECK_LAUNCH();
        }
        if (paired) return rd.finish();
    }
    if (d_values) {
        if (sparse && rev_ptr && rev_row) launch_sparse_cols(a, sp, nbr_complete != 0, s); else launch_cols(a, s);
        PIT_CHECK_LAUNCH();
    }
    return rd.finish();
}

#ifdef PIT_STAMPS
extern "C" int pit_debug_read_stamps(unsigned long long* host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(pit_dbg_stamps), 64 * sizeof(unsigned long long));
}
#endif

extern "C" int pit_posatt_dhead_finish(int n_layers, double* const* workspaces, float* const* d_heads,
                                       const float* const* heads, const float* const* scales, const int* n_heads,
                                       const int* flags, const pit_mlp_params_job* rider, void* stream) {
    auto run_rider = [&]() -> int {
        if (!rider) return 0;
        return pit_mlp_bwd_params(rider->x, rider->ldx, rider->rows, rider->n0, rider->n1, rider->n2, rider->h, rider->out_gelu,
                                  rider->d_y, rider->ld_dy, rider->d_w1, rider->d_b1, rider->d_w2, rider->d_b2, rider->accumulate,
                                  rider->scratch, rider->math_mode, stream);
    };
    if (n_layers <= 0) return run_rider();
    if (n_layers > FINISH_MAX_LAYERS) return PIT_ERR_SIZE;
    if (!workspaces || !d_heads || !heads || !scales || !n_heads || !flags) return PIT_ERR_NULL;
    FinishBatch fb;
    fb.n = n_layers;
    int total = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!workspaces[l] || !d_heads[l] || !heads[l]) return PIT_ERR_NULL;
        if (n_heads[l] <= 0) return PIT_ERR_SIZE;
        fb.ws[l] = workspaces[l]; fb.d_head[l] = d_heads[l]; fb.head[l] = heads[l]; fb.scale[l] = scales[l];
        fb.n_head[l] = n_heads[l]; fb.flags[l] = flags[l];
        fb.wg_base[l] = total;
        total += n_heads[l];
    }
    fb.wg_base[n_layers] = total;
    pit_detail::DwPair dw;
    if (rider && pit_detail::plan_dw_pair(*rider, 4, &dw)) {           // small enough to share the launch
        hipLaunchKernelGGL(posatt_dhead_finish_dw, dim3((unsigned)(total + dw.n1 + dw.n2)), dim3(256), DW_SMEM_4WAVES,
                           (hipStream_t)stream, fb, total, dw);
        PIT_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(posatt_dhead_finish_batch, dim3(total), dim3(256), 0, (hipStream_t)stream, fb);
    PIT_CHECK_LAUNCH();
    return run_rider();
}
