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
//   * forward (MODE 0): O_h = (E_h X) / rowsum_h, rowsum and mbar = sum_j P m from the fp32 weights; backward d(values) (MODE 1):
//     E is symmetric, so d(values) = residual + sum_h E_h G_h is the same contraction with both heads accumulating into ONE set of
//     tiles; d(scale) (MODE 2): -(sum G_h . (E_h (m - mbar_h)) X), fp64 partial sums into the layer's slots.
// Distances, c, rowsum, mbar and the d(scale) reduction stay fp32 / fp64; only MFMA operands are bf16.  The fp32 mode never comes here.
#include "pit_common.h"
#include "pit_block_dev.h"

namespace {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int KC = 32;       // keys per step (one v_mfma_f32_16x16x32_bf16 k)
constexpr int PADE = 8;      // bf16 pad of an LDS tile row

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
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void* p) { return make_rsrc(p, 0x7ffffff0u); }
constexpr unsigned OOB = 0x7ffffff8u;

__device__ __forceinline__ v8s_t frag_tr(const unsigned short* t, int pitch, int n0, int l15, int kq) {
    const unsigned short* a0 = t + (8 * kq + (l15 >> 2)) * pitch + n0 + 4 * (l15 & 3);
    typedef v4s_t __attribute__((address_space(3))) * lds_v4;
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0 + 4 * pitch));
    v8s_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
__device__ __forceinline__ f32x4_t mma(v8s_t a, v8s_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, a), __builtin_bit_cast(v8bf_t, b), c, 0, 0, 0);
}

// 32 keys x DIM columns of a bf16 row-major tensor (rows beyond `nrows` load zeros): DIM * 32 / 8 / 256 16-byte pieces per thread
template <int DIM> struct Tile { u32x4_t v[DIM * KC / 8 / 256]; };
template <int DIM>
__device__ __forceinline__ void tile_load(const unsigned short* src, long nrows, int k0, int tid, Tile<DIM>& t) {
    const __amdgpu_buffer_rsrc_t r = wide_rsrc(src);
#pragma unroll
    for (int u = 0; u < DIM * KC / 8 / 256; ++u) {
        const int e = tid + 256 * u, k = e / (DIM / 8), c = e % (DIM / 8);
        const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((k0 + k) < nrows ? (unsigned)((((long)(k0 + k)) * DIM + 8 * c) * 2) : OOB), 0, 0);
        t.v[u] = u32x4_t{(unsigned)q.x, (unsigned)q.y, (unsigned)q.z, (unsigned)q.w};
    }
}
template <int DIM>
__device__ __forceinline__ void tile_park(unsigned short* dst, int tid, const Tile<DIM>& t) {
#pragma unroll
    for (int u = 0; u < DIM * KC / 8 / 256; ++u) {
        const int e = tid + 256 * u, k = e / (DIM / 8), c = e % (DIM / 8);
        *reinterpret_cast<u32x4_t*>(dst + k * (DIM + PADE) + 8 * c) = t.v[u];
    }
}

// MODE 0: out; MODE 1: d(values); MODE 2: d(scale).  NB = B tensors contracted per step (MODE 1: one per head), NA = accumulator sets
template <int H, int DIM, int MODE, bool PERIODIC>
__global__ __launch_bounds__(256) void satt_kernel(SattArgs g) {
    constexpr int NCT = DIM / 16, TP = DIM + PADE, NB = (MODE == 1) ? H : 1, NA = (MODE == 1) ? 1 : H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* keys = reinterpret_cast<float4*>(smem_raw);                        // [Lp] key coordinates (zero-padded to 3)
    const int Lp = (g.L + 2 * KC - 1) / (2 * KC) * (2 * KC);       // whole rounds of two steps: no conditional step (its loads would cost the counted waits)
    unsigned short* tb = reinterpret_cast<unsigned short*>(keys + Lp);         // [2][NB][KC][TP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    int b, t;
    if (!slab_of_xcd((int)blockIdx.x, g.batch, g.tiles, b, t)) return;
    const int mb = g.mesh_batch == 1 ? 0 : b;
    const int row = t * 64 + wave * 16 + l15;                                  // the A-fragment row of this lane
    const int rowc = row < g.L ? row : g.L - 1;
    const float* mesh = g.mesh + (long)mb * g.L * g.sdim;
    // B tensors of this sample
    const unsigned short* bsrc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) bsrc[q] = g.b16 + ((long)b * NB + q) * g.L * DIM;
    Tile<DIM> reg[2][NB];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < NB; ++q) tile_load<DIM>(bsrc[q], g.L, KC * s, tid, reg[s][q]);
    // key coordinates -> LDS, this lane's row point, head scales, saved row statistics
    for (int j = tid; j < Lp; j += 256) {
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < g.L) {
            p.x = mesh[(long)j * g.sdim];
            if (g.used > 1) p.y = mesh[(long)j * g.sdim + 1];
            if (g.used > 2) p.z = mesh[(long)j * g.sdim + 2];
        }
        keys[j] = p;
    }
    const float rx = mesh[(long)rowc * g.sdim], ry = g.used > 1 ? mesh[(long)rowc * g.sdim + 1] : 0.0f,
                rz = g.used > 2 ? mesh[(long)rowc * g.sdim + 2] : 0.0f;
    float c[H], mbar[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        c[h] = g.head_is_scale ? g.head[h] : head_scale_from_lmda(g.head[h]);
        mbar[h] = (MODE == 2) ? g.rowstat_r[(((long)mb * H + h) * g.L + rowc) * 4 + 3] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) tile_park<DIM>(tb + q * KC * TP, tid, reg[0][q]);
#pragma unroll
    for (int q = 0; q < NB; ++q) tile_load<DIM>(bsrc[q], g.L, 2 * KC, tid, reg[0][q]);
    f32x4_t acc[NA][NCT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[a][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float rs[H], sm[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { rs[h] = 0.0f; sm[h] = 0.0f; }
    const int nsteps = Lp / KC;
    // step s contracts keys [32 s, 32 s + 32): its B tile(s) sit in buffer s & 1; register set (s + 1) & 1 holds tile s + 1 (parked
    // now, into the buffer step s - 1 just left) and is re-loaded with tile s + 3; two steps per round: set indices are compile-time
#define PIT_SATT_STEP(s_, j_)                                                                                          \
    do {                                                                                                              \
        __syncthreads();                                                                                              \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) tile_park<DIM>(tb + ((((s_) + 1) & 1) * NB + q) * KC * TP, tid, reg[((j_) + 1) & 1][q]); \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) tile_load<DIM>(bsrc[q], g.L, KC * ((s_) + 3), tid, reg[((j_) + 1) & 1][q]); \
        v8s_t af[H];                                                                                                  \
        {                                                                                                             \
            float w_[H][8];                                                                                           \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                           \
                const int j = KC * (s_) + 8 * kq + e;                                                                 \
                const float4 kp = keys[j];                                                                            \
                const float m = sq_dist3t<PERIODIC>(rx, ry, rz, kp.x, kp.y, kp.z, g.period);                          \
                _Pragma("unroll") for (int h = 0; h < H; ++h) {                                                       \
                    float ev = (j < g.L) ? __expf(-__fmul_rn(m, c[h])) : 0.0f;                                        \
                    if (MODE == 0) { rs[h] += ev; sm[h] = fmaf(ev, m, sm[h]); }                                       \
                    if (MODE == 2) ev *= (m - mbar[h]);                                                               \
                    w_[h][e] = ev;                                                                                    \
                }                                                                                                     \
            }                                                                                                         \
            _Pragma("unroll") for (int h = 0; h < H; ++h)                                                             \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) af[h][e] = bf16_bits(w_[h][e]);                         \
        }                                                                                                             \
        const unsigned short* tile_ = tb + (((s_) & 1) * NB) * KC * TP;                                               \
        _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) {                                                          \
            if (MODE == 1) {                                                                                          \
                _Pragma("unroll") for (int h = 0; h < H; ++h)                                                         \
                    acc[0][ct] = mma(af[h], frag_tr(tile_ + h * KC * TP, TP, 16 * ct, l15, kq), acc[0][ct]);          \
            } else {                                                                                                  \
                const v8s_t bf_ = frag_tr(tile_, TP, 16 * ct, l15, kq);                                               \
                _Pragma("unroll") for (int h = 0; h < H; ++h) acc[h][ct] = mma(af[h], bf_, acc[h][ct]);               \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    for (int sb = 0; sb < nsteps; sb += 2) {
        PIT_SATT_STEP(sb, 0);
        PIT_SATT_STEP(sb + 1, 1);
    }
#undef PIT_SATT_STEP
    // ---- epilogues.  Accumulator register i of this lane is row 4 kq + i of the wave's 16, column 16 ct + l15; the per-row
    // quantities live on the lanes whose l15 is the row (summed over the four key quarters)
    const int r0 = t * 64 + wave * 16;
    if (MODE == 0) {
        float inv[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            rs[h] += __shfl_xor(rs[h], 16, 64); rs[h] += __shfl_xor(rs[h], 32, 64);
            sm[h] += __shfl_xor(sm[h], 16, 64); sm[h] += __shfl_xor(sm[h], 32, 64);
            inv[h] = 1.0f / rs[h];
            if (kq == 0 && row < g.L && (g.mesh_batch > 1 || b == 0)) {
                float4 st;
                st.x = 3.0e38f; st.y = 0.0f; st.z = inv[h]; st.w = sm[h] * inv[h];
                *reinterpret_cast<float4*>(g.rowstat + (((long)mb * H + h) * g.L + row) * 4) = st;
            }
        }
        if (blockIdx.x == 0 && tid < H && g.scale_out) g.scale_out[tid] = c[tid];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float iv = __shfl(inv[h], 4 * kq + i, 64);
                const int n = r0 + 4 * kq + i;
                if (n < g.L) {
                    float* dst = g.out + (long)b * g.out_bstride + (long)n * g.ld_out + g.out_col0 + h * DIM + l15;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) dst[16 * ct] = acc[h][ct][i] * iv;
                }
            }
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = r0 + 4 * kq + i;
            if (n < g.L) {
                float* dst = g.d_values + (long)b * g.dv_bstride + (long)n * g.ld_dv + l15;
                const float* res = g.d_out + (long)b * g.dout_bstride + (long)n * g.ld_dout + l15;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) dst[16 * ct] = acc[0][ct][i] + (g.add_residual ? res[16 * ct] : 0.0f);
            }
        }
    } else {
        double part[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            part[h] = 0.0;
            const float inv_l = g.rowstat_r[(((long)mb * H + h) * g.L + rowc) * 4 + 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float iv = __shfl(inv_l, 4 * kq + i, 64);
                const int n = r0 + 4 * kq + i;
                const __amdgpu_buffer_rsrc_t rd = wide_rsrc(g.d_out);
                float s = 0.0f;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const float dv = buf_load(rd, n < g.L ? (unsigned)(((long)b * g.dout_bstride + (long)n * g.ld_dout + g.out_col0 + h * DIM + 16 * ct + l15) * 4) : OOB);
                    s = fmaf(acc[h][ct][i], dv, s);
                }
                part[h] += (double)s * (double)iv;
            }
            part[h] = wave_sum_d(part[h]);
        }
        __shared__ double wred[4][2];
        if (lane == 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) wred[wave][h] = part[h];
        }
        __syncthreads();
        if (tid < H) {
            const double tot = wred[0][tid] + wred[1][tid] + wred[2][tid] + wred[3][tid];
            atomicAdd(g.dscale + (long)tid * PIT_DSCALE_SLOTS + ((int)blockIdx.x & (PIT_DSCALE_SLOTS - 1)), -tot);
        }
    }
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

size_t satt_smem(int L, int dim, int nb) { return (size_t)((L + 2 * KC - 1) / (2 * KC) * (2 * KC)) * 16 + (size_t)2 * nb * KC * (dim + PADE) * 2; }
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int H, int DIM, int MODE>
void launch_satt(const SattArgs& g, int periodic, hipStream_t s) {
    const dim3 grid((unsigned)slab_grid(g.batch, g.tiles));
    const size_t sm = satt_smem(g.L, DIM, MODE == 1 ? H : 1);
    if (periodic) {
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(satt_kernel<H, DIM, MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), true);
        (void)once;
        hipLaunchKernelGGL((satt_kernel<H, DIM, MODE, true>), grid, dim3(256), sm, s, g);
    } else {
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(satt_kernel<H, DIM, MODE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), true);
        (void)once;
        hipLaunchKernelGGL((satt_kernel<H, DIM, MODE, false>), grid, dim3(256), sm, s, g);
    }
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

// out[b, n, out_col0 + h*dim + d] = sum_j softmax_j(-c_h m[n, j]) values[b, j, d]; copy_inputs: out[b, n, 0:dim] = values.  x16: scratch of
// batch*n_pts*dim bf16 (kept by the caller for pit_satt_bwd); rowstat (mesh_batch, n_head, n_pts, 4), scale_out (n_head) as pit_posatt_fwd.
extern "C" int pit_satt_fwd(const float* mesh, int mesh_batch, int n_pts, int space_dim, int metric, float period,
                            const float* values, long ld_values, long values_bstride, int batch, int dim,
                            const float* head, int n_head, int head_is_scale, unsigned short* x16,
                            float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                            float* rowstat, float* scale_out, void* stream) {
    if (!values || !head || !x16 || !out || !rowstat) return PIT_ERR_NULL;
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
    hipLaunchKernelGGL(satt_prep_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, s, p, 0);
    PIT_CHECK_LAUNCH();
    g.head = head; g.head_is_scale = head_is_scale; g.b16 = x16; g.out = out; g.ld_out = ld_out; g.out_bstride = out_bstride;
    g.out_col0 = out_col0; g.rowstat = rowstat; g.scale_out = scale_out;
    dispatch_satt<0>(g, metric != PIT_METRIC_EUCLID, s);
    PIT_CHECK_LAUNCH();
    return 0;
}

// Backward: d_values[b, j, :] = (add_residual ? d_out[b, j, 0:dim] : 0) + sum_h sum_n P_h[n, j] d_out[b, n, out_col0 + h*dim + :] (NULL: not
// needed) and the layer's d(scale) accumulators (PIT_HEAD_DEFER convention; NULL: not needed).  scale: the c of the forward; x16: the
// forward's; g16: scratch of batch*n_head*n_pts*dim bf16.
extern "C" int pit_satt_bwd(const float* mesh, int mesh_batch, int n_pts, int space_dim, int metric, float period,
                            int batch, int dim, const float* scale, int n_head, const float* rowstat,
                            const unsigned short* x16, unsigned short* g16,
                            const float* d_out, long ld_dout, long dout_bstride, int out_col0,
                            float* d_values, long ld_dvalues, long dvalues_bstride, int add_residual,
                            double* dscale, void* stream) {
    if (!scale || !rowstat || !x16 || !g16 || !d_out || (!d_values && !dscale)) return PIT_ERR_NULL;
    if (!pit_satt_supported(n_pts, n_head, dim, batch, mesh_batch) || space_dim < 1 || space_dim > 3) return PIT_ERR_UNSUPPORTED;
    if (ld_dout % 4 || dout_bstride % 4 || out_col0 % 4 || !al16(d_out) || !al16(g16) || !al16(x16)) return PIT_ERR_SIZE;
    SattArgs g;
    if (int rc = satt_fill(g, mesh, mesh_batch, n_pts, space_dim, metric, period, batch, n_head, dim)) return rc;
    hipStream_t s = (hipStream_t)stream;
    g.head = scale; g.head_is_scale = 1; g.rowstat_r = rowstat; g.d_out = d_out; g.ld_dout = ld_dout; g.dout_bstride = dout_bstride;
    g.out_col0 = out_col0;
    if (d_values) {
        PrepArgs p = PrepArgs();
        p.src = d_out; p.ld = ld_dout; p.bstride = dout_bstride; p.col0 = out_col0; p.batch = batch; p.L = n_pts; p.dim = dim; p.n_head = n_head;
        p.mesh_batch = mesh_batch; p.dst = g16; p.rowstat = rowstat;
        const long total = (long)batch * n_head * n_pts * (dim / 4);
        hipLaunchKernelGGL(satt_prep_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, s, p, 1);
        PIT_CHECK_LAUNCH();
        g.b16 = g16; g.d_values = d_values; g.ld_dv = ld_dvalues; g.dv_bstride = dvalues_bstride; g.add_residual = add_residual;
        dispatch_satt<1>(g, metric != PIT_METRIC_EUCLID, s);
        PIT_CHECK_LAUNCH();
    }
    if (dscale) {
        g.b16 = x16; g.dscale = dscale;
        dispatch_satt<2>(g, metric != PIT_METRIC_EUCLID, s);
        PIT_CHECK_LAUNCH();
    }
    return 0;
}
