// Pointwise MLP (kaiming_mlp, pit.py:13-26) forward and backward on fp32 MFMA.
//
// One GEMM template (v_mfma_f32_32x32x2_f32, exact fp32 fma chains) with strided operand
// views and fused prologue / epilogues covers every contraction of the layer:
//   forward   Z1 = X W1^T + b1, H = gelu(Z1);   Y = H W2^T + b2 (optionally gelu, Z2 kept)
//   backward  dZ2 = dY * gelu'(Z2) (prologue, when the layer ends in a gelu)
//             dZ1 = (dZ2 W2) * gelu'(Z1);  dX = dZ1 W1;
//             dW2 += dZ2^T H, db2 += colsum(dZ2);  dW1 += dZ1^T X, db1 += colsum(dZ1)
//
// The PiT layers are small (rows 2k-80k, widths 1-768): at these sizes a GEMM is bound by
// dependent-latency chains, not by MFMA rate, so the kernel is organised for memory-level
// parallelism rather than operand reuse:
//   * a workgroup owns one 32 x (32*TN) output tile and its W waves split the K range
//     (in-workgroup split-K, LDS tree reduction at the end) - many short independent chains;
//   * operand fragments go global -> registers directly in MFMA layout, 4 consecutive k per
//     lane (one 16-B load when K is the contiguous axis, four coalesced 128-B row reads
//     otherwise); the k order inside a step is permuted identically for A and B, which a
//     contraction does not care about.  No LDS staging, no barrier in the main loop, so the
//     compiler keeps several steps of loads in flight.
// The weight-gradient GEMMs reduce over the row axis: additionally split over row slabs
// (blockIdx.z) and accumulated with fp32 atomics straight into the destination (which may be
// the parameter's .grad); the bias gradient rides along as a virtual all-ones B column.
#include "pit_common.h"
#include "pit_gemm_rd.h"
#include <cstdlib>
#include <type_traits>

// Diagnostic switches (PIT_NO_* / PIT_LDS_* / PIT_RR_* / PIT_THIN_*: force another kernel family or tile shape for an A/B
// measurement) are compiled in only with -DPIT_EXPERIMENTS (tools/variant_build.sh): the production library reads no environment
// variable on this path, and nobody has to re-measure configurations nobody runs.
#ifdef PIT_EXPERIMENTS
static inline const char* exp_env(const char* name) { return getenv(name); }
#else
static inline const char* exp_env(const char*) { return nullptr; }
#endif

namespace {

template <int TN, int EPI>
__global__ __launch_bounds__(512) void gemm_rd_kernel(GemmArgs g) {
    gemm_rd_body<TN, EPI>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// two independent GEMMs (the dW2 and dW1 reductions of one MLP) in ONE launch: workgroups
// [0, nblk1) run the first, the rest the second; each decodes its own (x, y, z) tile index.
template <int TN, int EPI>
__global__ __launch_bounds__(512) void gemm_rd_pair_kernel(GemmArgs g1, GemmArgs g2, int nblk1, int gx1, int gy1,
                                                           int gx2, int gy2) {
    int id = blockIdx.x;
    if (id < nblk1) {
        gemm_rd_body<TN, EPI>(g1, id % gx1, (id / gx1) % gy1, id / (gx1 * gy1));
    } else {
        id -= nblk1;
        gemm_rd_body<TN, EPI>(g2, id % gx2, (id / gx2) % gy2, id / (gx2 * gy2));
    }
}

// the three GEMMs that follow dZ1 in an MLP backward - dX = dZ1 W1 (EPI_STORE, on the critical path:
// its workgroups come first) and the two weight-gradient reductions - are independent of each
// other: one launch, so the small latency-bound grids share the chip
template <int TN>
__global__ __launch_bounds__(512) void gemm_rd_triple_kernel(GemmArgs gx, GemmArgs g1, GemmArgs g2, int nblkx, int gxx,
                                                             int nblk1, int gx1, int gy1, int gx2, int gy2) {
    int id = blockIdx.x;
    if (id < nblkx) {
        gemm_rd_body<TN, EPI_STORE>(gx, id % gxx, id / gxx, 0);
        return;
    }
    id -= nblkx;
    if (id < nblk1) {
        gemm_rd_body<1, EPI_ATOMIC>(g1, id % gx1, (id / gx1) % gy1, id / (gx1 * gy1));
    } else {
        id -= nblk1;
        gemm_rd_body<1, EPI_ATOMIC>(g2, id % gx2, (id / gx2) % gy2, id / (gx2 * gy2));
    }
}

// ------------------------------------------------------------------------------------
// Small-regime fused MLP forward: Y = W2 gelu(W1 X + b1) + b2 (+ trailing gelu) in ONE launch.
// A workgroup owns 16 rows and all n1 hidden columns (n1/16 waves, one 16x16 tile of
// v_mfma_f32_16x16x4_f32 each): 16-row slabs keep rows/16 workgroups in flight (2048 rows -> 128,
// where the 32-row fused variant of round 1 had 64 and lost to two separate launches), the hidden
// activations never leave the workgroup (LDS) between the two contractions, and the second weight
// matrix is prefetched before the first contraction starts.  Z1, H (and Z2) are still written for the
// backward pass.  Operand fragments go global -> registers, 4 consecutive k per lane and step (the k
// order inside a 16-k step is permuted identically for A and B).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma_16x16x4(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct FusedMlpArgs {
    const float* x; long ldx; int rows, n0, n1, n2;
    const float *w1, *b1, *w2, *b2;
    int out_gelu;
    float *z1, *h, *z2, *y; long ldy;
    unsigned x_bytes, w1_bytes, w2_bytes;
    int x_vec, w1_vec;                   // 16-B fragment loads legal
};

template <int N1, int KS>
__global__ __launch_bounds__(N1 * 4) void mlp_fwd16_kernel(FusedMlpArgs g) {
    constexpr int HP = N1 + 4;                          // LDS pitch of the hidden tile (floats)
    constexpr int S2 = N1 / 16;                         // 16-k steps of the second contraction
    __shared__ __attribute__((aligned(16))) float hs[16 * HP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int row = m0 + l15;
    const bool rvalid = row < g.rows;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.x, g.x_bytes);
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(g.w1, g.w1_bytes);
    const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(g.w2, g.w2_bytes);
    const int c1 = wave * 16 + l15;                     // this lane's column of the hidden layer (B operand row of W1)
    const bool do2 = g.n2 > 4 && wave * 16 < g.n2;      // waves that own an output tile
    const int c2 = wave * 16 + l15;

    // EVERY operand fragment of the first contraction is requested before the first MFMA (KS 16-k steps,
    // compile time): one memory round trip for the whole K range instead of one per 64 k
    float av[KS][4], bv[KS][4];
    const unsigned xbase = (unsigned)row * (unsigned)g.ldx * 4u, wbase = (unsigned)c1 * (unsigned)g.n0 * 4u;
    const bool fast = g.x_vec && g.w1_vec;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk = 16 * s + 4 * kq;
        if (fast) {                                     // n0 % 4 == 0: a quad is in range or out of range as a whole
            const bool ok = kk < g.n0;
            buf_load4(rx, (ok && rvalid) ? xbase + (unsigned)kk * 4u : g.x_bytes, av[s]);
            buf_load4(rw1, ok ? wbase + (unsigned)kk * 4u : g.w1_bytes, bv[s]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = kk + e < g.n0;
                av[s][e] = buf_load(rx, (ok && rvalid) ? xbase + (unsigned)(kk + e) * 4u : g.x_bytes);
                bv[s][e] = buf_load(rw1, ok ? wbase + (unsigned)(kk + e) * 4u : g.w1_bytes);
            }
        }
    }
    // second weight matrix: its latency hides behind the first contraction (full tiles: this lane's B fragments;
    // thin output layer: the quarter of output 0's weight row this lane's row dot uses)
    float w2v[S2][4];
    const bool thin_out = g.n2 <= 4;
#pragma unroll
    for (int s = 0; s < S2; ++s) {
        const unsigned off = thin_out ? (unsigned)(kq * (N1 / 4) + 4 * s) * 4u
                                      : ((unsigned)c2 * (unsigned)N1 + (unsigned)(16 * s + 4 * kq)) * 4u;
        buf_load4(rw2, (thin_out ? wave == 0 : (do2 && c2 < g.n2)) ? off : g.w2_bytes, w2v[s]);
    }

    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e & 1) acc1 = mfma_16x16x4(av[s][e], bv[s][e], acc1);
            else acc0 = mfma_16x16x4(av[s][e], bv[s][e], acc0);
        }
    {   // bias + gelu; Z1 / H to memory (backward) and H to LDS (second contraction)
        const int col = wave * 16 + l15;
        const float bias = g.b1[col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            const float z = acc0[i] + acc1[i] + bias;
            const float hv = gelu_erf(z);
            hs[r * HP + col] = hv;
            if (m0 + r < g.rows) {
                g.z1[(long)(m0 + r) * N1 + col] = z;
                g.h[(long)(m0 + r) * N1 + col] = hv;
            }
        }
    }
    __syncthreads();
    if (g.n2 <= 4) {
        // thin output layer (out_dim <= 4, pit.py:106 `de`): a row dot per output, one wave, no MFMA tile
        if (wave != 0) return;
        const int r = m0 + l15;
        for (int o = 0; o < g.n2; ++o) {
            float part = 0.0f;
#pragma unroll
            for (int k = 0; k < N1 / 4; k += 4) {
                const float4 a = *reinterpret_cast<const float4*>(hs + l15 * HP + kq * (N1 / 4) + k);
                float w[4];
                if (o == 0) {                              // (prefetched at kernel entry; S2 = N1/16 quads = N1/4 floats)
                    w[0] = w2v[k / 4][0]; w[1] = w2v[k / 4][1]; w[2] = w2v[k / 4][2]; w[3] = w2v[k / 4][3];
                } else {
                    const float* wp = g.w2 + (long)o * N1 + kq * (N1 / 4) + k;
                    w[0] = wp[0]; w[1] = wp[1]; w[2] = wp[2]; w[3] = wp[3];
                }
                part += (a.x * w[0] + a.y * w[1]) + (a.z * w[2] + a.w * w[3]);
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (kq == 0 && r < g.rows) {
                float v = part + g.b2[o];
                if (g.out_gelu) { g.z2[(long)r * g.n2 + o] = v; v = gelu_erf(v); }
                g.y[(long)r * g.ldy + o] = v;
            }
        }
        return;
    }
    if (!do2) return;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S2; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(hs + l15 * HP + 16 * s + 4 * kq);
        o0 = mfma_16x16x4(a.x, w2v[s][0], o0);
        o1 = mfma_16x16x4(a.y, w2v[s][1], o1);
        o0 = mfma_16x16x4(a.z, w2v[s][2], o0);
        o1 = mfma_16x16x4(a.w, w2v[s][3], o1);
    }
    if (c2 >= g.n2) return;
    const float bias2 = g.b2[c2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = m0 + 4 * kq + i;
        if (r >= g.rows) continue;
        float v = o0[i] + o1[i] + bias2;
        if (g.out_gelu) { g.z2[(long)r * g.n2 + c2] = v; v = gelu_erf(v); }
        g.y[(long)r * g.ldy + c2] = v;
    }
}

// eligibility + launch; false = use the two-launch path
bool try_launch_mlp_fwd16(const float* x, long ldx, int rows, int n0, int n1, int n2, const float* w1, const float* b1,
                          const float* w2, const float* b2, int out_gelu, float* z1, float* h, float* z2, float* y,
                          long ldy, hipStream_t s) {
    static const bool off = exp_env("PIT_NO_FUSED_MLP") != nullptr;
    if (off) return false;
    if (n1 != 32 && n1 != 64 && n1 != 128) return false;
    if (n2 > 4 && (n2 % 16 != 0 || n2 > n1)) return false;        // full tiles, or the thin (out_dim <= 4) output layer
    if (rows < 256 || (long)rows * n1 * (n0 + n2) > (1L << 27)) return false;    // the LDS-tiled GEMMs take over above
    if (n0 > 256) return false;                                                  // K range held in registers
    const unsigned long long xb = ((unsigned long long)(rows - 1) * ldx + n0) * 4ull;
    if (xb > PIT_MAX_BUFFER_BYTES) return false;
    if ((reinterpret_cast<uintptr_t>(w2) & 15) != 0) return false;               // W2 rows are read as 16-B fragments
    FusedMlpArgs g;
    g.x = x; g.ldx = ldx; g.rows = rows; g.n0 = n0; g.n1 = n1; g.n2 = n2;
    g.w1 = w1; g.b1 = b1; g.w2 = w2; g.b2 = b2; g.out_gelu = out_gelu;
    g.z1 = z1; g.h = h; g.z2 = z2; g.y = y; g.ldy = ldy;
    g.x_bytes = (unsigned)xb; g.w1_bytes = (unsigned)((size_t)n1 * n0 * 4); g.w2_bytes = (unsigned)((size_t)n2 * n1 * 4);
    g.x_vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ldx % 4 == 0;
    g.w1_vec = (reinterpret_cast<uintptr_t>(w1) & 15) == 0 && n0 % 4 == 0;
    const dim3 grid((rows + 15) / 16);
    const int ks = (n0 + 63) / 64 * 4;               // 16-k steps, in groups of 4
#define PIT_F16(N1_) do {                                                                                       \
        if (ks == 4) hipLaunchKernelGGL((mlp_fwd16_kernel<N1_, 4>), grid, dim3(N1_ * 4), 0, s, g);              \
        else if (ks == 8) hipLaunchKernelGGL((mlp_fwd16_kernel<N1_, 8>), grid, dim3(N1_ * 4), 0, s, g);         \
        else if (ks == 12) hipLaunchKernelGGL((mlp_fwd16_kernel<N1_, 12>), grid, dim3(N1_ * 4), 0, s, g);       \
        else hipLaunchKernelGGL((mlp_fwd16_kernel<N1_, 16>), grid, dim3(N1_ * 4), 0, s, g);                     \
    } while (0)
    if (n1 == 32) PIT_F16(32); else if (n1 == 64) PIT_F16(64); else PIT_F16(128);
#undef PIT_F16
    return true;
}

// ------------------------------------------------------------------------------------
// Small-regime fused MLP backward, data path: dZ2 = dY * gelu'(Z2) (or dY), dZ1 = (dZ2 W2) * gelu'(Z1),
// dX = dZ1 W1 in ONE launch on 16-row slabs (the mirror of mlp_fwd16_kernel): the three steps of a slab
// only depend on that slab, dZ2 / dZ1 stay in LDS between them, and every weight fragment is requested at
// kernel entry.  dZ2 (when the layer ends in a gelu) and dZ1 are also written to scratch for the
// weight-gradient reductions, which remain a separate (row-reducing) launch.
struct FusedMlpBwdArgs {
    int rows, n0, n1, n2, out_gelu;
    const float *w1, *w2, *z1, *z2, *d_y; long ld_dy;
    float *d_x; long ld_dx;
    float *dz1, *dz2;                    // scratch: (rows, n1) and (rows, n2)
};

template <int N1, int TPW>                // TPW = dX column tiles (16 wide) per wave
__global__ __launch_bounds__(N1 * 4) void mlp_bwd16_kernel(FusedMlpBwdArgs g) {
    constexpr int NW = N1 / 16;
    constexpr int P1 = N1 + 4;                          // LDS pitch of the dZ1 tile
    constexpr int S1 = N1 / 16;                         // 16-k steps of the dX contraction (K = n1)
    __shared__ __attribute__((aligned(16))) float ds1[16 * P1];
    __shared__ __attribute__((aligned(16))) float ds2[16 * P1];       // dZ2 tile, n2 <= n1 columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int n0 = g.n0, n2 = g.n2;
    const bool thin = n2 <= 4;
    // ---- everything that depends on nothing is requested first: W1 fragments of this wave's dX tiles, the
    // gelu' argument of this wave's dZ1 tile, and (full tiles) the W2 fragments of the dZ1 contraction
    float w1v[TPW][S1][4];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int col = (wave + t * NW) * 16 + l15;     // dX column
#pragma unroll
        for (int s = 0; s < S1; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * s + 4 * kq + e;      // B(k, n) = w1[k][n]: the 16 lanes of a k read 64 consecutive bytes
                w1v[t][s][e] = (col < n0) ? g.w1[(long)k * n0 + col] : 0.0f;
            }
    }
    const int c1 = wave * 16 + l15;                     // this lane's dZ1 column
    float z1v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = m0 + 4 * kq + i;
        z1v[i] = (r < g.rows) ? g.z1[(long)r * N1 + c1] : 0.0f;
    }
    constexpr int S2MAX = N1 / 16;                      // n2 <= n1
    float w2v[S2MAX][4];
#pragma unroll
    for (int s = 0; s < S2MAX; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 16 * s + 4 * kq + e;          // B(k, n) = w2[k][n]
            // (thin output layer: w2[o][c1], o < n2 <= 4, in w2v[0][o] - requested here, used after the barrier)
            const bool want = thin ? (s == 0 && e < n2) : (k < n2);
            w2v[s][e] = want ? g.w2[(long)(thin ? e : k) * N1 + c1] : 0.0f;
        }
    // ---- phase A: dZ2 tile (16 x n2) -> LDS (+ scratch when it differs from dY).  16 n2 <= 16 N1 = 4 elements per
    // thread: all of their loads are issued together with the weight fragments above (as a loop with the loads
    // inside it was up to four dependent memory round trips in front of the first MFMA)
    {
        float dyv[4], z2g[4];
        int rr[4], cc[4];
        bool in[4], ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + u * NW * 64;
            in[u] = idx < 16 * n2;
            rr[u] = in[u] ? idx / n2 : 0;
            cc[u] = in[u] ? idx - rr[u] * n2 : 0;
            ok[u] = in[u] && m0 + rr[u] < g.rows;
            const long row = ok[u] ? m0 + rr[u] : 0;
            dyv[u] = ok[u] ? g.d_y[row * g.ld_dy + cc[u]] : 0.0f;
            z2g[u] = (ok[u] && g.out_gelu) ? g.z2[row * n2 + cc[u]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = dyv[u];
            if (g.out_gelu) {
                v *= gelu_erf_grad(z2g[u]);
                if (ok[u]) g.dz2[(long)(m0 + rr[u]) * n2 + cc[u]] = v;
            }
            if (in[u]) ds2[rr[u] * P1 + cc[u]] = ok[u] ? v : 0.0f;
        }
    }
    __syncthreads();
    // ---- phase B: dZ1 tile of this wave = (dZ2 W2) * gelu'(Z1)
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (thin) {                                         // out_dim <= 4: an outer product per output, no MFMA tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < n2) acc += ds2[(4 * kq + i) * P1 + o] * w2v[0][o];
            a0[i] = acc;
        }
    } else {
        const int steps = n2 / 16;
#pragma unroll
        for (int s = 0; s < S2MAX; ++s) {
            if (s < steps) {
                const float4 a = *reinterpret_cast<const float4*>(ds2 + l15 * P1 + 16 * s + 4 * kq);
                a0 = mfma_16x16x4(a.x, w2v[s][0], a0);
                a1 = mfma_16x16x4(a.y, w2v[s][1], a1);
                a0 = mfma_16x16x4(a.z, w2v[s][2], a0);
                a1 = mfma_16x16x4(a.w, w2v[s][3], a1);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * kq + i;
        const float v = (a0[i] + a1[i]) * gelu_erf_grad(z1v[i]);
        ds1[r * P1 + c1] = v;
        if (m0 + r < g.rows) g.dz1[(long)(m0 + r) * N1 + c1] = v;
    }
    __syncthreads();
    if (!g.d_x) return;
    // ---- phase C: dX tiles of this wave = dZ1 W1
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + t * NW;
        if (tile * 16 >= n0) break;
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S1; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(ds1 + l15 * P1 + 16 * s + 4 * kq);
            o0 = mfma_16x16x4(a.x, w1v[t][s][0], o0);
            o1 = mfma_16x16x4(a.y, w1v[t][s][1], o1);
            o0 = mfma_16x16x4(a.z, w1v[t][s][2], o0);
            o1 = mfma_16x16x4(a.w, w1v[t][s][3], o1);
        }
        const int col = tile * 16 + l15;
        if (col < n0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = m0 + 4 * kq + i;
                if (r < g.rows) g.d_x[(long)r * g.ld_dx + col] = o0[i] + o1[i];
            }
        }
    }
}

bool mlp_bwd16_eligible(int rows, int n0, int n1, int n2) {
    static const bool off = exp_env("PIT_NO_FUSED_MLP_BWD") != nullptr;
    if (off) return false;
    if (n1 != 32 && n1 != 64 && n1 != 128) return false;
    if (n2 > 4 && (n2 % 16 != 0 || n2 > n1)) return false;
    if (rows < 256 || (long)rows * n1 * (n0 + n2) > (1L << 27)) return false;
    return ((n0 + 15) / 16 + n1 / 16 - 1) / (n1 / 16) <= 4;      // n0 tiles per wave
}

// eligibility + launch of the fused data path; false = the separate dZ1 and dX launches
bool try_launch_mlp_bwd16(int rows, int n0, int n1, int n2, const float* w1, const float* w2, const float* z1,
                          const float* z2, int out_gelu, const float* d_y, long ld_dy, float* d_x, long ld_dx,
                          float* dz1, float* dz2, hipStream_t s) {
    if (!mlp_bwd16_eligible(rows, n0, n1, n2)) return false;
    const int nw = n1 / 16;
    const int tiles = (n0 + 15) / 16;
    const int tpw = (tiles + nw - 1) / nw;
    FusedMlpBwdArgs g;
    g.rows = rows; g.n0 = n0; g.n1 = n1; g.n2 = n2; g.out_gelu = out_gelu;
    g.w1 = w1; g.w2 = w2; g.z1 = z1; g.z2 = z2; g.d_y = d_y; g.ld_dy = ld_dy;
    g.d_x = d_x; g.ld_dx = ld_dx; g.dz1 = dz1; g.dz2 = dz2;
    const dim3 grid((rows + 15) / 16);
#define PIT_B16(N1_) do {                                                                                     \
        if (tpw <= 1) hipLaunchKernelGGL((mlp_bwd16_kernel<N1_, 1>), grid, dim3(N1_ * 4), 0, s, g);           \
        else if (tpw == 2) hipLaunchKernelGGL((mlp_bwd16_kernel<N1_, 2>), grid, dim3(N1_ * 4), 0, s, g);      \
        else if (tpw == 3) hipLaunchKernelGGL((mlp_bwd16_kernel<N1_, 3>), grid, dim3(N1_ * 4), 0, s, g);      \
        else hipLaunchKernelGGL((mlp_bwd16_kernel<N1_, 4>), grid, dim3(N1_ * 4), 0, s, g);                    \
    } while (0)
    if (n1 == 32) PIT_B16(32); else if (n1 == 64) PIT_B16(64); else PIT_B16(128);
#undef PIT_B16
    return true;
}

// ------------------------------------------------------------------------------------
// Large-regime GEMM: operands staged through LDS.
//
// The register-direct kernel above has every lane fetch its own fragment: a wave-level 16-B load
// touches 32 different cache lines (32 rows x 32 B), so at scale it is bound by the L1 tag rate,
// not by bytes or MFMA (measured 20-40 TF/s on 5120 x 768 x 256).  Here a 256-thread workgroup
// owns a 128 x 64 output tile and walks K in chunks of 32: both operand chunks are fetched with
// fully coalesced 16-B loads (8 full lines per wave instruction) into registers one chunk ahead,
// written to LDS, and every wave (32 rows x 64 columns, two accumulator tiles) reads its MFMA
// fragments from there.  No split-K inside the workgroup, so no reduction; the weight-gradient
// GEMMs split K over blockIdx.z with fp32 atomics as before.
// An operand is either k-contiguous in memory (KC: LDS [i][k], fragment = one 16-B LDS read) or
// i-contiguous (IC: LDS [k][i], fragment = 4 conflict-free 4-B reads).
// Tile height LBM = 128 (wave = 32 rows x 64 columns) when that still gives >= 2 workgroups per CU,
// else 64 (wave = 32 rows x 32 columns: twice the workgroups for the mid-sized layers).
#ifdef PIT_STAMPS
__device__ unsigned long long pit_mlp_stamps[64];
__device__ unsigned long long pit_mlp_wgrec[4096 * 4];     // per workgroup: entry, exit (100 MHz), HW_ID, XCC_ID
__device__ int pit_mlp_rec_k;                               // gemm_lds_kernel records the launches whose K equals this
#define MREC(slot_, cond_) do { const unsigned lid_ = blockIdx.x + blockIdx.y * gridDim.x;                                 \
    if (threadIdx.x == 0 && (cond_) && lid_ < 4096) {                                                                      \
        pit_mlp_wgrec[lid_ * 4 + (slot_)] = __builtin_amdgcn_s_memrealtime();                                              \
        if ((slot_) == 0) { pit_mlp_wgrec[lid_ * 4 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      \
                            pit_mlp_wgrec[lid_ * 4 + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); } } } while (0)
#else
#define MREC(slot_, cond_) do { } while (0)
#endif
constexpr int LBN = 64, LBK = 32;
constexpr int LPK = LBK + 4;      // row pitch of the [i][k] layouts (floats): 144 B, 16-B aligned, bank-skewed
constexpr int LPN = LBN + 4;      // [k][n]

// AGZ (round 3; k-contiguous A only): the trailing-gelu prologue A(m,k) *= gelu'(a_gz[m][k]) applied while the chunk goes from the
// staging registers to LDS, the product written to a_out by the first column block - was an elementwise pass of its own
// (mul_gelu_grad_kernel: 10 us per MLP backward at 65 536 rows)
template <int LBM, bool A_KC, bool B_KC, int EPI, bool BF, bool AGZ = false>
__global__ __launch_bounds__(256) void gemm_lds_kernel(GemmArgs g) {
    static_assert(!AGZ || A_KC, "the fused prologue reads a_gz with A's k-contiguous pattern");
    constexpr int LPM = LBM + 4;      // [k][m]
    constexpr int TN = (LBM == 128) ? 2 : 1;      // accumulator tiles per wave
    constexpr int PA = LBM / 32;                  // staging passes for A (4 floats per thread and pass)
    // LBM == 32: 32 x 64 tile, the waves pair up on the K axis (each pair member takes two of the four
    // 8-k sub-steps of a chunk and the partner's partial tile is added through LDS at the end):
    // twice the workgroups of the 64-row tile for layers that would otherwise not fill the chip
    constexpr bool KSPLIT = (LBM == 32);
    __shared__ __attribute__((aligned(16))) float As[A_KC ? LBM * LPK : LBK * LPM];
    __shared__ __attribute__((aligned(16))) float Bs[B_KC ? LBN * LPK : LBK * LPN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.remap_gx) {                  // XCD-aware order (GemmArgs::remap_gx)
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        bx = t % g.remap_gx;
        by = (t / g.remap_gx) * 8 + xcd;
        if (by >= g.remap_gy) return;
    }
    const int m0 = by * LBM, n0 = bx * LBN;
    const int wrow = (LBM == 128) ? wave * 32 : (LBM == 64 ? (wave & 1) * 32 : 0);   // this wave's rows / first column in the tile
    const int wcol = (LBM == 128) ? 0 : (LBM == 64 ? (wave >> 1) * 32 : (wave & 1) * 32);
    const int ksel = KSPLIT ? (wave >> 1) : 0;
    const int kbeg = blockIdx.z * g.k_slab;
    const int kend = min(g.K, kbeg + g.k_slab);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, g.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(g.B, g.b_bytes);
    const int n_real = (EPI == EPI_ATOMIC && g.ones_col >= 0) ? g.N - 1 : g.N;   // columns that exist in memory

    MREC(0, g.K == pit_mlp_rec_k && blockIdx.z == 0);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(AGZ ? g.a_gz : g.A, g.a_bytes);
    float sa[PA][4], sb[2][4];         // staging registers: next chunk in flight during the MFMAs
    float sz[AGZ ? PA : 1][4];
    auto gload = [&](int kc) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            unsigned off;
            bool ok;
            if (A_KC) {                // 8 threads cover the 32 k of one row
                const int row = m0 + p * 32 + (tid >> 3), k = kc + (tid & 7) * 4;
                ok = row < g.M && k < kend;
                off = ((unsigned)row * (unsigned)g.a_rs + (unsigned)k) * 4u;
            } else {                   // LBM/4 threads cover the m of one k
                constexpr int TPR = LBM / 4, KPP = 256 / TPR;
                const int k = kc + p * KPP + tid / TPR, m = m0 + (tid % TPR) * 4;
                ok = k < kend && m < g.M;
                off = ((unsigned)k * (unsigned)g.a_cs + (unsigned)m) * 4u;
            }
            buf_load4(ra, ok ? off : g.a_bytes, sa[p]);     // (a trailing-gelu prologue: AGZ, or its own pass before this kernel)
            if constexpr (AGZ) buf_load4(rz, ok ? off : g.a_bytes, sz[p]);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (B_KC) {                // 8 threads cover the 32 k of one column's row
                const int n = n0 + p * 32 + (tid >> 3), k = kc + (tid & 7) * 4;
                const bool ok = n < n_real && k < kend;
                buf_load4(rb, ok ? ((unsigned)n * (unsigned)g.b_cs + (unsigned)k) * 4u : g.b_bytes, sb[p]);
            } else {                   // 16 threads cover the 64 n of one k
                const int k = kc + p * 16 + (tid >> 4), n = n0 + (tid & 15) * 4;
                const bool ok = k < kend && n < n_real;
                buf_load4(rb, ok ? ((unsigned)k * (unsigned)g.b_rs + (unsigned)n) * 4u : g.b_bytes, sb[p]);
            }
        }
    };
    auto lstore = [&](int kc) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            constexpr int TPR = LBM / 4, KPP = 256 / TPR;
            if constexpr (AGZ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sa[p][e] *= gelu_erf_grad(sz[p][e]);
                const int row = m0 + p * 32 + (tid >> 3), k = kc + (tid & 7) * 4;
                if (bx == 0 && row < g.M && k < kend)
                    *reinterpret_cast<float4*>(g.a_out + (long)row * g.a_out_rs + k) = make_float4(sa[p][0], sa[p][1], sa[p][2], sa[p][3]);
            }
            float* dst = A_KC ? As + (p * 32 + (tid >> 3)) * LPK + (tid & 7) * 4
                              : As + (p * KPP + tid / TPR) * LPM + (tid % TPR) * 4;
            *reinterpret_cast<float4*>(dst) = make_float4(sa[p][0], sa[p][1], sa[p][2], sa[p][3]);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float* dst = B_KC ? Bs + (p * 32 + (tid >> 3)) * LPK + (tid & 7) * 4
                              : Bs + (p * 16 + (tid >> 4)) * LPN + (tid & 15) * 4;
            *reinterpret_cast<float4*>(dst) = make_float4(sb[p][0], sb[p][1], sb[p][2], sb[p][3]);
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // the bias gradient (the "ones column" of the register-direct kernel) is the row sum of the A
    // operand: accumulated from the fragments by the first column block, no extra MFMA tile
    const bool want_rowsum = (EPI == EPI_ATOMIC) && g.ones_col >= 0 && bx == 0 && wcol == 0;
    float rsum = 0.0f;

    gload(kbeg);
    for (int kc = kbeg; kc < kend; kc += LBK) {
        lstore(kc);
        __syncthreads();
        if (kc + LBK < kend) gload(kc + LBK);
#pragma unroll
        for (int sub0 = 0; sub0 < (KSPLIT ? 2 : 4); ++sub0) {
            const int sub = KSPLIT ? 2 * ksel + sub0 : sub0;
            const int ks = sub * 8 + half * 4;          // this half-wave's 4 k of the sub-step
            float av[4], bv[TN][4];
            if (A_KC) {
                const float4 q = *reinterpret_cast<const float4*>(As + (wrow + l31) * LPK + ks);
                av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = As[(ks + e) * LPM + wrow + l31];
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (B_KC) {
                    const float4 q = *reinterpret_cast<const float4*>(Bs + (wcol + t * 32 + l31) * LPK + ks);
                    bv[t][0] = q.x; bv[t][1] = q.y; bv[t][2] = q.z; bv[t][3] = q.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[t][e] = Bs[(ks + e) * LPN + wcol + t * 32 + l31];
                }
            }
            if (want_rowsum) rsum += (av[0] + av[1]) + (av[2] + av[3]);
            if (BF) {                  // compile-time: a runtime branch here shuttles the accumulators between register files
                const bf16x4 ap = pack_bf16(av[0], av[1], av[2], av[3]);
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc[t] = mfma_32x32x8_bf16(ap, pack_bf16(bv[t][0], bv[t][1], bv[t][2], bv[t][3]), acc[t]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < TN; ++t) acc[t] = mfma_32x32x2(av[e], bv[t][e], acc[t]);
            }
        }
        __syncthreads();
    }

    if (KSPLIT) {                      // pair reduction: the k-half-1 waves hand their tile to their partners
        float* red = Bs;               // 2 waves x 16 registers x 64 lanes = 8 KB: fits the (idle) B chunk
        if (ksel == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave & 1) * 16 + r) * 64 + lane] = acc[0][r];
        }
        __syncthreads();
        if (ksel == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += red[((wave & 1) * 16 + r) * 64 + lane];
    }
    // epilogue: every (remaining) wave owns its block outright
    if (want_rowsum) {
        rsum += __shfl_xor(rsum, 32);
        const int row = m0 + wrow + l31;
        if (half == 0 && row < g.M) atomicAdd(g.C2 + row, rsum);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + wcol + t * 32 + l31;
        if (col >= n_real) continue;
        float bias = 0.0f;
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bias = g.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wrow + acc_row(r, half);
            if (row >= g.M) continue;
            float v = acc[t][r] + bias;
            if (EPI == EPI_BIAS_GELU) { g.Z[(long)row * g.ldz + col] = v; v = gelu_erf(v); }
            if (EPI == EPI_MUL_GELU_GRAD) v *= gelu_erf_grad(g.G[(long)row * g.ldg + col]);
            if (EPI == EPI_ATOMIC) atomicAdd(g.C + (long)row * g.ldc + col, v);
            else g.C[(long)row * g.ldc + col] = v;
        }
    }
    MREC(1, g.K == pit_mlp_rec_k && blockIdx.z == 0);
}

// ------------------------------------------------------------------------------------
// Row-reducing GEMM of the weight gradients, fp32 (round 3): C[m][n] += sum_k A[k][m] B[k][n] over this workgroup's K
// slab, both operands i-contiguous in memory (A = dZ, B = the layer input; K = rows of the batch >> M, N).
// gemm_lds_kernel<128, false, false, EPI_ATOMIC> ran these at 25-45 % MFMA busy: a 128-row tile whatever M is (M = 64:
// half of the waves multiplied zeros), two barriers and ONE chunk of prefetch per 32 k.  Here
//  * tiles are 64 RM x 64 RN (2 x 2 waves of (32 RM) x (32 RN)), chosen per shape so that no wave multiplies padding;
//    row tile t of a wave = rows {RM l + t} (INTERLEAVED: one LDS read of RM consecutive floats is the lane's A operand
//    for all its row tiles); column tiles stay contiguous - the fp32 atomics of the epilogue run at a fraction of their
//    rate when a wave's 32 lanes are 8 or 12 B apart (measured: 127 us against 30 for the Darcy b=256 layer);
//  * both operand images are [k][i], written by the coalesced 16-B global loads as they are (no transpose, no padding);
//  * LDS is double-buffered: one barrier per chunk, the next chunk's global loads are in flight during the whole
//    contraction of the current one, whose LDS reads run one k-step ahead of the MFMAs;
//  * the K chunks are dealt to the slabs as evenly as whole chunks allow (slab s = chunks [s C / S, (s+1) C / S)) and
//    S x tiles never exceeds the workgroups the chip holds at once - one more and the launch takes two rounds;
//  * the slabs of one tile set are dealt to ONE XCD (grid id -> (slab, tile) below): the operand rows a slab's tiles
//    share come from HBM once.
// Bias gradient = row sums of A, accumulated from the fragments by the first column block (as gemm_lds_kernel).
// BF (PIT_MATH_BF16, fp32-stored operands): the same images, rounded to bf16 (RNE) on the way from LDS to the
// v_mfma_f32_32x32x16_bf16 operands - 16 k per instruction, this half-wave's 8 k of a step = 8 LDS reads + 4 packs per operand.
// (the tile itself: gemm_rr_tile, pit_gemm_rd.h)
template <int RM, int RN, int BK, bool BF = false>
__global__ __launch_bounds__(256) void gemm_rr_kernel(GemmArgs g1, GemmArgs g2, int tx1, int T1, int tx2, int T2, int slabs,
                                                      int nchunks) {
    constexpr int BM = 64 * RM, BN = 64 * RN;
    __shared__ __attribute__((aligned(16))) float As[2 * BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BK * BN];
    // work item w = slab * (T1 + T2) + tile, dealt to the XCDs in CONTIGUOUS ranges (hardware deals workgroup ids round-robin):
    // every XCD gets the same number of items, and the tiles of a slab - which share operand rows - mostly meet in one L2
    const int tiles = T1 + T2, P = tiles * slabs, Q = (P + 7) / 8;
    const int w = (blockIdx.x & 7) * Q + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= Q || w >= P) return;
    const int slab = w / tiles, tile_all = w % tiles;
    const bool first = tile_all < T1;
    const GemmArgs& g = first ? g1 : g2;
    const int tile = first ? tile_all : tile_all - T1, tiles_x = first ? tx1 : tx2;
    const int kbeg = (int)((long)slab * nchunks / slabs) * BK;
    const int kend = min(g.K, (int)((long)(slab + 1) * nchunks / slabs) * BK);
    MREC(0, tiles > 1);
    gemm_rr_tile<RM, RN, BK, BF>(g, tile % tiles_x, tile / tiles_x, kbeg, kend, As, Bs);
    MREC(1, tiles > 1);
}

// ------------------------------------------------------------------------------------
// Large-regime GEMM of the bf16 math mode: operands are rounded to bf16 ONCE, while they are staged
// into LDS, and the contraction runs on v_mfma_f32_32x32x16_bf16 (16 k per instruction, fp32
// accumulation): 1/16 of the matrix-pipe time of the fp32 form and half the LDS bytes.
// Both operand images are [i][k] with 8 consecutive k = one 16-B fragment read (the lane map of the
// instruction: lane (r, h) holds X[r][8h .. 8h+7]); rows are BPK = 40 bf16 = 80 B apart, which makes
// the 16-lane groups of ds_read_b128 and the 8-lane groups of ds_write_b128 hit 64 / 32 different
// banks.  A k-contiguous operand (KC) is staged with coalesced 16-B loads and written as 8-B packed
// quads; an i-contiguous operand (IC) needs the transpose: lane = i, each lane fetches its own 4/8/16
// consecutive k with coalesced 4-B loads (64 consecutive i per wave instruction) and writes whole
// 16-B fragments - no cross-lane movement, no 2-byte LDS stores.
constexpr int BPK = LBK + 8;      // LDS row pitch in bf16 elements
// four consecutive elements of a tensor stored as fp32 or (PIT_IO_*: is16) bf16; idx in elements, 4-element aligned
__device__ __forceinline__ float4 ld4e(const float* base, long idx, int is16) {
    if (is16) {
        const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
        return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u),
                           __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(base + idx);
}
__device__ __forceinline__ void st4e(float* base, long idx, float4 v, int is16) {
    if (is16) {
        uint2 q;
        q.x = pack2_bf16(v.x, v.y); q.y = pack2_bf16(v.z, v.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx) = q;
    } else {
        *reinterpret_cast<float4*>(base + idx) = v;
    }
}
__device__ __forceinline__ float ld1e(const float* base, long idx, int is16) {
    return is16 ? bf16_to_f(reinterpret_cast<const unsigned short*>(base)[idx]) : base[idx];
}
__device__ __forceinline__ void st1e(float* base, long idx, float v, int is16) {
    if (is16) reinterpret_cast<unsigned short*>(base)[idx] = f_to_bf16(v); else base[idx] = v;
}

// IO16: the instantiation that honours the bf16-STORAGE flags of GemmArgs (a16 .. g16); the plain one carries none of that
// code (as runtime-only branches the extra staging registers cost the fp32-storage launches 8-26 %).
// BK = k per LDS chunk (round 3).  With 32 a chunk is 2 MFMAs per wave and tile (64 cycles of matrix pipe) against one
// global round trip of ~1.5 k cycles with a single chunk of prefetch: the mid-sized layers (5120 rows, 24 chunks) were 24
// dependent round trips long - 1-2 % MFMA busy.  128 (tiles up to 64 rows) / 64 (128-row tiles: staging registers) cut the
// number of dependent iterations 4x / 2x; LDS rows are BK + 8 bf16 apart.
template <int LBM, bool A_KC, bool B_KC, int EPI, bool IO16 = false, int BK = 32>
__global__ __launch_bounds__(256) void gemm_bfl_kernel(GemmArgs g) {
    // (IO16: the A operand IS bf16-stored, and in the row-reducing kind B as well - compile-time, so that the fp32 staging
    // registers of the other path do not exist: as runtime flags both sets were live, 196 VGPRs = 2 waves per SIMD)
    constexpr bool a16 = IO16, b16 = IO16 && (EPI == EPI_ATOMIC);
    const bool c16 = IO16 && g.c16, z16 = IO16 && g.z16, g16 = IO16 && g.g16;
    constexpr int TN = (LBM == 128) ? 2 : 1;
    constexpr int PK = BK + 8;                    // LDS row pitch (bf16 elements)
    constexpr int TPR = BK / 4;                   // KC fp32 staging: threads per row (4 floats each)
    constexpr int RPP = 256 / TPR;                // ... rows per pass
    constexpr int PA = (LBM + RPP - 1) / RPP;     // ... passes for A
    constexpr int PB = (LBN + RPP - 1) / RPP;     // ... and for B
    constexpr int KPA = BK / (256 / LBM);         // IC fp32 staging: k per thread for A (lane = row)
    constexpr int KPB = BK / (256 / LBN);         // ... and for B
    constexpr bool KSPLIT = (LBM == 32);
    constexpr int STEPS = BK / 16;                // 16-k MFMA steps per chunk
    static_assert(!KSPLIT || STEPS % 2 == 0, "the wave pairs split the steps of a chunk");
    __shared__ __attribute__((aligned(16))) unsigned short As[LBM * PK];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[LBN * PK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.remap_gx) {                  // XCD-aware order (GemmArgs::remap_gx)
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        bx = t % g.remap_gx;
        by = (t / g.remap_gx) * 8 + xcd;
        if (by >= g.remap_gy) return;
    }
    const int m0 = by * LBM, n0 = bx * LBN;
    const int wrow = (LBM == 128) ? wave * 32 : (LBM == 64 ? (wave & 1) * 32 : 0);
    const int wcol = (LBM == 128) ? 0 : (LBM == 64 ? (wave >> 1) * 32 : (wave & 1) * 32);
    const int ksel = KSPLIT ? (wave >> 1) : 0;
    const int kbeg = blockIdx.z * g.k_slab;
    const int kend = min(g.K, kbeg + g.k_slab);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, g.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(g.B, g.b_bytes);
    const int n_real = (EPI == EPI_ATOMIC && g.ones_col >= 0) ? g.N - 1 : g.N;

    float sa[a16 ? 1 : (A_KC ? PA * 4 : KPA)], sb[b16 ? 1 : (B_KC ? PB * 4 : KPB)];
    // bf16 STORAGE of an operand (PIT_IO_*, wave-uniform flags): its chunk needs no conversion on the way to LDS.
    //   k-contiguous A: BK/8 threads cover the BK k of a row with one 16-B load each;
    //   i-contiguous A (LBM = 128) / B: a lane owns TWO adjacent rows / columns (one 4-B load = the pair at one k), takes
    //   BK/4 (A) / BK/8 (B) consecutive k and writes each row's fragment pieces as 16-B / 8-B LDS stores.
    constexpr int TPR16 = BK / 8, RPP16 = 256 / TPR16;
    constexpr int PA16 = (LBM + RPP16 - 1) / RPP16;
    constexpr int KA2 = BK / 4, KB2 = BK / 8;     // 4-B pair loads per thread (A: 64 row pairs x 4 k groups; B: 32 pairs x 8)
    uint4 sa16[(a16 && A_KC) ? PA16 : 1];
    unsigned sa2[(a16 && !A_KC) ? KA2 : 1], sb2[b16 ? KB2 : 1];
    // the bias gradient (virtual ones column) = row sums of the A operand, exact fp32 from the staging
    // registers of the first column block (A is i-contiguous in every row-reducing GEMM)
    const bool want_rowsum = (EPI == EPI_ATOMIC) && !A_KC && g.ones_col >= 0 && bx == 0;
    float rsum = 0.0f, rsum_hi = 0.0f;
    auto gload = [&](int kc) {
        if constexpr (a16 && A_KC) {
#pragma unroll
            for (int p = 0; p < PA16; ++p) {
                const int rl = p * RPP16 + tid / TPR16, row = m0 + rl, k = kc + (tid % TPR16) * 8;
                const bool ok = rl < LBM && row < g.M && k < kend;
                const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? (int)(((unsigned)row * (unsigned)g.a_rs + (unsigned)k) * 2u) : (int)g.a_bytes, 0, 0);
                sa16[p] = make_uint4((unsigned)q.x, (unsigned)q.y, (unsigned)q.z, (unsigned)q.w);
            }
        } else if constexpr (a16 && !A_KC) {
            const int m = m0 + 2 * (tid & 63), k0 = kc + (tid >> 6) * KA2;
#pragma unroll
            for (int e = 0; e < KA2; ++e) {
                const bool ok = m < g.M && k0 + e < kend;
                sa2[e] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, ok ? (int)(((unsigned)(k0 + e) * (unsigned)g.a_cs + (unsigned)m) * 2u) : (int)g.a_bytes, 0, 0);
            }
        } else if constexpr (A_KC) {
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int rl = p * RPP + tid / TPR, row = m0 + rl, k = kc + (tid % TPR) * 4;
                const bool ok = rl < LBM && row < g.M && k < kend;
                float q[4];
                buf_load4(ra, ok ? ((unsigned)row * (unsigned)g.a_rs + (unsigned)k) * 4u : g.a_bytes, q);
#pragma unroll
                for (int e = 0; e < 4; ++e) sa[p * 4 + e] = q[e];
            }
        } else {
            const int m = m0 + tid % LBM, k0 = kc + (tid / LBM) * KPA;
#pragma unroll
            for (int e = 0; e < KPA; ++e) {
                const bool ok = m < g.M && k0 + e < kend;
                sa[e] = buf_load(ra, ok ? ((unsigned)(k0 + e) * (unsigned)g.a_cs + (unsigned)m) * 4u : g.a_bytes);
            }
        }
        if constexpr (b16 && !B_KC) {
            const int n = n0 + 2 * (tid & 31), k0 = kc + (tid >> 5) * KB2;
#pragma unroll
            for (int e = 0; e < KB2; ++e) {
                const bool ok = n < n_real && k0 + e < kend;
                sb2[e] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rb, ok ? (int)(((unsigned)(k0 + e) * (unsigned)g.b_rs + (unsigned)n) * 2u) : (int)g.b_bytes, 0, 0);
            }
        } else if constexpr (B_KC) {
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int nl = p * RPP + tid / TPR, n = n0 + nl, k = kc + (tid % TPR) * 4;
                const bool ok = nl < LBN && n < n_real && k < kend;
                float q[4];
                buf_load4(rb, ok ? ((unsigned)n * (unsigned)g.b_cs + (unsigned)k) * 4u : g.b_bytes, q);
#pragma unroll
                for (int e = 0; e < 4; ++e) sb[p * 4 + e] = q[e];
            }
        } else {
            const int n = n0 + tid % LBN, k0 = kc + (tid / LBN) * KPB;
#pragma unroll
            for (int e = 0; e < KPB; ++e) {
                const bool ok = n < n_real && k0 + e < kend;
                sb[e] = buf_load(rb, ok ? ((unsigned)(k0 + e) * (unsigned)g.b_rs + (unsigned)n) * 4u : g.b_bytes);
            }
        }
    };
    auto lstore = [&]() {
        if constexpr (a16 && A_KC) {
#pragma unroll
            for (int p = 0; p < PA16; ++p) {
                const int rl = p * RPP16 + tid / TPR16;
                if (rl < LBM) *reinterpret_cast<uint4*>(As + rl * PK + (tid % TPR16) * 8) = sa16[p];
            }
        } else if constexpr (a16 && !A_KC) {
            unsigned short* dst = As + (2 * (tid & 63)) * PK + (tid >> 6) * KA2;
#pragma unroll
            for (int e = 0; e < KA2; e += 8) {
                uint4 lo, hi;
                lo.x = (sa2[e] & 0xffffu) | (sa2[e + 1] << 16);     hi.x = (sa2[e] >> 16) | (sa2[e + 1] & 0xffff0000u);
                lo.y = (sa2[e + 2] & 0xffffu) | (sa2[e + 3] << 16); hi.y = (sa2[e + 2] >> 16) | (sa2[e + 3] & 0xffff0000u);
                lo.z = (sa2[e + 4] & 0xffffu) | (sa2[e + 5] << 16); hi.z = (sa2[e + 4] >> 16) | (sa2[e + 5] & 0xffff0000u);
                lo.w = (sa2[e + 6] & 0xffffu) | (sa2[e + 7] << 16); hi.w = (sa2[e + 6] >> 16) | (sa2[e + 7] & 0xffff0000u);
                *reinterpret_cast<uint4*>(dst + e) = lo;
                *reinterpret_cast<uint4*>(dst + PK + e) = hi;
            }
            if (want_rowsum) {
#pragma unroll
                for (int e = 0; e < KA2; ++e) { rsum += __uint_as_float(sa2[e] << 16); rsum_hi += __uint_as_float(sa2[e] & 0xffff0000u); }
            }
        } else if constexpr (A_KC) {
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int rl = p * RPP + tid / TPR;
                uint2 w;
                w.x = pack2_bf16(sa[p * 4 + 0], sa[p * 4 + 1]);
                w.y = pack2_bf16(sa[p * 4 + 2], sa[p * 4 + 3]);
                if (rl < LBM) *reinterpret_cast<uint2*>(As + rl * PK + (tid % TPR) * 4) = w;
            }
        } else {
            unsigned short* dst = As + (tid % LBM) * PK + (tid / LBM) * KPA;
            if (want_rowsum) {
#pragma unroll
                for (int e = 0; e < KPA; ++e) rsum += sa[e];
            }
#pragma unroll
            for (int e = 0; e < KPA; e += 4) {
                uint2 w;
                w.x = pack2_bf16(sa[e], sa[e + 1]);
                w.y = pack2_bf16(sa[e + 2], sa[e + 3]);
                *reinterpret_cast<uint2*>(dst + e) = w;
            }
        }
        if constexpr (b16 && !B_KC) {
            unsigned short* dst = Bs + (2 * (tid & 31)) * PK + (tid >> 5) * KB2;
#pragma unroll
            for (int e = 0; e < KB2; e += 4) {
                uint2 lo, hi;
                lo.x = (sb2[e] & 0xffffu) | (sb2[e + 1] << 16);     hi.x = (sb2[e] >> 16) | (sb2[e + 1] & 0xffff0000u);
                lo.y = (sb2[e + 2] & 0xffffu) | (sb2[e + 3] << 16); hi.y = (sb2[e + 2] >> 16) | (sb2[e + 3] & 0xffff0000u);
                *reinterpret_cast<uint2*>(dst + e) = lo;
                *reinterpret_cast<uint2*>(dst + PK + e) = hi;
            }
        } else if constexpr (B_KC) {
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int nl = p * RPP + tid / TPR;
                uint2 w;
                w.x = pack2_bf16(sb[p * 4 + 0], sb[p * 4 + 1]);
                w.y = pack2_bf16(sb[p * 4 + 2], sb[p * 4 + 3]);
                if (nl < LBN) *reinterpret_cast<uint2*>(Bs + nl * PK + (tid % TPR) * 4) = w;
            }
        } else {
            unsigned short* dst = Bs + (tid % LBN) * PK + (tid / LBN) * KPB;
#pragma unroll
            for (int e = 0; e < KPB; e += 4) {
                uint2 w;
                w.x = pack2_bf16(sb[e], sb[e + 1]);
                w.y = pack2_bf16(sb[e + 2], sb[e + 3]);
                *reinterpret_cast<uint2*>(dst + e) = w;
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    gload(kbeg);
    for (int kc = kbeg; kc < kend; kc += BK) {
        lstore();
        __syncthreads();
        if (kc + BK < kend) gload(kc + BK);
#pragma unroll
        for (int st0 = 0; st0 < (KSPLIT ? STEPS / 2 : STEPS); ++st0) {
            const int st = KSPLIT ? ksel * (STEPS / 2) + st0 : st0;
            const int ks = st * 16 + half * 8;            // lane (r, h): k = 8h .. 8h+7 of the 16-k step
            const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(As + (wrow + l31) * PK + ks);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const bf16x8_t bfr = *reinterpret_cast<const bf16x8_t*>(Bs + (wcol + t * 32 + l31) * PK + ks);
                acc[t] = mfma_32x32x16_bf16(af, bfr, acc[t]);
            }
        }
        __syncthreads();
    }

    if (want_rowsum) {
        if (a16) {
            const int row = m0 + 2 * (tid & 63);
            if (row < g.M) atomicAdd(g.C2 + row, rsum);
            if (row + 1 < g.M) atomicAdd(g.C2 + row + 1, rsum_hi);
        } else {
            const int row = m0 + tid % LBM;
            if (row < g.M) atomicAdd(g.C2 + row, rsum);
        }
    }
    if (KSPLIT) {                      // pair reduction: the k-half-1 waves hand their tile to their partners
        __shared__ float red[KSPLIT ? 2 * 16 * 64 : 1];    // 2 waves x 16 registers x 64 lanes
        if (ksel == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave & 1) * 16 + r) * 64 + lane] = acc[0][r];
        }
        __syncthreads();
        if (ksel == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += red[((wave & 1) * 16 + r) * 64 + lane];
    }
    // bf16-stored outputs: 2-B stores per lane are store-issue bound (MI355X_MICROARCH.md: a `short` store costs ~12x a
    // dwordx4 per byte; the decoder GEMM1 wrote its two 42 MB outputs at 1.7 TB/s) - the tile goes through LDS (the
    // operand chunks are dead) and leaves as 16-B row pieces.  Needs whole 8-column groups (n_real, ld % 8 == 0).
    if constexpr (IO16 && EPI != EPI_ATOMIC) if ((c16 || (EPI == EPI_BIAS_GELU && z16)) && (n_real % 8 == 0)) {
        constexpr int CP = LBN + 8;                              // tile pitch (bf16)
        static_assert(LBM * CP <= LBM * PK + LBN * PK, "the output tile reuses the operand chunks");
        unsigned short* Cs = As;                                 // (As and Bs are adjacent __shared__ arrays: As alone fits for BK >= 64)
        static_assert(CP <= PK, "output tile rows fit the A chunk rows");
        float biasv[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int col = n0 + wcol + t * 32 + l31;
            biasv[t] = ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && col < n_real) ? g.bias[col] : 0.0f;
        }
        // value (t, r) of an output, formed from the accumulators when its tile is flushed (holding both outputs of the
        // gelu layer in registers next to the accumulators cost an occupancy step: 164 + 32 registers)
        auto value = [&](int t, int r, bool activated) {
            float v = acc[t][r] + biasv[t];
            if (EPI == EPI_BIAS_GELU && activated) v = gelu_erf(v);
            if (EPI == EPI_MUL_GELU_GRAD) {
                const int row = m0 + wrow + acc_row(r, half), col = n0 + wcol + t * 32 + l31;
                v *= (col < n_real && row < g.M) ? gelu_erf_grad(ld1e(g.G, (long)row * g.ldg + col, g16)) : 0.0f;
            }
            return v;
        };
        auto flush = [&](bool activated, float* dst, long ld, bool as16) {
            if (!as16) {                                          // this output stays fp32: plain stores
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const int col = n0 + wcol + t * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wrow + acc_row(r, half);
                        if (col < n_real && row < g.M) dst[(long)row * ld + col] = value(t, r, activated);
                    }
                }
                return;
            }
            __syncthreads();                                      // the tile buffer is free (K loop / previous flush done)
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wrow + acc_row(r, half)) * PK + wcol + t * 32 + l31] = f_to_bf16(value(t, r, activated));
            __syncthreads();
            unsigned short* out = reinterpret_cast<unsigned short*>(dst);
#pragma unroll
            for (int p = 0; p < LBM / 32; ++p) {
                const int rl = p * 32 + (tid >> 3), c8 = (tid & 7) * 8;
                const int row = m0 + rl, col = n0 + c8;
                if (row < g.M && col < n_real)
                    *reinterpret_cast<uint4*>(out + (long)row * ld + col) = *reinterpret_cast<const uint4*>(Cs + rl * PK + c8);
            }
        };
        if (EPI == EPI_BIAS_GELU) flush(false, g.Z, g.ldz, z16);
        flush(true, g.C, g.ldc, c16);
        return;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + wcol + t * 32 + l31;
        if (col >= n_real) continue;
        float bias = 0.0f;
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bias = g.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wrow + acc_row(r, half);
            if (row >= g.M) continue;
            float v = acc[t][r] + bias;
            if (EPI == EPI_BIAS_GELU) { st1e(g.Z, (long)row * g.ldz + col, v, z16); v = gelu_erf(v); }
            if (EPI == EPI_MUL_GELU_GRAD) v *= gelu_erf_grad(ld1e(g.G, (long)row * g.ldg + col, g16));
            if (EPI == EPI_ATOMIC) atomicAdd(g.C + (long)row * g.ldc + col, v);
            else st1e(g.C, (long)row * g.ldc + col, v, c16);
        }
    }
}

// out[m][k] = a[m][k] * gelu'(z[m][k]) (rows a_rs / out_rs apart, K % 4 == 0, 16-B aligned): the
// trailing-gelu prologue as its own pass when the GEMM behind it is large - inside the GEMM every
// column block would redo the erf for the whole A tile
__global__ __launch_bounds__(256) void mul_gelu_grad_kernel(const float* a, long a_rs, const float* z, float* out,
                                                            long out_rs, int M, int K) {
    const int kq = K / 4;
    const long total = (long)M * kq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / kq;
        const int k = (int)(i - m * kq) * 4;
        const float4 av = *reinterpret_cast<const float4*>(a + m * a_rs + k);
        const float4 zv = *reinterpret_cast<const float4*>(z + m * a_rs + k);
        float4 o;
        o.x = av.x * gelu_erf_grad(zv.x); o.y = av.y * gelu_erf_grad(zv.y);
        o.z = av.z * gelu_erf_grad(zv.z); o.w = av.w * gelu_erf_grad(zv.w);
        *reinterpret_cast<float4*>(out + m * out_rs + k) = o;
    }
}

// ------------------------------------------------------------------------------------
// Thin contractions of the output layer (out_dim = n2 in {1..4}: pit.py:106 `de`): with one
// dimension that small an MFMA tile is 31/32 padding and the operation is a memory-bound
// elementwise / row-dot / column-sum pass.  Used from 2^19 outputs / 8192 reduced rows (the small regime stays
// on the latency-optimised GEMM).
constexpr int THIN_MAX = 4;

// dZ1[m,n] = (sum_k dY[m,k] * W2[k,n]) * gelu'(Z1[m,n]),  K = n2 <= 4   (optionally dY *= gelu'(Z2), kept)
__global__ __launch_bounds__(256) void thin_dz1_kernel(GemmArgs g) {
    const int nq = g.N / 4;
    const long total = (long)g.M * nq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / nq;
        const int n = (int)(i - m * nq) * 4;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = 0; k < g.K; ++k) {
            float a = g.A[m * g.a_rs + k];
            if (g.a_gz) {
                a *= gelu_erf_grad(g.a_gz[m * g.a_rs + k]);
                if (g.a_out && n == 0) g.a_out[m * g.a_out_rs + k * g.a_out_cs] = a;
            }
            const float4 w = *reinterpret_cast<const float4*>(g.B + (long)k * g.b_rs + n);
            acc[0] += a * w.x; acc[1] += a * w.y; acc[2] += a * w.z; acc[3] += a * w.w;
        }
        const float4 z = ld4e(g.G, m * g.ldg + n, g.g16);
        float4 o;
        o.x = acc[0] * gelu_erf_grad(z.x); o.y = acc[1] * gelu_erf_grad(z.y);
        o.z = acc[2] * gelu_erf_grad(z.z); o.w = acc[3] * gelu_erf_grad(z.w);
        st4e(g.C, m * g.ldc + n, o, g.c16);
    }
}

// Y[m,n] = sum_k H[m,k] * W2[n,k] + b[n] (optionally gelu, Z kept),  N = n2 <= 4: tpr = K/4 (<= 64)
// lanes share a row (one 16-B load each per 4*tpr columns), 64/tpr rows per wavefront and pass
__global__ __launch_bounds__(256) void thin_fwd_kernel(GemmArgs g, int tpr, int streamed) {
    const int lane = threadIdx.x & 63;
    const int q = lane % tpr, sub = lane / tpr, rpw = 64 / tpr;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    auto finish = [&](long m, bool mv, float (&acc)[THIN_MAX]) {
        if (g.N >= 2) {
            // four partial sums per lane, tpr lanes per row: the first two exchanges HALVE the sums a lane carries (lane bit 0
            // picks the output pair, bit 1 the output of the pair), the rest add one value - 6 cross-lane moves for tpr = 32
            // instead of 20 (the reduction bounded this kernel more than its loads: NACA decoder 63 -> 50 us, 36 us with the streamed loads)
            const bool odd = q & 1, hi = (q >> 1) & 1;
            const float r0 = (odd ? acc[2] : acc[0]) + __shfl_xor(odd ? acc[0] : acc[2], 1);
            const float r1 = (odd ? acc[3] : acc[1]) + __shfl_xor(odd ? acc[1] : acc[3], 1);
            float r = (hi ? r1 : r0) + __shfl_xor(hi ? r0 : r1, 2);
            for (int o = tpr >> 1; o >= 4; o >>= 1) r += __shfl_xor(r, o);
            const int n = 2 * (q & 1) + ((q >> 1) & 1);            // the output lane q < 4 holds
            if (mv && q < 4 && n < g.N) {
                float v = r + g.bias[n];
                if (g.epi == EPI_BIAS_GELU) { g.Z[m * g.ldz + n] = v; v = gelu_erf(v); }
                g.C[m * g.ldc + n] = v;
            }
            return;
        }
        for (int o = tpr >> 1; o > 0; o >>= 1) acc[0] += __shfl_xor(acc[0], o);
        if (mv && q == 0) {
            float v = acc[0] + g.bias[0];
            if (g.epi == EPI_BIAS_GELU) { g.Z[m * g.ldz] = v; v = gelu_erf(v); }
            g.C[m * g.ldc] = v;
        }
    };
    if (streamed) {
        // the whole fp32 row in one pass (K <= 256): the lane's weight quads live in registers for the launch and EIGHT row groups
        // are requested per wavefront before the first is used - raw buffer loads, predicated on the OFFSET (a predicated global
        // load is sunk into an exec-masked block with its own s_waitcnt: one load in flight per lane, 1.9 TB/s on the NACA
        // decoder's 225 k x 128 activations)
        constexpr int U = 8;
        const int k = q * 4;
        const bool kv = k < g.K;
        const unsigned a_bytes = (unsigned)((((long)g.M - 1) * g.a_rs + g.K) * 4);
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, a_bytes);
        float4 w[THIN_MAX];
#pragma unroll
        for (int n = 0; n < THIN_MAX; ++n)
            w[n] = (kv && n < g.N) ? *reinterpret_cast<const float4*>(g.B + (long)n * g.b_cs + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        const long stride = nwaves * rpw;
        for (long mbase = wave0 * rpw; mbase < g.M; mbase += U * stride) {
            float h[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long m = mbase + u * stride + sub;
                buf_load4(ra, (m < g.M && kv) ? (unsigned)((m * g.a_rs + k) * 4) : a_bytes, h[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long m = mbase + u * stride + sub;
                if (mbase + u * stride >= g.M) break;             // wave-uniform
                float acc[THIN_MAX];
#pragma unroll
                for (int n = 0; n < THIN_MAX; ++n) acc[n] = (h[u][0] * w[n].x + h[u][1] * w[n].y) + (h[u][2] * w[n].z + h[u][3] * w[n].w);
                finish(m, m < g.M, acc);
            }
        }
        return;
    }
    for (long mbase = wave0 * rpw; mbase < g.M; mbase += nwaves * rpw) {
        const long m = mbase + sub;
        const bool mv = m < g.M;
        float acc[THIN_MAX] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = q * 4; k < g.K; k += tpr * 4) {
            const float4 h = mv ? ld4e(g.A, m * g.a_rs + k, g.a16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < THIN_MAX; ++n) {
                if (n >= g.N) break;
                const float4 w = *reinterpret_cast<const float4*>(g.B + (long)n * g.b_cs + k);
                acc[n] += (h.x * w.x + h.y * w.y) + (h.z * w.z + h.w * w.w);
            }
        }
        finish(m, mv, acc);
    }
}

// dW[j,n] += sum_m dZ[m,j] * X[m,n],  db[j] += sum_m dZ[m,j],  j < n2 <= 4: a workgroup owns a slab of
// rows, thread t the columns [4t, 4t+4) (and its copies for wider rows); fp32 atomics at the end
__global__ __launch_bounds__(256) void thin_dw_kernel(GemmArgs g, int slab) {
    __shared__ float s_red[256 * 4 * THIN_MAX];                   // [group][j][column] partial sums
    __shared__ float s_bias[4][THIN_MAX];
    const int n_real = g.N - 1;                                   // last "column" is the virtual ones column (bias)
    const long k0 = (long)blockIdx.x * slab, k1 = min((long)g.K, k0 + slab);
    // tpr threads cover one row (4 columns each), the 256/tpr thread groups interleave the slab's rows;
    // partial sums meet in LDS so that the workgroup issues ONE atomic per output element
    int tpr = 1;
    while (tpr < 256 && tpr * 4 < n_real) tpr <<= 1;
    const int groups = 256 / tpr, q = threadIdx.x % tpr, grp = threadIdx.x / tpr;
    const int ncov = tpr * 4;                                     // columns covered per pass
    for (int nb = 0; nb < n_real; nb += ncov) {
        const int n = nb + q * 4;
        float acc[THIN_MAX][4];
#pragma unroll
        for (int j = 0; j < THIN_MAX; ++j) { acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.0f; }
        if (n < n_real) {
            long k = k0 + grp;
            for (; k + 7L * groups < k1; k += 8L * groups) {      // eight rows (and their dZ values) in flight per thread
                float4 x[8];
                float d[8][THIN_MAX];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    x[r] = ld4e(g.B, (k + (long)r * groups) * g.b_rs + n, g.b16);
#pragma unroll
                    for (int j = 0; j < THIN_MAX; ++j) d[r][j] = (j < g.M) ? g.A[(k + (long)r * groups) * g.a_cs + j] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < THIN_MAX; ++j) {
                        acc[j][0] += d[r][j] * x[r].x; acc[j][1] += d[r][j] * x[r].y;
                        acc[j][2] += d[r][j] * x[r].z; acc[j][3] += d[r][j] * x[r].w;
                    }
            }
            for (; k < k1; k += groups) {
                const float4 x = ld4e(g.B, k * g.b_rs + n, g.b16);
#pragma unroll
                for (int j = 0; j < THIN_MAX; ++j) {
                    if (j >= g.M) break;
                    const float d = g.A[k * g.a_cs + j];
                    acc[j][0] += d * x.x; acc[j][1] += d * x.y; acc[j][2] += d * x.z; acc[j][3] += d * x.w;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < THIN_MAX; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s_red[(grp * THIN_MAX + j) * ncov + q * 4 + e] = acc[j][e];
        __syncthreads();
        for (int idx = threadIdx.x; idx < g.M * ncov; idx += 256) {
            const int j = idx / ncov, col = idx - j * ncov;
            if (nb + col >= n_real) continue;
            float v = 0.0f;
            for (int gq = 0; gq < groups; ++gq) v += s_red[(gq * THIN_MAX + j) * ncov + col];
            atomicAdd(g.C + (long)j * g.ldc + nb + col, v);
        }
    }
    if (g.C2) {                                                   // bias gradient: column sums of dZ over the slab
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int j = 0; j < g.M; ++j) {
            float sum = 0.0f;
            for (long k = k0 + threadIdx.x; k < k1; k += 256) sum += g.A[k * g.a_cs + j];
            sum = wave_sum(sum);
            if (lane == 0) s_bias[wave][j] = sum;
        }
        __syncthreads();
        if (threadIdx.x < g.M)
            atomicAdd(g.C2 + threadIdx.x, s_bias[0][threadIdx.x] + s_bias[1][threadIdx.x] + s_bias[2][threadIdx.x] + s_bias[3][threadIdx.x]);
    }
}

bool aligned16(const void* p);

// true if one of the thin kernels took the contraction
bool try_launch_thin(const GemmArgs& g, hipStream_t s) {
    static const bool off = exp_env("PIT_NO_THIN_GEMM") != nullptr;
    if (off) return false;
    static const long min_out = exp_env("PIT_THIN_MIN") ? atol(exp_env("PIT_THIN_MIN")) : (1L << 19);
    static const long min_rows = exp_env("PIT_THIN_MIN_ROWS") ? atol(exp_env("PIT_THIN_MIN_ROWS")) : 8192;
    // bf16 storage flags: thin_dz1 reads G / writes C in either format, thin_fwd reads A, thin_dw reads B; anything else
    // keeps the contraction off these kernels
    if (g.epi == EPI_MUL_GELU_GRAD && g.K <= THIN_MAX && g.a_cs == 1 && g.b_cs == 1 && g.N % 4 == 0 && g.b_rs % 4 == 0 &&
        g.ldg % 4 == 0 && g.ldc % 4 == 0 && aligned16(g.B) && aligned16(g.G) && aligned16(g.C) && !g.a16 && !g.b16 && !g.z16 &&
        (!(g.g16 || g.c16) || !g.a_gz) && (long)g.M * g.N >= min_out) {
        const long quads = (long)g.M * (g.N / 4);
        hipLaunchKernelGGL(thin_dz1_kernel, dim3((unsigned)std::min<long>((quads + 255) / 256, 8192)), dim3(256), 0, s, g);
        return true;
    }
    if ((g.epi == EPI_BIAS || g.epi == EPI_BIAS_GELU) && g.N <= THIN_MAX && g.a_cs == 1 && g.b_rs == 1 && g.K % 4 == 0 &&
        g.a_rs % 4 == 0 && g.b_cs % 4 == 0 && aligned16(g.A) && aligned16(g.B) && !g.a_gz && !g.b16 && !g.c16 && !g.z16 && !g.g16 &&
        (long)g.M * g.K >= min_out) {
        int tpr = 4;                                              // lanes per row: >= n2 (one output column each), <= 64
        while (tpr < 64 && tpr * 4 < g.K) tpr <<= 1;
        const long rows_per_wg = 4L * (64 / tpr);
        // single-pass fp32 rows of many-row launches: 2048 workgroups walk the rows with eight row groups in flight each
        const unsigned long long a_bytes = (unsigned long long)(((long)g.M - 1) * g.a_rs + g.K) * 4ull;
        static const bool no_stream = exp_env("PIT_NO_THIN_STREAM") != nullptr;      // (diagnostic switch, read once)
        const int streamed = (g.K <= tpr * 4 && !g.a16 && a_bytes < PIT_MAX_BUFFER_BYTES && g.M >= 65536 && !no_stream) ? 1 : 0;
        hipLaunchKernelGGL(thin_fwd_kernel, dim3((unsigned)std::min<long>((g.M + rows_per_wg - 1) / rows_per_wg, streamed ? 2048 : 16384)),
                           dim3(256), 0, s, g, tpr, streamed);
        return true;
    }
    if (g.epi == EPI_ATOMIC && g.M <= THIN_MAX && g.ones_col == g.N - 1 && g.a_rs == 1 && g.b_cs == 1 && (g.N - 1) % 4 == 0 &&
        g.b_rs % 4 == 0 && aligned16(g.B) && !g.a_gz && !g.a16 && !g.c16 && !g.z16 && !g.g16 && g.K >= min_rows) {
        // one workgroup per CU: every workgroup ends with one atomic per output element, and atomics on
        // one address serialise in L2 (~40 ns each) - 256 of them cost less than the pass over the rows
        const int slab = std::max(256, ((g.K + 255) / 256 + 15) / 16 * 16);
        hipLaunchKernelGGL(thin_dw_kernel, dim3((unsigned)((g.K + slab - 1) / slab)), dim3(256), 0, s, g, slab);
        return true;
    }
    return false;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 0 = not eligible; otherwise launches the LDS-staged kernel (called after prepare_gemm filled the
// buffer extents).  Eligibility: layouts that allow 16-B coalesced staging, and enough work that
// the tile pipeline pays (small layers stay on the latency-optimised register-direct kernel).
// k per LDS chunk of gemm_bfl_kernel by tile height (see the kernel): 128-row tiles 64 (staging registers), smaller ones 128
constexpr int BK128 = 64, BK64 = 128;        // bf16-stored operands (IO16)
#ifndef PIT_BKF128
#define PIT_BKF128 32
#define PIT_BKF64 64
#endif
constexpr int BKF128 = PIT_BKF128, BKF64 = PIT_BKF64;     // fp32-stored operands: converted in registers on the way to LDS
// the fp32 weight-gradient reductions of the large regime (gemm_rr_kernel): one (g2 == nullptr) or the two reductions of
// one MLP, which share K (the batch rows), in ONE launch; preconditions checked by the caller (gemm_rr_ok)
bool gemm_rr_ok(const GemmArgs& g) {
    const int n_real = (g.ones_col >= 0) ? g.N - 1 : g.N;
    static const bool no_bf = exp_env("PIT_NO_GEMM_RR_BF16") != nullptr;
    if (g.bf16 && (no_bf || exp_env("PIT_BF16_LEGACY"))) return false;
    return g.epi == EPI_ATOMIC && !g.a_gz && !(g.a16 || g.b16 || g.c16 || g.z16 || g.g16) &&
           g.a_rs == 1 && g.a_cs % 4 == 0 && g.M % 4 == 0 && g.b_cs == 1 && g.b_rs % 4 == 0 && n_real % 4 == 0 &&
           aligned16(g.A) && aligned16(g.B);
}
void launch_gemm_rr(const GemmArgs& g1, const GemmArgs* g2, hipStream_t s) {
    int rm = 1, rn = 1, bk = 64, per_cu = 0, pad_kb = 0;
    static const char* cfg = exp_env("PIT_RR_CFG");                 // experiments: "rm,rn,bk,workgroups per CU,LDS padding KB"
    if (cfg) sscanf(cfg, "%d,%d,%d,%d,%d", &rm, &rn, &bk, &per_cu, &pad_kb);
    const int bm = 64 * rm, bn = 64 * rn;
    auto tiles_of = [&](const GemmArgs& g, int& tx) {
        const int n_real = (g.ones_col >= 0) ? g.N - 1 : g.N;
        tx = (n_real + bn - 1) / bn;
        return tx * ((g.M + bm - 1) / bm);
    };
    int tx1 = 1, tx2 = 1;
    const int T1 = tiles_of(g1, tx1), T2 = g2 ? tiles_of(*g2, tx2) : 0;
    const int nchunks = (g1.K + bk - 1) / bk;
    // As many K slabs as keep every CU at exactly 1 (or 2) workgroups.  One workgroup more than the chip holds at once and
    // the launch takes two rounds; an XCD with 33 of them on its 32 CUs runs 1.6x as long (tools/stamp_dw.py: that, not the
    // inner loop, was what held gemm_lds_kernel<128, F, F, ATOMIC> at 25-45 % MFMA busy).  Two per CU when a workgroup then
    // still contracts >= 8 chunks: the second wave per SIMD hides the barriers, but prologue and epilogue are per workgroup.
    // Measured (us per MLP, both reductions; round-2 kernels -> one per CU / two per CU): Darcy b=256 62.7 -> 31.5 / 31.7,
    // Vorticity 51.8 -> 36.3 / 34.2, NACA 39.2 -> 22.4 / 22.4, Elasticity 26.7 -> 13.8 / 15.1, Cylinder b=200 307 -> 281 / 244
    if (per_cu <= 0) per_cu = (nchunks / std::max(1, 512 / (T1 + T2)) >= 8) ? 2 : 1;
    const int slabs = std::max(1, std::min(std::max(1, nchunks / 2), 256 * per_cu / (T1 + T2)));
    const int P = (T1 + T2) * slabs;
    const dim3 grid((unsigned)(8 * ((P + 7) / 8))), block(256);
    const GemmArgs& gb = g2 ? *g2 : g1;
#define PIT_RR(RM_, RN_, BK_, BF_) hipLaunchKernelGGL((gemm_rr_kernel<RM_, RN_, BK_, BF_>), grid, block, (size_t)pad_kb * 1024, s, g1, gb, tx1, T1, tx2, T2, slabs, nchunks)
    if (g1.bf16) PIT_RR(1, 1, 64, true);
    else if (rm == 1 && rn == 2) PIT_RR(1, 2, 32, false);          // (experiments only: 64 x 128 tiles)
    else PIT_RR(1, 1, 64, false);
#undef PIT_RR
}

bool try_launch_gemm_lds(GemmArgs g, hipStream_t s) {
    static const int mode = exp_env("PIT_LDS_GEMM") ? atoi(exp_env("PIT_LDS_GEMM")) : 1;   // 0 off, 1 auto, 2 always when legal
    if (mode == 0) return false;
    const int n_real = (g.ones_col >= 0) ? g.N - 1 : g.N;
    const bool a_kc = g.a_cs == 1 && g.a_rs % 4 == 0 && g.K % 4 == 0;
    const bool a_ic = g.a_rs == 1 && g.a_cs % 4 == 0 && g.M % 4 == 0;
    const bool b_kc = g.b_rs == 1 && g.b_cs % 4 == 0 && g.K % 4 == 0;
    const bool b_ic = g.b_cs == 1 && g.b_rs % 4 == 0 && n_real % 4 == 0;
    if (!aligned16(g.A) || !aligned16(g.B) || (g.a_gz && !aligned16(g.a_gz))) return false;
    const bool io16 = g.a16 || g.b16 || g.c16 || g.z16 || g.g16;
    if (io16) {                                      // bf16 storage: gemm_bfl_kernel only, and only the layouts it stages
        if (!g.bf16 || exp_env("PIT_BF16_LEGACY") || g.a_gz) return false;
        if (g.a16 && !((a_kc && g.a_rs % 8 == 0 && g.K % 8 == 0) || (g.epi == EPI_ATOMIC && a_ic && g.a_cs % 2 == 0 && g.M % 2 == 0))) return false;
        if (g.b16 && !(g.epi == EPI_ATOMIC && b_ic && g.b_rs % 2 == 0 && n_real % 2 == 0)) return false;
        if ((g.c16 || g.z16 || g.g16) && g.epi == EPI_ATOMIC) return false;
        // the storage-aware instantiation reads a bf16 A operand (and, row-reducing kind, a bf16 B): both are what the
        // decoder tail hands it (pit_mlp_*: x, Z1/H and dZ1 are bf16 together)
        if (!g.a16 || (g.epi == EPI_ATOMIC) != (g.b16 != 0)) return false;
    }
    const long work = (long)g.M * g.N * g.K;
    if (mode == 1 && !io16 && (work < (1L << 27) || g.N < 48)) return false;
    int kind = -1;                                   // which instantiation
    if (g.epi == EPI_ATOMIC) { if (a_ic && b_ic && !g.a_gz) kind = 4; }
    else if (a_kc && b_kc && !g.a_gz && (g.epi == EPI_BIAS || g.epi == EPI_BIAS_GELU)) kind = (g.epi == EPI_BIAS) ? 0 : 1;
    else if (a_kc && b_ic && (g.epi == EPI_MUL_GELU_GRAD || g.epi == EPI_STORE)) kind = (g.epi == EPI_MUL_GELU_GRAD) ? 2 : 3;
    if (kind < 0) return false;
    static const bool no_agz = exp_env("PIT_NO_FUSED_GELU_PROLOGUE") != nullptr;
    bool agz = false;
    if (g.a_gz) {                                    // trailing-gelu prologue
        if (!g.a_out || g.a_out_cs != 1 || g.a_out_rs % 4 != 0 || !aligned16(g.a_out)) return false;
        if (kind == 2 && !g.bf16 && !io16 && !no_agz && g.K <= 128) {    // (K = 256: neutral at Vorticity, +0.6 % at Cylinder)
            agz = true;                              // fp32 mode: inside the GEMM's staging (gemm_lds_kernel<..., AGZ>)
        } else {                                     // as its own elementwise pass
            const long quads = (long)g.M * (g.K / 4);
            hipLaunchKernelGGL(mul_gelu_grad_kernel, dim3((unsigned)std::min<long>((quads + 255) / 256, 4096)), dim3(256), 0, s,
                               g.A, g.a_rs, g.a_gz, g.a_out, g.a_out_rs, g.M, g.K);
            g.A = g.a_out; g.a_rs = g.a_out_rs;
            g.a_bytes = (unsigned)((((unsigned long long)(g.M - 1) * g.a_rs) + g.K) * 4ull);
            g.a_gz = nullptr; g.a_out = nullptr;
        }
    }
    const int gx = (n_real + LBN - 1) / LBN;
    // 128-row tiles while they still give two workgroups per CU (or when K slabs add parallelism)
    // (and 32-row tiles with an in-workgroup K split while even the 64-row tiling leaves CUs idle)
    int bm = ((long)gx * ((g.M + 127) / 128) >= 512 || g.epi == EPI_ATOMIC) ? 128 : 64;
    // short contractions (K <= 256: a workgroup is a load phase, a few chunks and a store phase, and all resident
    // workgroups go through them in step - tools/stamp_gemm.py: 512 workgroups, every one alive for the whole launch, 4 us of
    // MFMA in a 15 us life): 64-row tiles double the workgroups and let one's stores overlap another's loads - the hid-64 MLP
    // at 65 536 rows: forward 49.1 -> 42.5 us, backward data path 59.8 -> 51.9 us, Darcy b=256 step 2.04 -> 1.94 ms
    // (N <= 256 only: the decoder's K = 256, N = 768 GEMMs are compute-bound at 70 % MFMA busy and keep the tall tile -
    // Vorticity bf16 1.249 -> 1.221 ms with the limit)
    static const int short_n = exp_env("PIT_LDS_SHORT_N") ? atoi(exp_env("PIT_LDS_SHORT_N")) : 256;
    if (bm == 128 && g.epi != EPI_ATOMIC && g.K <= 256 && g.N <= short_n) bm = 64;
    static const int force_bm = exp_env("PIT_LDS_BM") ? atoi(exp_env("PIT_LDS_BM")) : 0;      // experiments
    if (force_bm && g.epi != EPI_ATOMIC) bm = force_bm;
    static const bool no32 = exp_env("PIT_LDS_NO_BM32") != nullptr;
    if (!force_bm && bm == 64 && (long)gx * ((g.M + 63) / 64) < 512 && !no32) bm = 32;
    const int gy = (g.M + bm - 1) / bm;
    int splits = 1;
    g.k_slab = ((g.K + LBK - 1) / LBK) * LBK;
    if (g.epi == EPI_ATOMIC) {                        // K slabs: >= 256 deep, enough workgroups to fill the chip twice
        splits = std::max(1, std::min(g.K / 256, (512 + gx * gy - 1) / (gx * gy)));
        int slab = (g.K + splits - 1) / splits;
        slab = ((slab + LBK - 1) / LBK) * LBK;
        splits = (g.K + slab - 1) / slab;
        g.k_slab = slab;
    }
    dim3 grid(gx, gy, splits), block(256);
    static const bool no_remap = exp_env("PIT_NO_GEMM_XCD_REMAP") != nullptr;
    if (!no_remap && g.epi != EPI_ATOMIC && gx >= 2 && gy >= 64 && 8L * gx * ((gy + 7) / 8) < 0x7fffffffL) {
        g.remap_gx = gx; g.remap_gy = gy;
        grid = dim3((unsigned)(8 * gx * ((gy + 7) / 8)), 1, 1);
    }
#define PIT_LDS_BF(A_, B_, EPI_, BF_)                                                                         \
    do {                                                                                                       \
        if (bm == 128) hipLaunchKernelGGL((gemm_lds_kernel<128, A_, B_, EPI_, BF_>), grid, block, 0, s, g);    \
        else if (bm == 64) hipLaunchKernelGGL((gemm_lds_kernel<64, A_, B_, EPI_, BF_>), grid, block, 0, s, g);  \
        else hipLaunchKernelGGL((gemm_lds_kernel<32, A_, B_, EPI_, BF_>), grid, block, 0, s, g);               \
    } while (0)
#define PIT_BFL(A_, B_, EPI_)                                                                                  \
    do {                                                                                                       \
        if (io16) {                                                                                            \
            if (bm == 128) hipLaunchKernelGGL((gemm_bfl_kernel<128, A_, B_, EPI_, true, BK128>), grid, block, 0, s, g);   \
            else if (bm == 64) hipLaunchKernelGGL((gemm_bfl_kernel<64, A_, B_, EPI_, true, BK64>), grid, block, 0, s, g); \
            else hipLaunchKernelGGL((gemm_bfl_kernel<32, A_, B_, EPI_, true, BK64>), grid, block, 0, s, g);    \
        }                                                                                                      \
        else if (bm == 128) hipLaunchKernelGGL((gemm_bfl_kernel<128, A_, B_, EPI_, false, BKF128>), grid, block, 0, s, g); \
        else if (bm == 64) hipLaunchKernelGGL((gemm_bfl_kernel<64, A_, B_, EPI_, false, BKF64>), grid, block, 0, s, g);    \
        else hipLaunchKernelGGL((gemm_bfl_kernel<32, A_, B_, EPI_, false, BKF64>), grid, block, 0, s, g);      \
    } while (0)
    static const bool legacy_bf = exp_env("PIT_BF16_LEGACY") != nullptr;      // the round-1 form: fp32 in LDS, 32x32x8 MFMA
    static const bool no_rr = exp_env("PIT_NO_GEMM_RR") != nullptr;           // the round-2 kernel for the fp32 weight gradients
#ifdef PIT_EXPERIMENTS
#define PIT_LDS(A_, B_, EPI_) do { if (g.bf16 && !legacy_bf) PIT_BFL(A_, B_, EPI_); else if (g.bf16) PIT_LDS_BF(A_, B_, EPI_, true); else PIT_LDS_BF(A_, B_, EPI_, false); } while (0)
#else
#define PIT_LDS(A_, B_, EPI_) do { (void)legacy_bf; if (g.bf16) PIT_BFL(A_, B_, EPI_); else PIT_LDS_BF(A_, B_, EPI_, false); } while (0)
#endif
    switch (kind) {
        case 0: PIT_LDS(true, true, EPI_BIAS); break;
        case 1: PIT_LDS(true, true, EPI_BIAS_GELU); break;
        case 2:
            if (agz) {
                if (bm == 128) hipLaunchKernelGGL((gemm_lds_kernel<128, true, false, EPI_MUL_GELU_GRAD, false, true>), grid, block, 0, s, g);
                else if (bm == 64) hipLaunchKernelGGL((gemm_lds_kernel<64, true, false, EPI_MUL_GELU_GRAD, false, true>), grid, block, 0, s, g);
                else hipLaunchKernelGGL((gemm_lds_kernel<32, true, false, EPI_MUL_GELU_GRAD, false, true>), grid, block, 0, s, g);
            } else PIT_LDS(true, false, EPI_MUL_GELU_GRAD);
            break;
        case 3: PIT_LDS(true, false, EPI_STORE); break;
        default:
            if (g.bf16 && !legacy_bf && io16) hipLaunchKernelGGL((gemm_bfl_kernel<128, false, false, EPI_ATOMIC, true, BK128>), grid, block, 0, s, g);
            else if (!no_rr && gemm_rr_ok(g)) launch_gemm_rr(g, nullptr, s);
            else if (g.bf16 && !legacy_bf) hipLaunchKernelGGL((gemm_bfl_kernel<128, false, false, EPI_ATOMIC, false, BKF128>), grid, block, 0, s, g);
#ifdef PIT_EXPERIMENTS
            else if (g.bf16) hipLaunchKernelGGL((gemm_lds_kernel<128, false, false, EPI_ATOMIC, true>), grid, block, 0, s, g);
#endif
            else hipLaunchKernelGGL((gemm_lds_kernel<128, false, false, EPI_ATOMIC, false>), grid, block, 0, s, g);
            break;
    }
#undef PIT_LDS
#undef PIT_BFL
#undef PIT_LDS_BF
    return true;
}

int pow2_floor_i(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }

bool vec_ok(const float* p, long i_stride, long k_stride) {
    return k_stride == 1 && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (i_stride % 4) == 0;
}

struct GemmLaunch { int tn, nwaves; dim3 grid; };

int prepare_gemm(GemmArgs& g, GemmLaunch& L, int force_tn = 0, int force_waves = 0, int target_wgs = 768) {
    const unsigned long long ab = ((unsigned long long)(g.M - 1) * g.a_rs + (unsigned long long)(g.K - 1) * g.a_cs + 1) * (g.a16 ? 2ull : 4ull);
    const int nb_cols = (g.ones_col >= 0) ? g.N - 1 : g.N;       // the ones column is virtual
    const unsigned long long bb = ((unsigned long long)(g.K - 1) * g.b_rs + (unsigned long long)(std::max(nb_cols, 1) - 1) * g.b_cs + 1) * (g.b16 ? 2ull : 4ull);
    if (ab > PIT_MAX_BUFFER_BYTES || bb > PIT_MAX_BUFFER_BYTES) return PIT_ERR_UNSUPPORTED;
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
    g.bf16 = (t_call_math == PIT_MATH_BF16);
    static const int seq_epi = exp_env("PIT_GEMM_RD_SEQ_EPI") != nullptr;
    g.seq_epi = seq_epi;
    g.a_vec = vec_ok(g.A, g.a_rs, g.a_cs) && (!g.a_gz || vec_ok(g.a_gz, g.a_rs, g.a_cs));
    g.b_vec = vec_ok(g.B, g.b_cs, g.b_rs);
    // two column tiles per wave (A fragment reused) only when that still leaves plenty of workgroups
    int tn = (g.N > 32 && (long)((g.M + 31) / 32) * ((g.N + 63) / 64) >= 512) ? 2 : 1;
    if (force_tn) tn = force_tn;
    const int tiles = ((g.M + 31) / 32) * ((g.N + 32 * tn - 1) / (32 * tn));
    // waves per workgroup: each wave wants >= 16 k; slabs over blockIdx.z only for the atomic
    // (row-reducing) GEMMs, sized so the launch has a few hundred workgroups
    int splits = 1;
    if (g.atomic) {
        // (fewer waves per workgroup than the usual 8 - a carrying launch with 256-thread workgroups: proportionally
        // shorter slabs, so that a wave's share of the rows, the critical path, stays what it is with 8)
        const int wscale = (force_waves > 0 && force_waves < 8) ? 8 / force_waves : 1;
        splits = std::max(1, std::min((g.K + 127) / 128, (target_wgs * wscale + tiles - 1) / tiles));
        // every split costs one fp32 atomic per output element: long reductions keep slabs >= 512
        if (g.K > 4096) splits = std::max(1, std::min(splits, g.K * wscale / 512));
    }
    int slab = (g.K + splits - 1) / splits;
    slab = ((slab + 7) / 8) * 8;
    splits = (g.K + slab - 1) / slab;
    g.k_slab = slab;
    // >= 2 waves always: the reduce-scatter epilogue indexes the parked tile in LDS
    L.nwaves = force_waves ? force_waves : std::max(2, std::min(8, pow2_floor_i(slab / 16)));
    L.tn = tn;
    L.grid = dim3((g.N + 32 * tn - 1) / (32 * tn), (g.M + 31) / 32, splits);
    return 0;
}

bool try_launch_gemm_lds(GemmArgs g, hipStream_t s);

int launch_gemm(GemmArgs g, hipStream_t s) {
    GemmLaunch L;
    if (int rc = prepare_gemm(g, L)) return rc;
    if (try_launch_thin(g, s)) return 0;
    if (try_launch_gemm_lds(g, s)) return 0;
    if (g.a16 || g.b16 || g.c16 || g.z16 || g.g16) return PIT_ERR_UNSUPPORTED;     // the register-direct kernel is fp32-storage only
    const int tn = L.tn;
    dim3 grid = L.grid, block(64 * L.nwaves);
    const size_t sm = (size_t)L.nwaves * tn * 16 * 64 * sizeof(float);
#define PIT_GEMM(TN_, EPI_) hipLaunchKernelGGL((gemm_rd_kernel<TN_, EPI_>), grid, block, sm, s, g)
#define PIT_GEMM_TN(EPI_) do { if (tn == 2) PIT_GEMM(2, EPI_); else PIT_GEMM(1, EPI_); } while (0)
    switch (g.epi) {
        case EPI_STORE: PIT_GEMM_TN(EPI_STORE); break;
        case EPI_BIAS: PIT_GEMM_TN(EPI_BIAS); break;
        case EPI_BIAS_GELU: PIT_GEMM_TN(EPI_BIAS_GELU); break;
        case EPI_MUL_GELU_GRAD: PIT_GEMM_TN(EPI_MUL_GELU_GRAD); break;
        default: PIT_GEMM_TN(EPI_ATOMIC); break;
    }
#undef PIT_GEMM_TN
#undef PIT_GEMM
    return 0;
}

// both weight-gradient reductions of an MLP in one launch (EPI_ATOMIC) when they are small;
// big reductions keep their own launches (and their own tile shapes)
int launch_gemm_pair_atomic(GemmArgs g1, GemmArgs g2, hipStream_t s) {
    const bool io16 = g1.a16 || g1.b16 || g2.a16 || g2.b16;
    if (io16 || (long)g1.M * g1.N * g1.K + (long)g2.M * g2.N * g2.K > (1L << 28)) {
        static const bool no_rr = exp_env("PIT_NO_GEMM_RR") != nullptr || exp_env("PIT_NO_GEMM_RR_PAIR") != nullptr;
        if (!no_rr && !io16 && g1.K == g2.K && std::min(g1.N, g2.N) >= 48 && std::min(g1.M, g2.M) >= 32) {
            GemmLaunch L;
            if (int rc = prepare_gemm(g1, L)) return rc;           // (operand extents, math mode)
            if (int rc = prepare_gemm(g2, L)) return rc;
            if (gemm_rr_ok(g1) && gemm_rr_ok(g2)) {
                launch_gemm_rr(g1, &g2, s);           // both reductions share the batch rows: one balanced launch
                return 0;
            }
        }
        if (int rc = launch_gemm(g1, s)) return rc;
        return launch_gemm(g2, s);
    }
    GemmLaunch L1, L2;
    if (int rc = prepare_gemm(g1, L1, 1, 8)) return rc;
    if (int rc = prepare_gemm(g2, L2, 1, 8)) return rc;
    const int n1 = L1.grid.x * L1.grid.y * L1.grid.z, n2 = L2.grid.x * L2.grid.y * L2.grid.z;
    const size_t sm = (size_t)8 * 16 * 64 * sizeof(float);
    hipLaunchKernelGGL((gemm_rd_pair_kernel<1, EPI_ATOMIC>), dim3(n1 + n2), dim3(512), sm, s, g1, g2, n1,
                       (int)L1.grid.x, (int)L1.grid.y, (int)L2.grid.x, (int)L2.grid.y);
    return 0;
}

// dX GEMM + both weight-gradient reductions in one launch when everything is small; otherwise the
// dX GEMM alone followed by launch_gemm_pair_atomic
int launch_gemm_bwd_tail(GemmArgs gx, GemmArgs g1, GemmArgs g2, hipStream_t s) {
    static const bool off = exp_env("PIT_NO_BWD_PAIR") != nullptr;
    const long work = (long)gx.M * gx.N * gx.K + (long)g1.M * g1.N * g1.K + (long)g2.M * g2.N * g2.K;
    // the dX GEMM runs with 8 k-splitting waves here: only worth it while it is itself tiny
    const bool io16 = gx.a16 || gx.c16 || g1.a16 || g1.b16 || g2.a16 || g2.b16;
    if (off || io16 || work > (1L << 28) || (long)gx.M * gx.N * gx.K > (1L << 27)) {
        if (int rc = launch_gemm(gx, s)) return rc;
        return launch_gemm_pair_atomic(g1, g2, s);
    }
    GemmLaunch Lx, L1, L2;
    if (int rc = prepare_gemm(gx, Lx, 0, 8)) return rc;
    if (int rc = prepare_gemm(g1, L1, 1, 8)) return rc;
    if (int rc = prepare_gemm(g2, L2, 1, 8)) return rc;
    const int nx = Lx.grid.x * Lx.grid.y;                 // not split over z (non-atomic)
    const int n1 = L1.grid.x * L1.grid.y * L1.grid.z, n2 = L2.grid.x * L2.grid.y * L2.grid.z;
    const size_t sm = (size_t)8 * Lx.tn * 16 * 64 * sizeof(float);
    if (Lx.tn == 2)
        hipLaunchKernelGGL((gemm_rd_triple_kernel<2>), dim3(nx + n1 + n2), dim3(512), sm, s, gx, g1, g2, nx, (int)Lx.grid.x,
                           n1, (int)L1.grid.x, (int)L1.grid.y, (int)L2.grid.x, (int)L2.grid.y);
    else
        hipLaunchKernelGGL((gemm_rd_triple_kernel<1>), dim3(nx + n1 + n2), dim3(512), sm, s, gx, g1, g2, nx, (int)Lx.grid.x,
                           n1, (int)L1.grid.x, (int)L1.grid.y, (int)L2.grid.x, (int)L2.grid.y);
    return 0;
}

GemmArgs blank() {
    GemmArgs g{};
    g.ones_col = -1;
    return g;
}

// dX = dZ1 W1
GemmArgs make_dx(const float* dz1, const float* w1, int rows, int n0, int n1, float* d_x, long ld_dx) {
    GemmArgs g = blank();
    g.A = dz1; g.a_rs = n1; g.a_cs = 1;
    g.B = w1; g.b_rs = n0; g.b_cs = 1;
    g.M = rows; g.N = n0; g.K = n1;
    g.C = d_x; g.ldc = ld_dx; g.epi = EPI_STORE;
    return g;
}
// dW2 += dZ2^T H (+ db2 as the ones column)
GemmArgs make_dw2(const float* dz2, long ld_dz2, const float* h, int rows, int n1, int n2, float* d_w2, float* d_b2) {
    GemmArgs g = blank();
    g.A = dz2; g.a_rs = 1; g.a_cs = ld_dz2;        // A(m,k) = dz2[k][m]
    g.B = h; g.b_rs = n1; g.b_cs = 1;
    g.M = n2; g.N = n1 + 1; g.K = rows; g.ones_col = n1;
    g.C = d_w2; g.ldc = n1; g.C2 = d_b2; g.atomic = 1; g.epi = EPI_ATOMIC;
    return g;
}
// dW1 += dZ1^T X (+ db1)
GemmArgs make_dw1(const float* dz1, const float* x, long ldx, int rows, int n0, int n1, float* d_w1, float* d_b1) {
    GemmArgs g = blank();
    g.A = dz1; g.a_rs = 1; g.a_cs = n1;
    g.B = x; g.b_rs = ldx; g.b_cs = 1;
    g.M = n1; g.N = n0 + 1; g.K = rows; g.ones_col = n0;
    g.C = d_w1; g.ldc = n0; g.C2 = d_b1; g.atomic = 1; g.epi = EPI_ATOMIC;
    return g;
}

int zero_param_grads(float* d_w1, float* d_b1, float* d_w2, float* d_b2, int n0, int n1, int n2, hipStream_t s) {
    hipError_t e;
    if ((e = hipMemsetAsync(d_w1, 0, sizeof(float) * (size_t)n1 * n0, s)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(d_b1, 0, sizeof(float) * (size_t)n1, s)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(d_w2, 0, sizeof(float) * (size_t)n2 * n1, s)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(d_b2, 0, sizeof(float) * (size_t)n2, s)) != hipSuccess) return (int)e;
    return 0;
}

// dZ1 = (dZ2 W2) * gelu'(Z1), with dZ2 = dY * gelu'(Z2) formed in the A prologue and kept
int launch_dz1(int rows, int n1, int n2, const float* w2, const float* z1, const float* z2, int out_gelu,
               const float* d_y, long ld_dy, float* dz1, float* dz2buf, hipStream_t s, int save16 = 0) {
    GemmArgs g = blank();
    g.g16 = g.c16 = save16;
    g.A = d_y; g.a_rs = ld_dy; g.a_cs = 1;
    if (out_gelu) {
        if (ld_dy != n2) return PIT_ERR_SIZE;     // prologue reads z2 with d_y's indexing
        g.a_gz = z2; g.a_out = dz2buf; g.a_out_rs = n2; g.a_out_cs = 1;
    }
    g.B = w2; g.b_rs = n1; g.b_cs = 1;            // B(k,n) = w2[k][n]
    g.M = rows; g.N = n1; g.K = n2;
    g.G = z1; g.ldg = n1; g.C = dz1; g.ldc = n1; g.epi = EPI_MUL_GELU_GRAD;
    return launch_gemm(g, s);
}

}  // namespace

// the reductions of a postponed pit_mlp_bwd_params, laid out for a launch that carries them along
// (pit_posatt.hip: posatt_bwd_pair_dw_kernel); same tiling as launch_gemm_pair_atomic
bool pit_detail::plan_dw_pair(const pit_mlp_params_job& j, int waves, DwPair* out, int target_wgs, bool allow_rr) {
    static const bool off = exp_env("PIT_NO_DW_RIDER") != nullptr;
    if (off || !j.accumulate) return false;
    if (!j.x || !j.h || !j.d_y || !j.d_w1 || !j.d_b1 || !j.d_w2 || !j.d_b2 || !j.scratch) return false;
    if (j.rows <= 0 || j.n0 <= 0 || j.n1 <= 0 || j.n2 <= 0) return false;
    const float* dz2 = j.out_gelu ? j.scratch + (long)j.rows * j.n1 : j.d_y;
    const long ld_dz2 = j.out_gelu ? j.n2 : j.ld_dy;
    out->g1 = make_dw2(dz2, ld_dz2, j.h, j.rows, j.n1, j.n2, j.d_w2, j.d_b2);
    out->g2 = make_dw1(j.scratch, j.x, j.ldx, j.rows, j.n0, j.n1, j.d_w1, j.d_b1);
    if ((long)out->g1.M * out->g1.N * out->g1.K + (long)out->g2.M * out->g2.N * out->g2.K > (1L << 28)) return false;
    const int saved = t_call_math;
    t_call_math = j.math_mode;
    GemmLaunch L1, L2;
    const int fw = waves >= 8 ? 8 : (waves >= 4 ? 4 : 2);
    const bool ok = prepare_gemm(out->g1, L1, 1, fw, target_wgs) == 0 && prepare_gemm(out->g2, L2, 1, fw, target_wgs) == 0;
    t_call_math = saved;
    if (!ok) return false;
    out->n1 = L1.grid.x * L1.grid.y * L1.grid.z; out->n2 = L2.grid.x * L2.grid.y * L2.grid.z;
    out->gx1 = L1.grid.x; out->gy1 = L1.grid.y; out->gx2 = L2.grid.x; out->gy2 = L2.grid.y;
    out->rr1 = out->rr2 = 0;
    out->tx1 = out->tiles1 = out->slabs1 = out->tx2 = out->tiles2 = out->slabs2 = 0;
    out->nchunks = (j.rows + RR_BK - 1) / RR_BK;
    // a reduction with full 64-wide tiles rides as gemm_rr_tile tiles when the carrying kernel can run them: LDS-staged,
    // coalesced, a few chunks per workgroup - the register-direct form of the same reduction keeps 6-12x as many
    // workgroups busy for the same MACs and lengthens the launch it rides in
    static const bool no_rr = exp_env("PIT_NO_RR_RIDER") != nullptr;
    static const int per_wg = exp_env("PIT_RR_RIDER_CHUNKS") ? std::max(1, atoi(exp_env("PIT_RR_RIDER_CHUNKS"))) : 2;
    if (allow_rr && !no_rr && (j.math_mode & 0xff) == 0) {
        auto as_rr = [&](const GemmArgs& g, int& rr, int& tx, int& tiles, int& slabs, int& n) {
            const int n_real = g.ones_col >= 0 ? g.N - 1 : g.N;
            if (!gemm_rr_ok(g) || g.M < 32 || n_real < 48) return;
            rr = 1;
            tx = (n_real + 63) / 64;
            tiles = tx * ((g.M + 63) / 64);
            slabs = std::max(1, out->nchunks / per_wg);
            n = tiles * slabs;
        };
        as_rr(out->g1, out->rr1, out->tx1, out->tiles1, out->slabs1, out->n1);
        as_rr(out->g2, out->rr2, out->tx2, out->tiles2, out->slabs2, out->n2);
    }
    return true;
}

// the reductions of a postponed pit_mlp_bwd_params as gemm_rr_tile tiles for pit_satt.hip's backward launch: the chain MLPs' dW1 | dW2
// (Vorticity: 17 us as a launch of their own between the MLP's backward and the attention's, which does not depend on them)
bool pit_detail::plan_rr_rider(const pit_mlp_params_job& j, DwPair* out, int budget_wgs) {
    static const bool off = exp_env("PIT_NO_DW_RIDER") != nullptr;
    if (off || !j.accumulate) return false;
    if (!j.x || !j.h || !j.d_y || !j.d_w1 || !j.d_b1 || !j.d_w2 || !j.d_b2 || !j.scratch) return false;
    if (j.rows <= 0 || j.n0 <= 0 || j.n1 <= 0 || j.n2 <= 0 || (j.out_gelu && j.ld_dy != j.n2)) return false;
    const float* dz2 = j.out_gelu ? j.scratch + (long)j.rows * j.n1 : j.d_y;
    const long ld_dz2 = j.out_gelu ? j.n2 : j.ld_dy;
    *out = DwPair();
    out->g1 = make_dw2(dz2, ld_dz2, j.h, j.rows, j.n1, j.n2, j.d_w2, j.d_b2);
    out->g2 = make_dw1(j.scratch, j.x, j.ldx, j.rows, j.n0, j.n1, j.d_w1, j.d_b1);
    const int saved = t_call_math;
    t_call_math = j.math_mode;
    GemmLaunch L1, L2;
    const bool ok = prepare_gemm(out->g1, L1, 1, 4, 768) == 0 && prepare_gemm(out->g2, L2, 1, 4, 768) == 0;
    t_call_math = saved;
    if (!ok || !gemm_rr_ok(out->g1) || !gemm_rr_ok(out->g2)) return false;
    auto tiles_of = [](const GemmArgs& g, int& tx) {
        const int n_real = g.ones_col >= 0 ? g.N - 1 : g.N;
        tx = (n_real + 63) / 64;
        return tx * ((g.M + 63) / 64);
    };
    out->tiles1 = tiles_of(out->g1, out->tx1);
    out->tiles2 = tiles_of(out->g2, out->tx2);
    out->nchunks = (j.rows + RR_BK - 1) / RR_BK;
    const int slabs = std::max(1, std::min(std::max(1, out->nchunks / 2), budget_wgs / (out->tiles1 + out->tiles2)));
    out->rr1 = out->rr2 = 1;
    out->slabs1 = out->slabs2 = slabs;
    out->n1 = out->tiles1 * slabs;
    out->n2 = out->tiles2 * slabs;
    return true;
}

// include/pit_hip.h: the shapes whose forward / backward run entirely on the kernels that honour the PIT_IO_* flags (the
// thin output-layer kernels + gemm_bfl_kernel)
extern "C" int pit_mlp_bf16_io_supported(int rows, int n0, int n1, int n2, int out_gelu) {
    static const bool off = exp_env("PIT_NO_BF16_IO") != nullptr || exp_env("PIT_NO_THIN_GEMM") != nullptr ||
                            exp_env("PIT_BF16_LEGACY") != nullptr || (exp_env("PIT_LDS_GEMM") && atoi(exp_env("PIT_LDS_GEMM")) == 0);
    if (off || out_gelu) return 0;
    if (n2 < 1 || n2 > THIN_MAX || n0 % 8 != 0 || n1 % 8 != 0 || n0 < 48 || n1 < 48) return 0;
    return (long)rows * n1 >= (1L << 19) && rows >= 8192 && (long)rows * n1 * n0 >= (1L << 27);
}

// pit_mlp_slab.hip
bool try_launch_mlp_fwd64(const float* x, long ldx, int rows, int n0, int n1, int n2, const float* w1, const float* b1,
                          const float* w2, const float* b2, int out_gelu, float* z1, float* h, float* z2, float* y, long ldy,
                          hipStream_t s);
bool try_launch_mlp_bwd64(int rows, int n0, int n1, int n2, const float* w1, const float* w2, const float* z1, const float* z2,
                          int out_gelu, const float* d_y, long ld_dy, float* d_x, long ld_dx, float* dz1, float* dz2, hipStream_t s);
bool pit_mlp_slab_preferred(int rows, int n0, int n1, int n2);

extern "C" int pit_mlp_bwd_params_deferrable(int rows, int n0, int n1, int n2, int out_gelu, long ld_dy) {
    static const bool off = exp_env("PIT_NO_DW_RIDER") != nullptr;
    return !off && (!out_gelu || ld_dy == n2) && mlp_bwd16_eligible(rows, n0, n1, n2);
}

extern "C" int pit_mlp_fwd(const float* x, long ldx, int rows, int n0, int n1, int n2,
                           const float* w1, const float* b1, const float* w2, const float* b2, int out_gelu,
                           float* z1, float* h, float* z2, float* y, long ldy, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (!x || !w1 || !b1 || !w2 || !b2 || !z1 || !h || !y) return PIT_ERR_NULL;
    if (out_gelu && !z2) return PIT_ERR_NULL;
    if (rows <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0 || ldx < n0 || ldy < n2) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    const int x16 = (math_mode & PIT_IO_X_BF16) ? 1 : 0, save16 = (math_mode & PIT_IO_SAVE_BF16) ? 1 : 0;
    if (math_mode & ~(0xff | PIT_IO_X_BF16 | PIT_IO_SAVE_BF16)) return PIT_ERR_UNSUPPORTED;
    if ((x16 || save16) && !pit_mlp_bf16_io_supported(rows, n0, n1, n2, out_gelu)) return PIT_ERR_UNSUPPORTED;
    const bool slab_first = !x16 && !save16 && pit_mlp_slab_preferred(rows, n0, n1, n2);     // (exact fp32 like mlp_fwd16, in every math mode)
    if (slab_first && try_launch_mlp_fwd64(x, ldx, rows, n0, n1, n2, w1, b1, w2, b2, out_gelu, z1, h, z2, y, ldy, s)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    if (!x16 && !save16 && try_launch_mlp_fwd16(x, ldx, rows, n0, n1, n2, w1, b1, w2, b2, out_gelu, z1, h, z2, y, ldy, s)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    // large regime, hid 64, fp32: ONE fused launch on 64-row slabs staged through LDS (pit_mlp_slab.hip)
    if (!x16 && !save16 && (math_mode & 0xff) == PIT_MATH_FP32 &&
        try_launch_mlp_fwd64(x, ldx, rows, n0, n1, n2, w1, b1, w2, b2, out_gelu, z1, h, z2, y, ldy, s)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    GemmArgs g = blank();
    g.a16 = x16; g.z16 = g.c16 = save16;
    g.A = x; g.a_rs = ldx; g.a_cs = 1;
    g.B = w1; g.b_rs = 1; g.b_cs = n0;            // B(k,n) = w1[n][k]
    g.M = rows; g.N = n1; g.K = n0;
    g.bias = b1; g.Z = z1; g.ldz = n1; g.epi = EPI_BIAS_GELU; g.C = h; g.ldc = n1;
    if (int rc = launch_gemm(g, s)) return rc;
    PIT_CHECK_LAUNCH();
    g = blank();
    g.a16 = save16;
    g.A = h; g.a_rs = n1; g.a_cs = 1;
    g.B = w2; g.b_rs = 1; g.b_cs = n1;
    g.M = rows; g.N = n2; g.K = n1;
    g.bias = b2; g.C = y; g.ldc = ldy; g.epi = EPI_BIAS;
    if (out_gelu) { g.Z = z2; g.ldz = n2; g.epi = EPI_BIAS_GELU; }
    if (int rc = launch_gemm(g, s)) return rc;
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_mlp_bwd_data(int rows, int n0, int n1, int n2, const float* w1, const float* w2,
                                const float* z1, const float* z2, int out_gelu, const float* d_y, long ld_dy,
                                float* d_x, long ld_dx, float* scratch, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (!w1 || !w2 || !z1 || !d_y || !scratch) return PIT_ERR_NULL;
    if (out_gelu && !z2) return PIT_ERR_NULL;
    if (rows <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    float* dz1 = scratch;                       // rows * n1 (bf16 elements with PIT_IO_SAVE_BF16)
    float* dz2buf = scratch + (long)rows * n1;  // rows * n2
    const int save16 = (math_mode & PIT_IO_SAVE_BF16) ? 1 : 0, dx16 = (math_mode & PIT_IO_DX_BF16) ? 1 : 0;
    if (math_mode & ~(0xff | PIT_IO_X_BF16 | PIT_IO_SAVE_BF16 | PIT_IO_DX_BF16)) return PIT_ERR_UNSUPPORTED;
    if ((save16 || dx16) && !pit_mlp_bf16_io_supported(rows, n0, n1, n2, out_gelu)) return PIT_ERR_UNSUPPORTED;
    if (!save16 && !dx16 && (!out_gelu || ld_dy == n2) && pit_mlp_slab_preferred(rows, n0, n1, n2) &&
        try_launch_mlp_bwd64(rows, n0, n1, n2, w1, w2, z1, z2, out_gelu, d_y, ld_dy, d_x, ld_dx, dz1, dz2buf, s)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    if (!save16 && !dx16 && (!out_gelu || ld_dy == n2) &&
        try_launch_mlp_bwd16(rows, n0, n1, n2, w1, w2, z1, z2, out_gelu, d_y, ld_dy, d_x, ld_dx, dz1, dz2buf, s)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    if (!save16 && !dx16 && (math_mode & 0xff) == PIT_MATH_FP32 && (!out_gelu || ld_dy == n2) &&
        try_launch_mlp_bwd64(rows, n0, n1, n2, w1, w2, z1, z2, out_gelu, d_y, ld_dy, d_x, ld_dx, dz1, dz2buf, s)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    if (int rc = launch_dz1(rows, n1, n2, w2, z1, z2, out_gelu, d_y, ld_dy, dz1, dz2buf, s, save16)) return rc;
    PIT_CHECK_LAUNCH();
    if (d_x) {
        GemmArgs gx = make_dx(dz1, w1, rows, n0, n1, d_x, ld_dx);
        gx.a16 = save16; gx.c16 = dx16;
        if (int rc = launch_gemm(gx, s)) return rc;
        PIT_CHECK_LAUNCH();
    }
    return 0;
}

// y = x W^T (no bias): the fold of the decoder MLP's first layer into the up-projection's values (pit_fold.hip), and its
// backward d_x = d_y W, d_w (+)= d_y^T x - the GEMMs of pit_mlp_fwd / _bwd without an activation between them.
extern "C" int pit_linear_fwd(const float* x, long ldx, int rows, int n_in, int n_out, const float* w, const float* zero_bias,
                              float* y, long ldy, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (math_mode & ~0xff) return PIT_ERR_UNSUPPORTED;
    if (!x || !w || !zero_bias || !y) return PIT_ERR_NULL;
    if (rows <= 0 || n_in <= 0 || n_out <= 0 || ldx < n_in || ldy < n_out) return PIT_ERR_SIZE;
    GemmArgs g = blank();
    g.A = x; g.a_rs = ldx; g.a_cs = 1;
    g.B = w; g.b_rs = 1; g.b_cs = n_in;            // B(k,n) = w[n][k]
    g.M = rows; g.N = n_out; g.K = n_in;
    g.bias = zero_bias; g.C = y; g.ldc = ldy; g.epi = EPI_BIAS;
    if (int rc = launch_gemm(g, (hipStream_t)stream)) return rc;
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_linear_bwd(const float* x, long ldx, int rows, int n_in, int n_out, const float* w, const float* d_y, long ld_dy,
                              float* d_x, long ld_dx, float* d_w, int accumulate, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (math_mode & ~0xff) return PIT_ERR_UNSUPPORTED;
    if (!w || !d_y || (!d_x && !d_w) || (d_w && !x)) return PIT_ERR_NULL;
    if (rows <= 0 || n_in <= 0 || n_out <= 0 || ld_dy < n_out || (d_x && ld_dx < n_in) || (d_w && ldx < n_in)) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    if (d_x) {
        GemmArgs g = blank();
        g.A = d_y; g.a_rs = ld_dy; g.a_cs = 1;
        g.B = w; g.b_rs = n_in; g.b_cs = 1;        // B(k,n) = w[k][n]
        g.M = rows; g.N = n_in; g.K = n_out;
        g.C = d_x; g.ldc = ld_dx; g.epi = EPI_STORE;
        if (int rc = launch_gemm(g, s)) return rc;
        PIT_CHECK_LAUNCH();
    }
    if (d_w) {
        if (!accumulate) {
            hipError_t e = hipMemsetAsync(d_w, 0, sizeof(float) * (size_t)n_out * n_in, s);
            if (e != hipSuccess) return (int)e;
        }
        GemmArgs g = blank();
        g.A = d_y; g.a_rs = 1; g.a_cs = ld_dy;     // A(m,k) = d_y[k][m]
        g.B = x; g.b_rs = ldx; g.b_cs = 1;
        g.M = n_out; g.N = n_in; g.K = rows;
        g.C = d_w; g.ldc = n_in; g.atomic = 1; g.epi = EPI_ATOMIC;
        if (int rc = launch_gemm(g, s)) return rc;
        PIT_CHECK_LAUNCH();
    }
    return 0;
}

#ifdef PIT_STAMPS
extern "C" int pit_mlp_debug_read_stamps(unsigned long long* host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(pit_mlp_stamps), 64 * sizeof(unsigned long long));
}
extern "C" int pit_mlp_debug_read_wgrec(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pit_mlp_wgrec), (size_t)n * 4 * sizeof(unsigned long long));
}
extern "C" int pit_mlp_debug_set_rec_k(int k) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(pit_mlp_rec_k), &k, sizeof(int)); }
extern "C" int pit_mlp_debug_reset_stamps() {
    unsigned long long z[64] = {0};
    z[62] = ~0ull; z[56] = ~0ull;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(pit_mlp_stamps), z, sizeof(z));
}
#endif

extern "C" int pit_mlp_bwd_params(const float* x, long ldx, int rows, int n0, int n1, int n2, const float* h,
                                  int out_gelu, const float* d_y, long ld_dy,
                                  float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                  int accumulate, const float* scratch, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (!x || !h || !d_y || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !scratch) return PIT_ERR_NULL;
    if (rows <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    const float* dz1 = scratch;
    const float* dz2 = out_gelu ? scratch + (long)rows * n1 : d_y;
    const long ld_dz2 = out_gelu ? n2 : ld_dy;
    if (!accumulate)
        if (int rc = zero_param_grads(d_w1, d_b1, d_w2, d_b2, n0, n1, n2, s)) return rc;
    const int x16 = (math_mode & PIT_IO_X_BF16) ? 1 : 0, save16 = (math_mode & PIT_IO_SAVE_BF16) ? 1 : 0;
    if (math_mode & ~(0xff | PIT_IO_X_BF16 | PIT_IO_SAVE_BF16 | PIT_IO_DX_BF16)) return PIT_ERR_UNSUPPORTED;
    if ((x16 || save16) && !pit_mlp_bf16_io_supported(rows, n0, n1, n2, out_gelu)) return PIT_ERR_UNSUPPORTED;
    GemmArgs g2 = make_dw2(dz2, ld_dz2, h, rows, n1, n2, d_w2, d_b2);
    GemmArgs g1 = make_dw1(dz1, x, ldx, rows, n0, n1, d_w1, d_b1);
    g2.b16 = save16;                          // h
    g1.a16 = save16; g1.b16 = x16;            // dZ1, x
    if (int rc = launch_gemm_pair_atomic(g2, g1, s)) return rc;
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_mlp_bwd(const float* x, long ldx, int rows, int n0, int n1, int n2,
                           const float* w1, const float* w2, const float* z1, const float* h, const float* z2,
                           int out_gelu, const float* d_y, long ld_dy,
                           float* d_x, long ld_dx, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                           int accumulate, float* scratch, int math_mode, void* stream) {
    PIT_ENTER_MATH(math_mode);
    if (!x || !w1 || !w2 || !z1 || !h || !d_y || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !scratch) return PIT_ERR_NULL;
    if (out_gelu && !z2) return PIT_ERR_NULL;
    if (rows <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0) return PIT_ERR_SIZE;
    if (math_mode & ~0xff) {                   // bf16 storage: the two halves, one after the other
        if (int rc = pit_mlp_bwd_data(rows, n0, n1, n2, w1, w2, z1, z2, out_gelu, d_y, ld_dy, d_x, ld_dx, scratch, math_mode, stream)) return rc;
        return pit_mlp_bwd_params(x, ldx, rows, n0, n1, n2, h, out_gelu, d_y, ld_dy, d_w1, d_b1, d_w2, d_b2, accumulate, scratch,
                                  math_mode, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    float* dz1 = scratch;
    float* dz2buf = scratch + (long)rows * n1;
    if (!accumulate)
        if (int rc = zero_param_grads(d_w1, d_b1, d_w2, d_b2, n0, n1, n2, s)) return rc;
    const float* dz2 = out_gelu ? dz2buf : d_y;
    const long ld_dz2 = out_gelu ? n2 : ld_dy;
    const GemmArgs g2 = make_dw2(dz2, ld_dz2, h, rows, n1, n2, d_w2, d_b2);
    const GemmArgs g1 = make_dw1(dz1, x, ldx, rows, n0, n1, d_w1, d_b1);
    if ((!out_gelu || ld_dy == n2) && !(math_mode & ~0xff) && pit_mlp_slab_preferred(rows, n0, n1, n2) &&
        try_launch_mlp_bwd64(rows, n0, n1, n2, w1, w2, z1, z2, out_gelu, d_y, ld_dy, d_x, ld_dx, dz1, dz2buf, s)) {
        PIT_CHECK_LAUNCH();                              // (the 64-row slabs also beat the 16-row kernel once most CUs get a slab)
        if (int rc = launch_gemm_pair_atomic(g2, g1, s)) return rc;
        PIT_CHECK_LAUNCH();
        return 0;
    }
    if ((!out_gelu || ld_dy == n2) &&
        try_launch_mlp_bwd16(rows, n0, n1, n2, w1, w2, z1, z2, out_gelu, d_y, ld_dy, d_x, ld_dx, dz1, dz2buf, s)) {
        PIT_CHECK_LAUNCH();                              // dZ2, dZ1 and dX of every 16-row slab in one launch ...
        if (int rc = launch_gemm_pair_atomic(g2, g1, s)) return rc;      // ... then both weight-gradient reductions
        PIT_CHECK_LAUNCH();
        return 0;
    }
    if ((math_mode & 0xff) == PIT_MATH_FP32 && !(math_mode & ~0xff) && (!out_gelu || ld_dy == n2) &&
        try_launch_mlp_bwd64(rows, n0, n1, n2, w1, w2, z1, z2, out_gelu, d_y, ld_dy, d_x, ld_dx, dz1, dz2buf, s)) {
        PIT_CHECK_LAUNCH();                              // large regime, hid 64: the data path on 64-row slabs in one launch ...
        if (int rc = launch_gemm_pair_atomic(g2, g1, s)) return rc;      // ... then both weight-gradient reductions (gemm_rr_kernel)
        PIT_CHECK_LAUNCH();
        return 0;
    }
    if (int rc = launch_dz1(rows, n1, n2, w2, z1, z2, out_gelu, d_y, ld_dy, dz1, dz2buf, s)) return rc;
    PIT_CHECK_LAUNCH();
    // everything after dZ1 is mutually independent: one launch when small
    if (d_x) { if (int rc = launch_gemm_bwd_tail(make_dx(dz1, w1, rows, n0, n1, d_x, ld_dx), g2, g1, s)) return rc; }
    else if (int rc = launch_gemm_pair_atomic(g2, g1, s)) return rc;
    PIT_CHECK_LAUNCH();
    return 0;
}
