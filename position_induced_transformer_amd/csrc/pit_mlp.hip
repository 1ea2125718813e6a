// Pointwise MLP (kaiming_mlp, pit.py:13-26) forward and backward on fp32 MFMA.
//
// One LDS-tiled GEMM template (v_mfma_f32_32x32x2_f32, exact fp32 fma chains) with
// strided operand views and fused epilogues covers every contraction of the layer:
//   forward   Z1 = X W1^T + b1, H = gelu(Z1);   Y = H W2^T + b2 (optionally gelu, Z2 kept)
//   backward  dZ1 = (dZ2 W2) * gelu'(Z1);  dX = dZ1 W1;
//             dW2 = dZ2^T H, db2 = colsum(dZ2);  dW1 = dZ1^T X, db1 = colsum(dZ1)
// The weight-gradient GEMMs reduce over the row axis: they are split over row slabs
// (blockIdx.z) and accumulated with fp32 atomics; the bias gradient rides along as a
// virtual all-ones column of the B operand.
//
// Tiles: both operands are staged k-major in LDS (As[k][m], Bs[k][n]) so a fragment read
// is 32 consecutive floats per half-wave (conflict-free ds_read_b32); the next k-slab is
// prefetched into registers while the current one feeds the MFMAs.
#include "pit_common.h"

namespace {

struct GemmArgs {
    const float* A; long a_rs, a_cs;     // A(m,k) = A[m*a_rs + k*a_cs]
    const float* B; long b_rs, b_cs;     // B(k,n) = B[k*b_rs + n*b_cs]
    int M, N, K;
    int k_slab;                          // K range per blockIdx.z
    const float* bias;                   // [N] or null
    float* C; long ldc;
    float* Z; long ldz;                  // optional pre-activation copy
    const float* G; long ldg;            // optional: multiply result by gelu'(G[m,n])
    int act;                             // 1: C = gelu(.)
    int atomic;                          // 1: atomicAdd into C / C2
    int ones_col;                        // >= 0: B(k, ones_col) == 1, that output column goes to C2[m]
    float* C2;
};

constexpr int BK = 16;

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr int RA = BM * BK / 256, RB = BN * BK / 256;
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_slab;
    const int kend = min(g.K, kbeg + g.k_slab);

    const bool a_kfast = (g.a_cs == 1);     // k contiguous in memory -> k fastest over threads
    const bool b_kfast = (g.b_rs == 1);

    float ra[RA], rb[RB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int e = tid + r * 256;
            const int mm = a_kfast ? e / BK : e % BM;
            const int kk = a_kfast ? e % BK : e / BM;
            const int m = m0 + mm, k = k0 + kk;
            ra[r] = (m < g.M && k < kend) ? g.A[(long)m * g.a_rs + (long)k * g.a_cs] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int e = tid + r * 256;
            const int nn = b_kfast ? e / BK : e % BN;
            const int kk = b_kfast ? e % BK : e / BN;
            const int n = n0 + nn, k = k0 + kk;
            float v = 0.0f;
            if (n < g.N && k < kend) v = (n == g.ones_col) ? 1.0f : g.B[(long)k * g.b_rs + (long)n * g.b_cs];
            rb[r] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int e = tid + r * 256;
            const int mm = a_kfast ? e / BK : e % BM;
            const int kk = a_kfast ? e % BK : e / BM;
            As[kk * LDA + mm] = ra[r];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int e = tid + r * 256;
            const int nn = b_kfast ? e / BK : e % BN;
            const int kk = b_kfast ? e % BK : e / BN;
            Bs[kk * LDB + nn] = rb[r];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (kbeg < kend) {
        fetch(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            __syncthreads();
            stash();
            __syncthreads();
            if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[(kk + half) * LDA + (wm * TM + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = Bs[(kk + half) * LDB + (wn * TN + j) * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x2(af[i], bf[j], acc[i][j]);
            }
        }
    }

    // ---- epilogue
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + l31;
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + acc_row(r, half);
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.Z) g.Z[(long)row * g.ldz + col] = v;
                if (g.act) v = gelu_erf(v);
                if (g.G) v *= gelu_erf_grad(g.G[(long)row * g.ldg + col]);
                if (col == g.ones_col) {
                    if (g.atomic) atomicAdd(g.C2 + row, v); else g.C2[row] = v;
                } else {
                    float* dst = g.C + (long)row * g.ldc + col;
                    if (g.atomic) atomicAdd(dst, v); else *dst = v;
                }
            }
        }
}

void launch_gemm(GemmArgs g, int target_wgs, hipStream_t s) {
    // tile shape by output width
    int bm, bn;
    if (g.N > 64) { bm = 64; bn = 128; }
    else if (g.N > 32) { bm = 64; bn = 64; }
    else { bm = 128; bn = 32; }
    const int tiles = ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
    int splits = 1;
    if (g.atomic) {
        splits = max(1, min((g.K + 4 * BK - 1) / (4 * BK), (target_wgs + tiles - 1) / tiles));
    }
    int slab = (g.K + splits - 1) / splits;
    slab = ((slab + BK - 1) / BK) * BK;
    splits = (g.K + slab - 1) / slab;
    g.k_slab = slab;
    dim3 grid((g.N + bn - 1) / bn, (g.M + bm - 1) / bm, splits), block(256);
    if (g.N > 64) hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 2>), grid, block, 0, s, g);
    else if (g.N > 32) hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<4, 1, 1, 1>), grid, block, 0, s, g);
}

GemmArgs blank() {
    GemmArgs g{};
    g.ones_col = -1;
    return g;
}

__global__ void gelu_bwd_kernel(const float* __restrict__ dy, long ld_dy, const float* __restrict__ z, int rows,
                                int n, float* __restrict__ dz) {
    const long total = (long)rows * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / n;
        const int c = (int)(e - r * n);
        dz[e] = dy[r * ld_dy + c] * gelu_erf_grad(z[e]);
    }
}

}  // namespace

extern "C" int pit_mlp_fwd(const float* x, long ldx, int rows, int n0, int n1, int n2,
                           const float* w1, const float* b1, const float* w2, const float* b2, int out_gelu,
                           float* z1, float* h, float* z2, float* y, long ldy, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !z1 || !h || !y) return PIT_ERR_NULL;
    if (out_gelu && !z2) return PIT_ERR_NULL;
    if (rows <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0 || ldx < n0 || ldy < n2) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g = blank();
    g.A = x; g.a_rs = ldx; g.a_cs = 1;
    g.B = w1; g.b_rs = 1; g.b_cs = n0;            // B(k,n) = w1[n][k]
    g.M = rows; g.N = n1; g.K = n0;
    g.bias = b1; g.Z = z1; g.ldz = n1; g.act = 1; g.C = h; g.ldc = n1;
    launch_gemm(g, 0, s);
    PIT_CHECK_LAUNCH();
    g = blank();
    g.A = h; g.a_rs = n1; g.a_cs = 1;
    g.B = w2; g.b_rs = 1; g.b_cs = n1;
    g.M = rows; g.N = n2; g.K = n1;
    g.bias = b2; g.C = y; g.ldc = ldy;
    if (out_gelu) { g.Z = z2; g.ldz = n2; g.act = 1; }
    launch_gemm(g, 0, s);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_mlp_bwd(const float* x, long ldx, int rows, int n0, int n1, int n2,
                           const float* w1, const float* w2, const float* z1, const float* h, const float* z2,
                           int out_gelu, const float* d_y, long ld_dy,
                           float* d_x, long ld_dx, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                           float* scratch, void* stream) {
    if (!x || !w1 || !w2 || !z1 || !h || !d_y || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !scratch) return PIT_ERR_NULL;
    if (out_gelu && !z2) return PIT_ERR_NULL;
    if (rows <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    float* dz1 = scratch;                       // rows * n1
    float* dz2buf = scratch + (long)rows * n1;  // rows * n2
    const float* dz2 = d_y;
    long ld_dz2 = ld_dy;
    if (out_gelu) {
        const long total = (long)rows * n2;
        const int blocks = (int)std::min<long>((total + 255) / 256, 2048L);
        hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, s, d_y, ld_dy, z2, rows, n2, dz2buf);
        PIT_CHECK_LAUNCH();
        dz2 = dz2buf; ld_dz2 = n2;
    }
    hipError_t e;
    if ((e = hipMemsetAsync(d_w1, 0, sizeof(float) * (size_t)n1 * n0, s)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(d_b1, 0, sizeof(float) * (size_t)n1, s)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(d_w2, 0, sizeof(float) * (size_t)n2 * n1, s)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(d_b2, 0, sizeof(float) * (size_t)n2, s)) != hipSuccess) return (int)e;

    // dZ1 = (dZ2 W2) * gelu'(Z1)
    GemmArgs g = blank();
    g.A = dz2; g.a_rs = ld_dz2; g.a_cs = 1;
    g.B = w2; g.b_rs = n1; g.b_cs = 1;            // B(k,n) = w2[k][n]
    g.M = rows; g.N = n1; g.K = n2;
    g.G = z1; g.ldg = n1; g.C = dz1; g.ldc = n1;
    launch_gemm(g, 0, s);
    PIT_CHECK_LAUNCH();
    // dW2 = dZ2^T H (+ db2 as the ones column)
    g = blank();
    g.A = dz2; g.a_rs = 1; g.a_cs = ld_dz2;        // A(m,k) = dz2[k][m]
    g.B = h; g.b_rs = n1; g.b_cs = 1;
    g.M = n2; g.N = n1 + 1; g.K = rows; g.ones_col = n1;
    g.C = d_w2; g.ldc = n1; g.C2 = d_b2; g.atomic = 1;
    launch_gemm(g, 512, s);
    PIT_CHECK_LAUNCH();
    // dX = dZ1 W1
    if (d_x) {
        g = blank();
        g.A = dz1; g.a_rs = n1; g.a_cs = 1;
        g.B = w1; g.b_rs = n0; g.b_cs = 1;
        g.M = rows; g.N = n0; g.K = n1;
        g.C = d_x; g.ldc = ld_dx;
        launch_gemm(g, 0, s);
        PIT_CHECK_LAUNCH();
    }
    // dW1 = dZ1^T X (+ db1)
    g = blank();
    g.A = dz1; g.a_rs = 1; g.a_cs = n1;
    g.B = x; g.b_rs = ldx; g.b_cs = 1;
    g.M = n1; g.N = n0 + 1; g.K = rows; g.ones_col = n0;
    g.C = d_w1; g.ldc = n0; g.C2 = d_b1; g.atomic = 1;
    launch_gemm(g, 512, s);
    PIT_CHECK_LAUNCH();
    return 0;
}
