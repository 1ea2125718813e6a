// Diagnostic: sustained rate of the matrix instructions the attention / GEMM kernels use, bare (no loads, no VALU).
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, const float* rnd) {
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int t = 0; t < NACC; ++t) { for (int i = 0; i < 16; ++i) acc[t][i] = 0.f; for (int i = 0; i < 4; ++i) acc4[t][i] = 0.f; }
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    float ra[8], rb[8];                                   // rnd != null: random operands (data-dependent power -> clock)
    for (int i = 0; i < 8; ++i) { ra[i] = rnd ? rnd[(threadIdx.x * 8 + i) & 4095] : a; rb[i] = rnd ? rnd[(threadIdx.x * 8 + i + 977) & 4095] : b; }
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)ra[i]; bb[i] = (__bf16)rb[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[r], rb[(r + t) & 7], acc[t], 0, 0, 0);
                else if (KIND == 1) acc4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[r], rb[(r + t) & 7], acc4[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) { for (int i = 0; i < 16; ++i) s += acc[t][i]; for (int i = 0; i < 4; ++i) s += acc4[t][i]; }
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int NACC>
void run(const char* name, double flop_per_mfma, int wgs_per_cu, bool random = false) {
    float* out; hipMalloc(&out, 4);
    float* rnd = nullptr;
    if (random) {
        float h[4096];
        unsigned s = 12345u;
        for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 16777216.0f - 0.5f) * 4.0f; }
        hipMalloc(&rnd, sizeof(h)); hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    }
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, rnd);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * 8 * NACC;
    printf("%-28s %s acc=%d  %d WG(4 waves)/CU: %8.1f TF/s   %6.1f cycles/MFMA/SIMD at 2.4 GHz\n", name, random ? "random" : "const ", NACC, wgs_per_cu,
           mfmas * flop_per_mfma / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (mfmas / 1024.0));
    hipFree(out);
}
int main() {
    run<0, 4>("v_mfma_f32_32x32x2_f32", 4096, 1);
    run<0, 4>("v_mfma_f32_32x32x2_f32", 4096, 2);
    run<0, 4>("v_mfma_f32_32x32x2_f32", 4096, 4);
    run<0, 1>("v_mfma_f32_32x32x2_f32", 4096, 2);
    run<1, 4>("v_mfma_f32_16x16x4_f32", 2048, 2);
    run<1, 4>("v_mfma_f32_16x16x4_f32", 2048, 4);
    run<2, 4>("v_mfma_f32_32x32x16_bf16", 32768, 2);
    run<2, 4>("v_mfma_f32_32x32x16_bf16", 32768, 4);
    run<0, 4>("v_mfma_f32_32x32x2_f32", 4096, 2, true);
    run<0, 4>("v_mfma_f32_32x32x2_f32", 4096, 4, true);
    run<1, 4>("v_mfma_f32_16x16x4_f32", 2048, 4, true);
    run<2, 4>("v_mfma_f32_32x32x16_bf16", 32768, 4, true);
    return 0;
}
