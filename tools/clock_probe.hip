// Effective-clock probe: a dependent chain of N v_mfma_f32_32x32x2_f32 (64 cycles each per SIMD)
// and of N v_fma_f32, timed with events, at two grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void mfma_chain(float* out, int n) {
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[5];
}
__global__ void fma_chain(float* out, int n) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, y, 0.5f);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void clk(unsigned long long* o) { unsigned long long t0 = clock64(), r0 = wall_clock64();
    float x = threadIdx.x; for (int i = 0; i < 200000; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    unsigned long long t1 = clock64(), r1 = wall_clock64(); o[0] = t1 - t0; o[1] = r1 - r0; o[2] = (unsigned long long)x; }
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    unsigned long long* o; hipMalloc(&o, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    int grids[3][2] = {{256, 64}, {256, 256}, {2048, 256}};
    for (int rep = 0; rep < 2; ++rep)
    for (auto& g : grids) {
        for (int k = 0; k < 2; ++k) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(mfma_chain, dim3(g[0]), dim3(g[1]), 0, 0, out, n);
            else hipLaunchKernelGGL(fma_chain, dim3(g[0]), dim3(g[1]), 0, 0, out, n * 16);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per = ms * 1e6 / (k == 0 ? n : n * 16);
            printf("%s grid %4d x %3d : %8.3f ms  -> %7.2f ns per op (%s)\n", k == 0 ? "mfma32x32x2" : "v_fma      ", g[0], g[1], ms, per,
                   k == 0 ? "64 cycles/SIMD" : "dependent ~4-8 cycles");
        }
    }
    hipLaunchKernelGGL(clk, dim3(1), dim3(64), 0, 0, o); hipDeviceSynchronize();
    unsigned long long h[3]; hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("clock64 ticks %llu, wall_clock64 ticks %llu (100 MHz) -> shader clock %.1f MHz (1 wave)\n", h[0], h[1], h[0] * 100.0 / h[1]);
    return 0;
}
