// Diagnostic: what, beside v_mfma_f32_32x32x2_f32, keeps the matrix pipe from its bare rate?  One ingredient of the
// attention key loop at a time: a fresh global dword per MFMA as the B operand, vector ALU work per MFMA, an LDS
// read per 4 MFMAs, a workgroup barrier per 32 MFMAs.
// hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LOADS, int VALU, int LDS, bool BARRIER>
__global__ __launch_bounds__(256) void k(float* out, const float* vals, int iters, int ld) {
    __shared__ float4 s_p[8 * 64];
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64; i += 256) s_p[i] = make_float4(i * 0.001f, 0.5f, 0.25f, 0.125f);
    __syncthreads();
    // LOADS 1: one dword per MFMA (32 adjacent columns per half-wave); 2: one dwordx4 per 4 MFMAs (4 adjacent columns per lane)
    const float* base = vals + (LOADS == 2 ? (size_t)(blockIdx.x % 15) * 128 + (lane & 31) * 4 : (size_t)(blockIdx.x % 61) * 32 + (lane & 31)) + (size_t)(lane >> 5) * ld;
    float x = lane * 0.01f, y = 0.3f, z = 0.f;
    float b[32];
    for (int u = 0; u < 32; ++u) b[u] = LOADS == 1 ? base[(size_t)(2 * u) * ld] : x + u;
    if (LOADS == 2)
        for (int g = 0; g < 8; ++g) { const float4 q = *reinterpret_cast<const float4*>(base + (size_t)(2 * g) * ld); b[4 * g] = q.x; b[4 * g + 1] = q.y; b[4 * g + 2] = q.z; b[4 * g + 3] = q.w; }
    for (int it = 0; it < iters; ++it) {
        float nb[32];
        const float* row = base + (size_t)((it + 1) & 15) * 64 * ld;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float4 pa = LDS == 1 ? s_p[g * 64 + lane] : make_float4(x, y, x, y);
            const float* sp = reinterpret_cast<const float*>(s_p);
            if (LDS == 2) { pa.x = sp[(g * 4 + 0) * 64 + lane]; pa.y = sp[(g * 4 + 1) * 64 + lane]; pa.z = sp[(g * 4 + 2) * 64 + lane]; pa.w = sp[(g * 4 + 3) * 64 + lane]; }   // a b32 read per MFMA
            const float pe[4] = {pa.x, pa.y, pa.z, pa.w};
            if (LOADS == 2) { const float4 q = *reinterpret_cast<const float4*>(row + (size_t)(2 * g) * ld); nb[4 * g] = q.x; nb[4 * g + 1] = q.y; nb[4 * g + 2] = q.z; nb[4 * g + 3] = q.w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int u = g * 4 + e;
                if (LOADS == 1) nb[u] = row[(size_t)(2 * u) * ld];
#pragma unroll
                for (int v = 0; v < VALU; ++v) { z = z * 1.0001f + x; x = x * 0.9999f + (v & 1 ? y : z); }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pe[e] + (VALU ? z : 0.f), b[u], acc, 0, 0, 0);
            }
        }
        if (LOADS) {
#pragma unroll
            for (int u = 0; u < 32; ++u) b[u] = nb[u];
        }
        if (BARRIER) __syncthreads();
    }
    float s = z;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

template <int LOADS, int VALU, int LDS, bool BARRIER>
void run(const char* name, int wgs_per_cu, const float* vals, int ld) {
    float* out; (void)hipMalloc(&out, 4);
    const int iters = 400, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<LOADS, VALU, LDS, BARRIER>), dim3(grid), dim3(256), 0, 0, out, vals, iters, ld);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfmas = (double)grid * 4 * iters * 32;
    printf("%-44s %d waves/SIMD: %7.1f TF/s\n", name, wgs_per_cu, mfmas * 4096 / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    const int ld = 256, rows = 64 * 16 + 64;
    float* vals; (void)hipMalloc(&vals, (size_t)rows * ld * 4 + 4096 * 4); (void)hipMemset(vals, 0, (size_t)rows * ld * 4 + 4096 * 4);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0, 0, 0, false>("bare", 1, vals, ld); run<1, 0, 0, false>("+ global dword per MFMA (B operand)", 1, vals, ld);
                      run<0, 6, 0, false>("+ 12 VALU per MFMA", 1, vals, ld); run<0, 3, 0, false>("+ 6 VALU per MFMA", 1, vals, ld);
                      run<0, 0, 1, false>("+ LDS b128 per 4 MFMAs", 1, vals, ld); run<0, 0, 2, false>("+ LDS b32 per MFMA", 1, vals, ld); run<2, 1, 2, true>("dwordx4 loads, 2 VALU, LDS b32 per MFMA", 1, vals, ld); run<2, 1, 1, true>("dwordx4 loads, 2 VALU, LDS b128 per 4", 1, vals, ld); run<0, 0, 0, true>("+ barrier per 32 MFMAs", 1, vals, ld);
                      run<1, 3, 1, true>("all (6 VALU)", 1, vals, ld); run<2, 0, 0, false>("+ global dwordx4 per 4 MFMAs", 1, vals, ld); run<2, 3, 1, true>("all, dwordx4 loads (6 VALU)", 1, vals, ld); run<2, 2, 1, true>("all, dwordx4 loads (4 VALU)", 1, vals, ld); }
        if (w == 2) { run<0, 0, 0, false>("bare", 2, vals, ld); run<1, 0, 0, false>("+ global dword per MFMA (B operand)", 2, vals, ld);
                      run<0, 6, 0, false>("+ 12 VALU per MFMA", 2, vals, ld); run<0, 3, 0, false>("+ 6 VALU per MFMA", 2, vals, ld);
                      run<0, 0, 1, false>("+ LDS b128 per 4 MFMAs", 2, vals, ld); run<0, 0, 2, false>("+ LDS b32 per MFMA", 2, vals, ld); run<2, 1, 2, true>("dwordx4 loads, 2 VALU, LDS b32 per MFMA", 2, vals, ld); run<2, 1, 1, true>("dwordx4 loads, 2 VALU, LDS b128 per 4", 2, vals, ld); run<0, 0, 0, true>("+ barrier per 32 MFMAs", 2, vals, ld);
                      run<1, 3, 1, true>("all (6 VALU)", 2, vals, ld); run<2, 0, 0, false>("+ global dwordx4 per 4 MFMAs", 2, vals, ld); run<2, 3, 1, true>("all, dwordx4 loads (6 VALU)", 2, vals, ld); run<2, 2, 1, true>("all, dwordx4 loads (4 VALU)", 2, vals, ld); }
        if (w == 4) { run<0, 0, 0, false>("bare", 4, vals, ld); run<1, 0, 0, false>("+ global dword per MFMA (B operand)", 4, vals, ld);
                      run<0, 6, 0, false>("+ 12 VALU per MFMA", 4, vals, ld); run<0, 3, 0, false>("+ 6 VALU per MFMA", 4, vals, ld);
                      run<0, 0, 1, false>("+ LDS b128 per 4 MFMAs", 4, vals, ld); run<0, 0, 2, false>("+ LDS b32 per MFMA", 4, vals, ld); run<2, 1, 2, true>("dwordx4 loads, 2 VALU, LDS b32 per MFMA", 4, vals, ld); run<2, 1, 1, true>("dwordx4 loads, 2 VALU, LDS b128 per 4", 4, vals, ld); run<0, 0, 0, true>("+ barrier per 32 MFMAs", 4, vals, ld);
                      run<1, 3, 1, true>("all (6 VALU)", 4, vals, ld); run<2, 0, 0, false>("+ global dwordx4 per 4 MFMAs", 4, vals, ld); run<2, 3, 1, true>("all, dwordx4 loads (6 VALU)", 4, vals, ld); run<2, 2, 1, true>("all, dwordx4 loads (4 VALU)", 4, vals, ld); }
    }
    return 0;
}
