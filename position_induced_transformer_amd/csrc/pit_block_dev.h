// Device pieces shared by the fused processor kernels: the per-block launches (pit_block.hip) and the persistent latent
// kernels (pit_latent.hip).  See pit_block.hip for the design notes.
#pragma once
#include "pit_common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma_16x16x4(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int BD = 64;                 // value width (hid_dim) of the fused path: 4 interleaved 16-column tiles per wave
constexpr int BW = 8;                  // waves per workgroup
constexpr int MAX_LAYERS = 16;

// Workgroup id -> (sample, slab) so that all slabs of a sample run on ONE XCD (workgroups are dealt round-robin to the 8
// XCDs): a sample's activations then live in that XCD's L2 from one block's launch to the next.  The grid has
// 8 * ceil(batch / 8) * slabs workgroups; ids whose sample is beyond the batch return false.
__device__ __forceinline__ bool slab_of_linear(int id, int batch, int slabs, int& sample, int& slab) {
    sample = id / slabs; slab = id % slabs;
    return sample < batch;
}
__device__ __forceinline__ bool slab_of_xcd(int id, int batch, int slabs, int& sample, int& slab) {
    const int x = id & 7, k = id >> 3;
    sample = x + 8 * (k / slabs);
    slab = k % slabs;
    return sample < batch;
}
__host__ __device__ inline int slab_grid(int batch, int slabs) { return 8 * ((batch + 7) / 8) * slabs; }

// park a 16 x 64 partial tile in slot `slot`: [(slot*4 + t)*4 + i][lane]
__device__ __forceinline__ void park(float* pk, int slot, int lane, const f32x4_t (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[((slot * 4 + t) * 4 + i) * 64 + lane] = acc[t][i];
}
// element (row 4*(lane>>4) + i, columns 4*(lane&15) .. +3) summed over slots w0, w0 + step, ... (nw of them)
__device__ __forceinline__ float4 parked_sum(const float* pk, int w0, int nw, int step, int i, int lane) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0, w = w0; q < nw; ++q, w += step) {
        s.x += pk[((w * 4 + 0) * 4 + i) * 64 + lane];
        s.y += pk[((w * 4 + 1) * 4 + i) * 64 + lane];
        s.z += pk[((w * 4 + 2) * 4 + i) * 64 + lane];
        s.w += pk[((w * 4 + 3) * 4 + i) * 64 + lane];
    }
    return s;
}

constexpr int PARK_FLOATS = BW * 16 * 64;              // 32 KiB: one 16 x 64 tile per wave

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
