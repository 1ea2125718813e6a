// Version / error text / layout probe of the C ABI (include/pit_hip.h).
#include "pit_common.h"

namespace {
// D(32x32) = A(32x8) * B(8x32) with the fragment maps documented in pit_common.h.
__global__ void mfma_probe_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k += 2) acc = mfma_32x32x2(a[l31 * 8 + k + half], b[(k + half) * 32 + l31], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[acc_row(r, half) * 32 + l31] = acc[r];
}
}  // namespace

thread_local int t_call_math = PIT_MATH_FP32;

extern "C" int pit_version(void) { return PIT_ABI_VERSION; }

extern "C" const char* pit_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case PIT_ERR_NULL: return "pit: required pointer is NULL";
        case PIT_ERR_SIZE: return "pit: invalid size / stride argument";
        case PIT_ERR_METRIC: return "pit: unknown metric";
        case PIT_ERR_UNSUPPORTED: return "pit: unsupported configuration";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "pit: unknown error";
    }
}

extern "C" int pit_debug_mfma_tile(const float* a, const float* b, float* d, void* stream) {
    if (!a || !b || !d) return PIT_ERR_NULL;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
    PIT_CHECK_LAUNCH();
    return 0;
}
