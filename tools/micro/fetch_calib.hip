// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern").  Each kernel reads the same N-byte buffer exactly once with one access shape of this
// repository's kernels; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and compare the counter with N:
//   read_dword     4 B/lane coalesced (global_load_dword)           - the attention kernels' value / d_out loads
//   read_dwordx4   16 B/lane coalesced (global_load_dwordx4)        - GEMM staging, mlp_fwd16 fragments
//   read_rows16    16 B/lane, 16 lanes per 256-B row, 4 rows/wave   - block_fwd/bwd_kernel value rows (slab_contract)
//   read_buf_dword 4 B/lane through a raw buffer descriptor         - posatt_rows_kernel
//   read_ushort    2 B/lane coalesced                               - bf16-stored d_out in the rows kernels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void read_dword(const float* p, long n, float* out) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void read_dwordx4(const float4* p, long n4, float* out) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
// rows of 64 floats (256 B): lane (c = l&15, kq = l>>4) reads 16 B of row 4*t + kq
__global__ void read_rows16(const float* p, long rows, float* out) {
    float s = 0.f;
    const int lane = threadIdx.x & 63, c = lane & 15, kq = lane >> 4;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long t = wave; t * 4 < rows; t += nw) { float4 v = *reinterpret_cast<const float4*>(p + (t * 4 + kq) * 64 + 4 * c); s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void read_buf_dword(const float* p, long n, float* out) {
    float s = 0.f;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n * 4), 0x00020000);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        s += __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4), 0, 0));
    if (s == 12345.678f) out[0] = s;
}
__global__ void read_ushort(const unsigned short* p, long n, float* out) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += (float)p[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void write_dword(float* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 1.0f;
}
__global__ void write_ushort(unsigned short* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 1;
}

int main(int argc, char** argv) {
    const long bytes = (argc > 1 ? atol(argv[1]) : 64L) << 20;      // MiB (default 64: inside the Infinity Cache; try 512 too)
    float *buf, *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 0, bytes);
    const long n = bytes / 4;
    const dim3 grid(4096), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_dword, grid, block, 0, 0, buf, n, out);
        hipLaunchKernelGGL(read_dwordx4, grid, block, 0, 0, (const float4*)buf, n / 4, out);
        hipLaunchKernelGGL(read_rows16, grid, block, 0, 0, buf, n / 64, out);
        hipLaunchKernelGGL(read_buf_dword, grid, block, 0, 0, buf, n > (1L << 29) ? (1L << 29) : n, out);
        hipLaunchKernelGGL(read_ushort, grid, block, 0, 0, (const unsigned short*)buf, bytes / 2, out);
        hipLaunchKernelGGL(write_dword, grid, block, 0, 0, buf, n);
        hipLaunchKernelGGL(write_ushort, grid, block, 0, 0, (unsigned short*)buf, bytes / 2);
    }
    hipDeviceSynchronize();
    printf("buffer %ld MiB = %ld KiB per kernel\n", bytes >> 20, bytes >> 10);
    return 0;
}
