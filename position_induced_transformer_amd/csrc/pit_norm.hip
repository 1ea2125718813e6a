// InstanceNorm1d over the point axis (train_vorticity.py:43,56,59: nn.InstanceNorm1d(hid_dim), no
// affine, no running statistics, biased variance, applied as norm(x.permute(0,2,1)).permute(0,2,1))
// directly on the (batch, points, channels) layout the attention / MLP kernels use - no permutes,
// no contiguous copies.  One workgroup owns 64 channels of one sample: 64 consecutive channels per
// row are one coalesced 256-B read, four row groups share the point axis; statistics in fp64.
#include "pit_common.h"

namespace {

__device__ __forceinline__ double block_sum4(double v, double* s, int cx, int rg) {
    s[rg * 64 + cx] = v;
    __syncthreads();
    const double t = s[cx] + s[64 + cx] + s[128 + cx] + s[192 + cx];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void instance_norm_fwd_kernel(const float* __restrict__ x, long ldx, long x_bstride,
                                                                 int npts, int nch, float eps, float* __restrict__ y,
                                                                 float* __restrict__ rstd_out) {
    __shared__ double s[256];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx, b = blockIdx.y;
    const bool cv = c < nch;
    const float* xb = x + (long)b * x_bstride + (cv ? c : 0);
    double sum = 0.0;
    for (int l = rg; l < npts; l += 4) sum += cv ? (double)xb[(long)l * ldx] : 0.0;
    const double mean = block_sum4(sum, s, cx, rg) / npts;
    double sq = 0.0;
    for (int l = rg; l < npts; l += 4) {
        const double d = cv ? (double)xb[(long)l * ldx] - mean : 0.0;
        sq += d * d;
    }
    const double var = block_sum4(sq, s, cx, rg) / npts;          // biased, as F.instance_norm
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    if (!cv) return;
    float* yb = y + ((long)b * npts) * nch + c;
    for (int l = rg; l < npts; l += 4) yb[(long)l * nch] = (xb[(long)l * ldx] - meanf) * rstd;
    if (rg == 0) rstd_out[(long)b * nch + c] = rstd;
}

// d_x = rstd * (d_y - mean_l(d_y) - y * mean_l(d_y * y)),  y = the normalised output
__global__ __launch_bounds__(256) void instance_norm_bwd_kernel(const float* __restrict__ d_y, const float* __restrict__ y,
                                                                 const float* __restrict__ rstd, int npts, int nch,
                                                                 float* __restrict__ d_x) {
    __shared__ double s[256];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx, b = blockIdx.y;
    const bool cv = c < nch;
    const long base = ((long)b * npts) * nch + (cv ? c : 0);
    double s1 = 0.0, s2 = 0.0;
    for (int l = rg; l < npts; l += 4) {
        const long e = base + (long)l * nch;
        const double g = cv ? (double)d_y[e] : 0.0;
        s1 += g;
        s2 += g * (cv ? (double)y[e] : 0.0);
    }
    const float m1 = (float)(block_sum4(s1, s, cx, rg) / npts);
    const float m2 = (float)(block_sum4(s2, s, cx, rg) / npts);
    if (!cv) return;
    const float r = rstd[(long)b * nch + c];
    for (int l = rg; l < npts; l += 4) {
        const long e = base + (long)l * nch;
        d_x[e] = r * (d_y[e] - m1 - y[e] * m2);
    }
}

}  // namespace

extern "C" int pit_instance_norm_fwd(const float* x, long ldx, long x_bstride, int batch, int npts, int nch, float eps,
                                     float* y, float* rstd, void* stream) {
    if (!x || !y || !rstd) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || ldx < nch || batch > 65535) return PIT_ERR_SIZE;
    hipLaunchKernelGGL(instance_norm_fwd_kernel, dim3((nch + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       x_bstride, npts, nch, eps, y, rstd);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_instance_norm_bwd(const float* d_y, const float* y, const float* rstd, int batch, int npts, int nch,
                                     float* d_x, void* stream) {
    if (!d_y || !y || !rstd || !d_x) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || batch > 65535) return PIT_ERR_SIZE;
    hipLaunchKernelGGL(instance_norm_bwd_kernel, dim3((nch + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, d_y, y,
                       rstd, npts, nch, d_x);
    PIT_CHECK_LAUNCH();
    return 0;
}
