// InstanceNorm1d over the point axis (train_vorticity.py:43,56,59: nn.InstanceNorm1d(hid_dim), no
// affine, no running statistics, biased variance, applied as norm(x.permute(0,2,1)).permute(0,2,1))
// directly on the (batch, points, channels) layout the attention / MLP kernels use - no permutes,
// no contiguous copies.  One workgroup owns 64 channels of one sample (coalesced 256-B row segments,
// 16-B per thread when alignment allows), its row groups share the point axis; statistics in fp64.
#include "pit_common.h"

namespace {

// 256 threads = CG channel groups x RG row groups; a thread owns VW consecutive channels.
// VW = 4 (16-B accesses, 64 channels x 16 row groups per workgroup) when the channel count and the
// strides allow it, else VW = 1 (64 channels x 4 row groups).
template <int VW>
struct NormShape {
    static constexpr int CG = (VW == 4) ? 16 : 64;
    static constexpr int RG = 256 / CG;
};

template <int VW>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VW]) {
    if (VW == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        v[0] = *p;
    }
}
template <int VW>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VW]) {
    if (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *p = v[0];
}

// sum over the row groups of per-thread partials (VW values each), result broadcast to all
template <int VW>
__device__ __forceinline__ void group_sum(double (&v)[VW], double* s, int cg, int rg) {
    constexpr int CG = NormShape<VW>::CG, RG = NormShape<VW>::RG;
#pragma unroll
    for (int e = 0; e < VW; ++e) s[(rg * CG + cg) * VW + e] = v[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        double t = 0.0;
        for (int r = 0; r < RG; ++r) t += s[(r * CG + cg) * VW + e];
        v[e] = t;
    }
    __syncthreads();
}

// RES (npts <= RES_ROWS * RG, the latent meshes of the Vorticity model: 256 points): a thread's rows stay in
// registers - ONE pass over memory with every load in flight at once instead of three dependent loops of loads;
// the sums are formed in the same order, so the results are identical bit for bit.
constexpr int RES_ROWS = 16;

template <int VW, bool RES = false>
__global__ __launch_bounds__(256) void instance_norm_fwd_kernel(const float* __restrict__ x, long ldx, long x_bstride,
                                                                 int npts, int nch, float eps, float* __restrict__ y,
                                                                 float* __restrict__ rstd_out) {
    constexpr int CG = NormShape<VW>::CG, RG = NormShape<VW>::RG;
    __shared__ double s[256 * VW];
    const int cg = threadIdx.x % CG, rg = threadIdx.x / CG;
    const int c = blockIdx.x * 64 + cg * VW, b = blockIdx.y;
    const bool cv = c < nch;                                   // VW divides nch: whole vectors are valid
    const float* xb = x + (long)b * x_bstride + (cv ? c : 0);
    double acc[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) acc[e] = 0.0;
    if (RES) {
        float xv[RES_ROWS][VW];
#pragma unroll
        for (int i = 0; i < RES_ROWS; ++i) {
            const int l = rg + i * RG;
            if (l < npts) load_vec<VW>(xb + (long)l * ldx, xv[i]);
            else {
#pragma unroll
                for (int e = 0; e < VW; ++e) xv[i][e] = 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < RES_ROWS; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[e] += (cv && rg + i * RG < npts) ? (double)xv[i][e] : 0.0;
        group_sum<VW>(acc, s, cg, rg);
        double mean[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) { mean[e] = acc[e] / npts; acc[e] = 0.0; }
#pragma unroll
        for (int i = 0; i < RES_ROWS; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const double d = (cv && rg + i * RG < npts) ? (double)xv[i][e] - mean[e] : 0.0;
                acc[e] += d * d;
            }
        group_sum<VW>(acc, s, cg, rg);
        float rstd[VW], meanf[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            rstd[e] = (float)(1.0 / sqrt(acc[e] / npts + (double)eps));
            meanf[e] = (float)mean[e];
        }
        if (!cv) return;
        float* yb = y + ((long)b * npts) * nch + c;
#pragma unroll
        for (int i = 0; i < RES_ROWS; ++i) {
            const int l = rg + i * RG;
            if (l >= npts) break;
            float v[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] = (xv[i][e] - meanf[e]) * rstd[e];
            store_vec<VW>(yb + (long)l * nch, v);
        }
        if (rg == 0) store_vec<VW>(rstd_out + (long)b * nch + c, rstd);
        return;
    }
    for (int l = rg; l < npts; l += RG) {
        float v[VW];
        load_vec<VW>(xb + (long)l * ldx, v);
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[e] += cv ? (double)v[e] : 0.0;
    }
    group_sum<VW>(acc, s, cg, rg);
    double mean[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { mean[e] = acc[e] / npts; acc[e] = 0.0; }
    for (int l = rg; l < npts; l += RG) {
        float v[VW];
        load_vec<VW>(xb + (long)l * ldx, v);
#pragma unroll
        for (int e = 0; e < VW; ++e) { const double d = cv ? (double)v[e] - mean[e] : 0.0; acc[e] += d * d; }
    }
    group_sum<VW>(acc, s, cg, rg);
    float rstd[VW], meanf[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        rstd[e] = (float)(1.0 / sqrt(acc[e] / npts + (double)eps));          // biased variance, as F.instance_norm
        meanf[e] = (float)mean[e];
    }
    if (!cv) return;
    float* yb = y + ((long)b * npts) * nch + c;
    for (int l = rg; l < npts; l += RG) {
        float v[VW];
        load_vec<VW>(xb + (long)l * ldx, v);
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] = (v[e] - meanf[e]) * rstd[e];
        store_vec<VW>(yb + (long)l * nch, v);
    }
    if (rg == 0) store_vec<VW>(rstd_out + (long)b * nch + c, rstd);
}

// d_x = rstd * (d_y - mean_l(d_y) - y * mean_l(d_y * y)),  y = the normalised output
template <int VW, bool RES = false>
__global__ __launch_bounds__(256) void instance_norm_bwd_kernel(const float* __restrict__ d_y, const float* __restrict__ y,
                                                                 const float* __restrict__ rstd, int npts, int nch,
                                                                 float* __restrict__ d_x) {
    constexpr int CG = NormShape<VW>::CG, RG = NormShape<VW>::RG;
    __shared__ double s[256 * VW];
    const int cg = threadIdx.x % CG, rg = threadIdx.x / CG;
    const int c = blockIdx.x * 64 + cg * VW, b = blockIdx.y;
    const bool cv = c < nch;
    const long base = ((long)b * npts) * nch + (cv ? c : 0);
    double s1[VW], s2[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
    if (RES) {                                                 // (see instance_norm_fwd_kernel)
        float gv[RES_ROWS][VW], yv[RES_ROWS][VW];
#pragma unroll
        for (int i = 0; i < RES_ROWS; ++i) {
            const int l = rg + i * RG;
            if (l < npts) {
                load_vec<VW>(d_y + base + (long)l * nch, gv[i]);
                load_vec<VW>(y + base + (long)l * nch, yv[i]);
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) { gv[i][e] = 0.0f; yv[i][e] = 0.0f; }
            }
        }
#pragma unroll
        for (int i = 0; i < RES_ROWS; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const double gd = (cv && rg + i * RG < npts) ? (double)gv[i][e] : 0.0;
                s1[e] += gd;
                s2[e] += gd * (double)yv[i][e];
            }
        group_sum<VW>(s1, s, cg, rg);
        group_sum<VW>(s2, s, cg, rg);
        if (!cv) return;
        float r[VW], m1[VW], m2[VW];
        load_vec<VW>(rstd + (long)b * nch + c, r);
#pragma unroll
        for (int e = 0; e < VW; ++e) { m1[e] = (float)(s1[e] / npts); m2[e] = (float)(s2[e] / npts); }
#pragma unroll
        for (int i = 0; i < RES_ROWS; ++i) {
            const int l = rg + i * RG;
            if (l >= npts) break;
            float g[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) g[e] = r[e] * (gv[i][e] - m1[e] - yv[i][e] * m2[e]);
            store_vec<VW>(d_x + base + (long)l * nch, g);
        }
        return;
    }
    for (int l = rg; l < npts; l += RG) {
        float g[VW], yv[VW];
        load_vec<VW>(d_y + base + (long)l * nch, g);
        load_vec<VW>(y + base + (long)l * nch, yv);
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const double gd = cv ? (double)g[e] : 0.0;
            s1[e] += gd;
            s2[e] += gd * (double)yv[e];
        }
    }
    group_sum<VW>(s1, s, cg, rg);
    group_sum<VW>(s2, s, cg, rg);
    if (!cv) return;
    float r[VW], m1[VW], m2[VW];
    load_vec<VW>(rstd + (long)b * nch + c, r);
#pragma unroll
    for (int e = 0; e < VW; ++e) { m1[e] = (float)(s1[e] / npts); m2[e] = (float)(s2[e] / npts); }
    for (int l = rg; l < npts; l += RG) {
        float g[VW], yv[VW];
        load_vec<VW>(d_y + base + (long)l * nch, g);
        load_vec<VW>(y + base + (long)l * nch, yv);
#pragma unroll
        for (int e = 0; e < VW; ++e) g[e] = r[e] * (g[e] - m1[e] - yv[e] * m2[e]);
        store_vec<VW>(d_x + base + (long)l * nch, g);
    }
}

bool vec4_ok(const void* p, long row_stride, long batch_stride, int nch) {
    return nch % 4 == 0 && row_stride % 4 == 0 && batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

}  // namespace

extern "C" int pit_instance_norm_fwd(const float* x, long ldx, long x_bstride, int batch, int npts, int nch, float eps,
                                     float* y, float* rstd, void* stream) {
    if (!x || !y || !rstd) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || ldx < nch || batch > 65535) return PIT_ERR_SIZE;
    const dim3 grid((nch + 63) / 64, batch);
    if (vec4_ok(x, ldx, x_bstride, nch) && vec4_ok(y, nch, 4, nch) && vec4_ok(rstd, 4, 4, nch)) {
        if (npts <= RES_ROWS * NormShape<4>::RG)
            hipLaunchKernelGGL((instance_norm_fwd_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, x_bstride,
                               npts, nch, eps, y, rstd);
        else
            hipLaunchKernelGGL(instance_norm_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, x_bstride, npts,
                               nch, eps, y, rstd);
    } else
        hipLaunchKernelGGL(instance_norm_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, x_bstride, npts,
                           nch, eps, y, rstd);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_instance_norm_bwd(const float* d_y, const float* y, const float* rstd, int batch, int npts, int nch,
                                     float* d_x, void* stream) {
    if (!d_y || !y || !rstd || !d_x) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || batch > 65535) return PIT_ERR_SIZE;
    const dim3 grid((nch + 63) / 64, batch);
    if (vec4_ok(d_y, nch, 4, nch) && vec4_ok(y, nch, 4, nch) && vec4_ok(d_x, nch, 4, nch) && vec4_ok(rstd, 4, 4, nch)) {
        if (npts <= RES_ROWS * NormShape<4>::RG)
            hipLaunchKernelGGL((instance_norm_bwd_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, d_y, y, rstd, npts,
                               nch, d_x);
        else
            hipLaunchKernelGGL(instance_norm_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, d_y, y, rstd, npts, nch, d_x);
    } else
        hipLaunchKernelGGL(instance_norm_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, d_y, y, rstd, npts, nch, d_x);
    PIT_CHECK_LAUNCH();
    return 0;
}
