// Relative Lp loss on device (RelLpNorm, utils.py:80-98) with the optional per-pixel
// affine de-normalisation the Darcy loop applies first (train_darcy.py:129,
// utils.py:25-34):   pred' = pred*scale + shift;
//   loss = sum_b mean_c ||true - pred'||_p / ||true||_p      (norms over the point axis)
// One workgroup per (sample, channel) owns both norms, so a single launch produces the
// loss (one atomicAdd per workgroup) and the two norms the backward needs.
#include "pit_common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ float pow_abs(float x, int p) {
    const float ax = fabsf(x);
    if (p == 1) return ax;
    if (p == 2) return ax * ax;
    return powf(ax, (float)p);
}

// Workspace layout (floats; ZERO before the first call, left zero by every call):
//   [0] loss accumulator, [1] arrival counter over the (sample, channel) pairs,
//   then per pair 4 floats: {num, den} as one fp64 each is too wide for the float view - so per pair
//   2 doubles (num, den) at double index 1 + 2*pair... see REL_WS_* below - and one arrival counter.
// One (sample, channel) series is split over gridDim.z workgroups (one dependent-latency round trip each
// instead of npts/256 in a row); partial sums meet in the pair's fp64 slots and the LAST workgroup to
// arrive finishes the pair: norms, the pair's term of the loss and - when asked for - the gradients of
// the whole series for d loss = 1 (it alone knows both norms; 4 points per thread in flight).
__device__ __forceinline__ double* rel_ws_sums(float* ws, int pair) { return reinterpret_cast<double*>(ws + 4) + 2 * pair; }
__device__ __forceinline__ unsigned* rel_ws_count(float* ws, int pairs, int pair) {
    return reinterpret_cast<unsigned*>(ws + 4 + 4 * pairs) + pair;
}

__global__ __launch_bounds__(256) void rel_lp_fwd_kernel(const float* __restrict__ tru, const float* __restrict__ pred,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int npts, int nch, int p, float* __restrict__ norms,
                                                          float* __restrict__ loss, float* __restrict__ ws,
                                                          float* __restrict__ d_pred_unit, float* __restrict__ d_true_unit,
                                                          float* __restrict__ clear_buf, long clear_n) {
    __shared__ double s_num[4], s_den[4];
    __shared__ float s_norm[3];                        // {||t - q||, ||t||, this pair's term of the loss}
    __shared__ int s_last;
    if (clear_buf) {                                  // fused memset of a caller buffer (gradient accumulators)
        const long nthreads = (long)gridDim.x * gridDim.y * gridDim.z * blockDim.x;
        const long first = (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        for (long i = first; i < clear_n; i += nthreads) clear_buf[i] = 0.0f;
    }
    const int c = blockIdx.x, b = blockIdx.y, parts = gridDim.z;
    const int pairs = gridDim.x * gridDim.y, pair = b * nch + c;
    const long base = (long)b * npts * nch + c;
    const int chunk = (npts + parts - 1) / parts;
    const int lbeg = blockIdx.z * chunk, lend = min(npts, lbeg + chunk);
    double num = 0.0, den = 0.0;
    for (int l0 = lbeg + threadIdx.x; l0 < lend; l0 += 4 * blockDim.x) {      // 4 points per thread in flight
        float qv[4], tv[4], sc[4], sh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * blockDim.x;
            const bool ok = l < lend;
            const long e = base + (long)(ok ? l : lbeg) * nch;
            qv[u] = ok ? pred[e] : 0.0f;
            tv[u] = ok ? tru[e] : 0.0f;
            sc[u] = (scale && ok) ? scale[(long)l * nch + c] : 1.0f;
            sh[u] = (scale && ok) ? shift[(long)l * nch + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float q = scale ? qv[u] * sc[u] + sh[u] : qv[u];
            num += (double)pow_abs(tv[u] - q, p);
            den += (double)pow_abs(tv[u], p);
        }
    }
    num = wave_sum_d(num);
    den = wave_sum_d(den);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_num[wave] = num; s_den[wave] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        num = s_num[0] + s_num[1] + s_num[2] + s_num[3];
        den = s_den[0] + s_den[1] + s_den[2] + s_den[3];
        double* sums = rel_ws_sums(ws, pair);
        unsigned* cnt = rel_ws_count(ws, pairs, pair);
        int last = 1;
        if (parts > 1) {
            const double o1 = atomicAdd(sums, num), o2 = atomicAdd(sums + 1, den);
            asm volatile("" ::"v"(o1), "v"(o2));      // returning atomics: performed before the ticket below
            last = (atomicAdd(cnt, 1u) == (unsigned)parts - 1u);
            if (last) {                                // everybody's partial sums are in: take them, leave zeros
                num = __longlong_as_double(atomicExch(reinterpret_cast<unsigned long long*>(sums), 0ull));
                den = __longlong_as_double(atomicExch(reinterpret_cast<unsigned long long*>(sums + 1), 0ull));
                atomicExch(cnt, 0u);
            }
        }
        s_last = last;
        if (last) {
            const double nn = (p == 1) ? num : (p == 2 ? sqrt(num) : pow(num, 1.0 / p));
            const double dn = (p == 1) ? den : (p == 2 ? sqrt(den) : pow(den, 1.0 / p));
            norms[(long)pair * 2 + 0] = (float)nn;
            norms[(long)pair * 2 + 1] = (float)dn;
            s_norm[0] = (float)nn; s_norm[1] = (float)dn; s_norm[2] = (float)(nn / dn / nch);
        }
    }
    __syncthreads();
    if (!s_last) return;
    const float nn = s_norm[0], dn = s_norm[1];
    if (threadIdx.x == 0) {
        // sum over (sample, channel) in a persistent accumulator; the last pair publishes the loss and
        // leaves accumulator and arrival counter zero for the next call (no memset launch).  After the barrier:
        // these two dependent L2 round trips run beside the gradient pass of the other 255 threads
        const float old = atomicAdd(ws, s_norm[2]);
        asm volatile("" ::"v"(old));
        unsigned* counter = reinterpret_cast<unsigned*>(ws + 1);
        const unsigned ticket = atomicAdd(counter, 1u);
        if (ticket == (unsigned)pairs - 1u) {
            *loss = atomicExch(ws, 0.0f);
            atomicExch(counter, 0u);
        }
    }
    if (!d_pred_unit && !d_true_unit) return;
    // gradients for an upstream gradient of 1, by the workgroup that completed the pair (it holds both
    // norms): the training step then needs no separate backward launch for the loss
    for (int l0 = threadIdx.x; l0 < npts; l0 += 4 * blockDim.x) {
        float qv[4], tv[4], sc[4], sh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * blockDim.x;
            const bool ok = l < npts;
            const long e = base + (long)(ok ? l : 0) * nch;
            qv[u] = ok ? pred[e] : 0.0f;
            tv[u] = ok ? tru[e] : 0.0f;
            sc[u] = (scale && ok) ? scale[(long)l * nch + c] : 1.0f;
            sh[u] = (scale && ok) ? shift[(long)l * nch + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * blockDim.x;
            if (l >= npts) continue;
            const long e = base + (long)l * nch;
            const float q = scale ? qv[u] * sc[u] + sh[u] : qv[u];
            const float t = tv[u];
            const float d = q - t;
            float dnorm;
            if (p == 1) dnorm = (d > 0.0f) ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            else if (p == 2) dnorm = (nn > 0.0f) ? d / nn : 0.0f;
            else dnorm = (nn > 0.0f) ? copysignf(powf(fabsf(d) / nn, (float)(p - 1)), d) : 0.0f;
            if (d_pred_unit) d_pred_unit[e] = dnorm * sc[u] / (dn * nch);
            if (d_true_unit) {
                float tnorm;
                if (p == 1) tnorm = (t > 0.0f) ? 1.0f : (t < 0.0f ? -1.0f : 0.0f);
                else if (p == 2) tnorm = (dn > 0.0f) ? t / dn : 0.0f;
                else tnorm = (dn > 0.0f) ? copysignf(powf(fabsf(t) / dn, (float)(p - 1)), t) : 0.0f;
                d_true_unit[e] = (-dnorm / dn - nn / (dn * dn) * tnorm) / nch;
            }
        }
    }
}

__global__ __launch_bounds__(256) void rel_lp_bwd_kernel(const float* __restrict__ tru, const float* __restrict__ pred,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int batch, int npts, int nch, int p,
                                                          const float* __restrict__ norms, const float* __restrict__ gloss,
                                                          float* __restrict__ d_pred, float* __restrict__ d_true) {
    const long total = (long)batch * npts * nch;
    const float g = gloss ? gloss[0] : 1.0f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % nch);
        const long bl = e / nch;
        const int l = (int)(bl % npts);
        const int b = (int)(bl / npts);
        float q = pred[e];
        float sc = 1.0f;
        if (scale) { sc = scale[(long)l * nch + c]; q = q * sc + shift[(long)l * nch + c]; }
        const float d = q - tru[e];
        const float nn = norms[((long)b * nch + c) * 2 + 0];
        const float dn = norms[((long)b * nch + c) * 2 + 1];
        float dnorm;                                    // d ||d||_p / d d
        if (p == 1) dnorm = (d > 0.0f) ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        else if (p == 2) dnorm = (nn > 0.0f) ? d / nn : 0.0f;
        else dnorm = (nn > 0.0f) ? copysignf(powf(fabsf(d) / nn, (float)(p - 1)), d) : 0.0f;
        if (d_pred) d_pred[e] = g * dnorm * sc / (dn * nch);
        if (d_true) {
            // d/dt [ ||t - q|| / ||t|| ] = -dnorm/||t|| - ||t-q||/||t||^2 * d||t||/dt
            const float t = tru[e];
            float tnorm;
            if (p == 1) tnorm = (t > 0.0f) ? 1.0f : (t < 0.0f ? -1.0f : 0.0f);
            else if (p == 2) tnorm = (dn > 0.0f) ? t / dn : 0.0f;
            else tnorm = (dn > 0.0f) ? copysignf(powf(fabsf(t) / dn, (float)(p - 1)), t) : 0.0f;
            d_true[e] = g * (-dnorm / dn - nn / (dn * dn) * tnorm) / nch;
        }
    }
}

// RelMaxNorm (utils.py:59-77): sum_b mean_c max_l |true - pred| / max_l |true|.  One workgroup per
// (sample, channel); max is exact in any order, so this equals the reference bit for bit up to the
// final sum over (sample, channel), which is accumulated in fp64 and published by the last arriver.
__global__ __launch_bounds__(256) void rel_max_fwd_kernel(const float* __restrict__ tru, const float* __restrict__ pred,
                                                           int npts, int nch, float* __restrict__ out,
                                                           double* __restrict__ ws) {
    __shared__ float s_num[4], s_den[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const long base = (long)b * npts * nch + c;
    float num = 0.0f, den = 0.0f;
    bool bad = false;                                  // torch.max propagates NaN
    for (int l = threadIdx.x; l < npts; l += blockDim.x) {
        const long e = base + (long)l * nch;
        const float t = tru[e], d = fabsf(t - pred[e]);
        bad |= (d != d) || (t != t);
        num = fmaxf(num, d);
        den = fmaxf(den, fabsf(t));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        num = fmaxf(num, __shfl_xor(num, o, 64));
        den = fmaxf(den, __shfl_xor(den, o, 64));
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_num[wave] = num; s_den[wave] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        num = fmaxf(fmaxf(s_num[0], s_num[1]), fmaxf(s_num[2], s_num[3]));
        den = fmaxf(fmaxf(s_den[0], s_den[1]), fmaxf(s_den[2], s_den[3]));
        float ratio = num / den;                       // fp32 division as in the reference (0/0 -> NaN, x/0 -> inf)
        if (any_bad) ratio = __builtin_nanf("");
        const double old = atomicAdd(ws, (double)ratio / nch);
        asm volatile("" ::"v"(old));
        unsigned long long* counter = reinterpret_cast<unsigned long long*>(ws + 1);
        const unsigned long long ticket = atomicAdd(counter, 1ull);
        if (ticket == (unsigned long long)gridDim.x * gridDim.y - 1ull) {
            *out = (float)__longlong_as_double(atomicExch(reinterpret_cast<unsigned long long*>(ws), 0ull));
            atomicExch(counter, 0ull);
        }
    }
}

// Round 4: ONE 1024-thread workgroup per (sample, channel) series when the series fits its registers (npts <= 1024 PTS, PTS <= 16:
// the NACA field of 11 271 points takes 12 - 31 -> 27 us against the split form, whose last arriver writes a series' gradients alone).
// rel_lp_fwd_kernel's chain for a Darcy-sized series (1849 points split over 8 workgroups) is: loads -> two returning fp64
// atomics -> a ticket -> three exchanges by the last arriver -> the gradient pass RE-READING the whole series: ~10.6 us of
// dependent round trips for 15 k values.  Here the series stays in registers: loads (all in flight) -> block reduction ->
// norms -> the gradients straight from the registers; the loss accumulates in an fp64 word, the arrivals in a ticket word of
// their own (round 5; non-finite terms propagate to *loss and the workspace is always left zero).
// PK: the exponent at compile time (1, 2; 0 = the run-time p): with a run-time p the generic powf path is inlined for every value of the
// series in both passes - 12.7 k instructions at PTS = 12, more than the instruction cache holds: the NACA loss took 26.7 us
template <int PTS, int PK>
__global__ __launch_bounds__(1024) void rel_lp_fwd1_kernel(const float* __restrict__ tru, const float* __restrict__ pred,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int npts, int nch, int p_arg, float* __restrict__ norms,
                                                           float* __restrict__ loss, float* __restrict__ ws,
                                                           float* __restrict__ d_pred_unit, float* __restrict__ d_true_unit,
                                                           float* __restrict__ clear_buf, long clear_n) {
    __shared__ double s_num[16], s_den[16];
    const int p = PK ? PK : p_arg;
    const int tid = threadIdx.x;
    if (clear_buf) {
        const long nthreads = (long)gridDim.x * gridDim.y * blockDim.x;
        const long first = ((long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + tid;
        for (long i = first; i < clear_n; i += nthreads) clear_buf[i] = 0.0f;
    }
    const int c = blockIdx.x, b = blockIdx.y;
    const int pairs = gridDim.x * gridDim.y, pair = b * nch + c;
    const long base = (long)b * npts * nch + c;
    float qv[PTS], tv[PTS], sc[PTS], sh[PTS];
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
        const int l = tid + u * 1024;
        const bool ok = l < npts;
        const long e = base + (long)(ok ? l : 0) * nch;
        qv[u] = ok ? pred[e] : 0.0f;
        tv[u] = ok ? tru[e] : 0.0f;
        sc[u] = (scale && ok) ? scale[(long)l * nch + c] : 1.0f;
        sh[u] = (scale && ok) ? shift[(long)l * nch + c] : 0.0f;
    }
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
        if (scale) qv[u] = qv[u] * sc[u] + sh[u];
        if (tid + u * 1024 < npts) {
            num += (double)pow_abs(tv[u] - qv[u], p);
            den += (double)pow_abs(tv[u], p);
        }
    }
    num = wave_sum_d(num);
    den = wave_sum_d(den);
    if ((tid & 63) == 0) { s_num[tid >> 6] = num; s_den[tid >> 6] = den; }
    __syncthreads();
    num = 0.0; den = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { num += s_num[w]; den += s_den[w]; }          // (every thread: same order, same value)
    const double nnd = (p == 1) ? num : (p == 2 ? sqrt(num) : pow(num, 1.0 / p));
    const double dnd = (p == 1) ? den : (p == 2 ? sqrt(den) : pow(den, 1.0 / p));
    const float nn = (float)nnd, dn = (float)dnd;
    if (tid == 0) {
        norms[(long)pair * 2 + 0] = nn;
        norms[(long)pair * 2 + 1] = dn;
        // the pair's term joins an fp64 accumulator, THEN the arrival ticket (its own word): a non-finite term (0/0 from an all-zero
        // target series, a diverged prediction) reaches *loss as nan / inf exactly as the reference's LpLoss reports it, and can
        // never disturb the count - a packed count|fixed-point word (round 4) turned inf into a finite wrong loss and, once the
        // borrow hit the count, left the accumulator armed for every later call
        const float term = (float)(nnd / dnd / nch);
        double* acc = reinterpret_cast<double*>(ws + 2);
        const double old = atomicAdd(acc, (double)term);
        asm volatile("" ::"v"(old));                   // returning atomic: performed before the ticket below
        unsigned* counter = reinterpret_cast<unsigned*>(ws + 1);
        if (atomicAdd(counter, 1u) == (unsigned)pairs - 1u) {
            *loss = (float)__longlong_as_double(atomicExch(reinterpret_cast<unsigned long long*>(acc), 0ull));
            atomicExch(counter, 0u);
        }
    }
    if (!d_pred_unit && !d_true_unit) return;
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
        const int l = tid + u * 1024;
        if (l >= npts) continue;
        const long e = base + (long)l * nch;
        const float t = tv[u];
        const float d = qv[u] - t;
        float dnorm;
        if (p == 1) dnorm = (d > 0.0f) ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        else if (p == 2) dnorm = (nn > 0.0f) ? d / nn : 0.0f;
        else dnorm = (nn > 0.0f) ? copysignf(powf(fabsf(d) / nn, (float)(p - 1)), d) : 0.0f;
        if (d_pred_unit) d_pred_unit[e] = dnorm * sc[u] / (dn * nch);
        if (d_true_unit) {
            float tnorm;
            if (p == 1) tnorm = (t > 0.0f) ? 1.0f : (t < 0.0f ? -1.0f : 0.0f);
            else if (p == 2) tnorm = (dn > 0.0f) ? t / dn : 0.0f;
            else tnorm = (dn > 0.0f) ? copysignf(powf(fabsf(t) / dn, (float)(p - 1)), t) : 0.0f;
            d_true_unit[e] = (-dnorm / dn - nn / (dn * dn) * tnorm) / nch;
        }
    }
}

// the single-workgroup form when the series fits (and the grid is not huge); false = the split form
bool launch_rel_lp_fwd1(const float* tru, const float* pred, const float* scale, const float* shift, int batch, int npts, int nch,
                        int p, float* norms, float* loss, float* ws, float* d_pred_unit, float* d_true_unit, float* clear_buf,
                        long clear_n, hipStream_t s) {
    static const bool off = getenv("PIT_NO_LOSS1") != nullptr;
    if (off || npts > 16384 || (long)batch * nch > 4096) return false;          // (series of up to 16 values per thread: 64 registers)
    const dim3 grid(nch, batch), block(1024);
    const int pts = (npts + 1023) / 1024;
#define PIT_L1K(P_, K_) hipLaunchKernelGGL((rel_lp_fwd1_kernel<P_, K_>), grid, block, 0, s, tru, pred, scale, shift, npts, nch, p, norms, loss, ws, \
                                           d_pred_unit, d_true_unit, clear_buf, clear_n)
#define PIT_L1(P_) do { if (p == 2) PIT_L1K(P_, 2); else if (p == 1) PIT_L1K(P_, 1); else PIT_L1K(P_, 0); } while (0)
    if (pts <= 1) PIT_L1(1); else if (pts <= 2) PIT_L1(2); else if (pts <= 4) PIT_L1(4); else if (pts <= 8) PIT_L1(8);
    else if (pts <= 12) PIT_L1(12); else PIT_L1(16);
#undef PIT_L1
#undef PIT_L1K
    return true;
}

// workgroups per (sample, channel) series: one 256-point trip each, at most 8, fewer for large batches
int rel_parts(int batch, int npts, int nch) {
    int parts = std::max(1, std::min(8, (npts + 255) / 256));
    while (parts > 1 && (long)parts * batch * nch > 2048) parts >>= 1;
    return parts;
}

}  // namespace

extern "C" int pit_rel_max_norm(const float* tru, const float* pred, int batch, int npts, int nch, float* out,
                                double* workspace, void* stream) {
    if (!tru || !pred || !out || !workspace) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || batch > 65535) return PIT_ERR_SIZE;
    hipLaunchKernelGGL(rel_max_fwd_kernel, dim3(nch, batch), dim3(256), 0, (hipStream_t)stream, tru, pred, npts, nch,
                       out, workspace);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_rel_lp_loss_fwd(const float* tru, const float* pred, const float* pred_scale,
                                   const float* pred_shift, int batch, int npts, int nch, int p,
                                   float* norms, float* loss, float* workspace, void* stream) {
    if (!tru || !pred || !norms || !loss || !workspace) return PIT_ERR_NULL;
    if ((pred_scale == nullptr) != (pred_shift == nullptr)) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || p < 1 || batch > 65535) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    if (launch_rel_lp_fwd1(tru, pred, pred_scale, pred_shift, batch, npts, nch, p, norms, loss, workspace, nullptr, nullptr, nullptr, 0L, s)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(rel_lp_fwd_kernel, dim3(nch, batch, rel_parts(batch, npts, nch)), dim3(256), 0, s, tru, pred,
                       pred_scale, pred_shift, npts, nch, p, norms, loss, workspace, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, 0L);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_rel_lp_loss_fwd_grad(const float* tru, const float* pred, const float* pred_scale,
                                        const float* pred_shift, int batch, int npts, int nch, int p,
                                        float* norms, float* loss, float* workspace, float* d_pred_unit,
                                        float* d_true_unit, float* clear_buf, long clear_n, void* stream) {
    if (!tru || !pred || !norms || !loss || !workspace) return PIT_ERR_NULL;
    if ((pred_scale == nullptr) != (pred_shift == nullptr)) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || p < 1 || batch > 65535 || clear_n < 0) return PIT_ERR_SIZE;
    if (clear_n > 0 && !clear_buf) return PIT_ERR_NULL;
    if (launch_rel_lp_fwd1(tru, pred, pred_scale, pred_shift, batch, npts, nch, p, norms, loss, workspace, d_pred_unit, d_true_unit,
                           clear_n > 0 ? clear_buf : nullptr, clear_n, (hipStream_t)stream)) {
        PIT_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(rel_lp_fwd_kernel, dim3(nch, batch, rel_parts(batch, npts, nch)), dim3(256), 0, (hipStream_t)stream,
                       tru, pred, pred_scale, pred_shift, npts, nch, p, norms, loss, workspace, d_pred_unit, d_true_unit,
                       clear_n > 0 ? clear_buf : nullptr, clear_n);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_rel_lp_loss_bwd(const float* tru, const float* pred, const float* pred_scale,
                                   const float* pred_shift, int batch, int npts, int nch, int p,
                                   const float* norms, const float* grad_loss, float* d_pred, float* d_true,
                                   void* stream) {
    if (!tru || !pred || !norms || (!d_pred && !d_true)) return PIT_ERR_NULL;
    if (batch <= 0 || npts <= 0 || nch <= 0 || p < 1) return PIT_ERR_SIZE;
    const long total = (long)batch * npts * nch;
    const int blocks = (int)std::min<long>((total + 255) / 256, 2048L);
    hipLaunchKernelGGL(rel_lp_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tru, pred, pred_scale,
                       pred_shift, batch, npts, nch, p, norms, grad_loss, d_pred, d_true);
    PIT_CHECK_LAUNCH();
    return 0;
}
