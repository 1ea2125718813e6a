// Fused Adam step over the flat parameter / gradient buffers with the cosine-annealed learning
// rate of the reference loops (torch.optim.Adam(lr=1e-3) + CosineAnnealingLR(T_max=iterations),
// train_darcy.py:115-116,131-134), sync-free: the step counter lives on the device so the launch
// pair can be replayed from a hipGraph.
#include "pit_common.h"

namespace {

// scalars[0] = lr_t, [1] = 1 - beta1^t, [2] = 1 - beta2^t
__global__ void adam_tick_kernel(long long* step, float lr0, float eta_min, int t_max, float beta1, float beta2,
                                 float* scalars) {
    const long long t = *step + 1;
    *step = t;
    // CosineAnnealingLR closed form: the rate used by the t-th optimizer.step() is that of epoch t-1
    float lr = lr0;
    if (t_max > 0) {
        const double ph = 3.14159265358979323846 * (double)(t - 1) / (double)t_max;
        lr = (float)((double)eta_min + 0.5 * ((double)lr0 - (double)eta_min) * (1.0 + cos(ph)));
    }
    scalars[0] = lr;
    scalars[1] = (float)(1.0 - pow((double)beta1, (double)t));
    scalars[2] = (float)(1.0 - pow((double)beta2, (double)t));
}

__global__ __launch_bounds__(256) void adam_update_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long n,
                                                          const float* __restrict__ scalars, float beta1, float beta2,
                                                          float eps, float weight_decay) {
    const float lr = scalars[0], bc1 = scalars[1], bc2 = scalars[2];
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (weight_decay != 0.0f) gi += weight_decay * p[i];
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;       // torch: (sqrt(v)/sqrt(bc2)) + eps
        p[i] -= step_size * (mi / denom);
    }
}

}  // namespace

extern "C" int pit_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                             long long* step, float lr0, float eta_min, int cosine_t_max, float beta1, float beta2,
                             float eps, float weight_decay, float* scalars, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !scalars) return PIT_ERR_NULL;
    if (n <= 0) return PIT_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, step, lr0, eta_min, cosine_t_max, beta1, beta2, scalars);
    PIT_CHECK_LAUNCH();
    const int blocks = (int)std::min<long>((n + 255) / 256, 2048L);
    hipLaunchKernelGGL(adam_update_kernel, dim3(blocks), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       scalars, beta1, beta2, eps, weight_decay);
    PIT_CHECK_LAUNCH();
    return 0;
}
