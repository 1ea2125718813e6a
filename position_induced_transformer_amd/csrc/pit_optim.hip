// Fused Adam step over the flat parameter / gradient buffers with the cosine-annealed learning
// rate of the reference loops (torch.optim.Adam(lr=1e-3) + CosineAnnealingLR(T_max=iterations),
// train_darcy.py:115-116,131-134), sync-free: the step counter lives on the device so the launch
// can be replayed from a hipGraph.
#include "pit_common.h"

namespace {

// One launch: every thread derives the step's learning rate and bias corrections from the device
// step counter (read before any workgroup can have advanced it: the LAST workgroup to finish -
// arrival ticket in scalars[3] - writes the new count and clears the ticket), updates its
// elements and, with zero_grads, clears the gradient it consumed (the next step's backward
// accumulates into zeros without a separate memset).
// scalars[0..2] = lr_t, 1 - beta1^t, 1 - beta2^t of the step just taken (for inspection); [3] = ticket
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, long long* step, float lr0,
                                                        float eta_min, int t_max, float beta1, float beta2, float eps,
                                                        float weight_decay, int zero_grads, float* scalars) {
    __shared__ float s_sc[3];
    const long long t = *step + 1;
    if (threadIdx.x == 0) {                    // fp64 cos / pow once per workgroup, not per thread
        // CosineAnnealingLR closed form: the rate used by the t-th optimizer.step() is that of epoch t-1
        float lr_t = lr0;
        if (t_max > 0) {
            const double ph = 3.14159265358979323846 * (double)(t - 1) / (double)t_max;
            lr_t = (float)((double)eta_min + 0.5 * ((double)lr0 - (double)eta_min) * (1.0 + cos(ph)));
        }
        s_sc[0] = lr_t;
        s_sc[1] = (float)(1.0 - pow((double)beta1, (double)t));
        s_sc[2] = (float)(1.0 - pow((double)beta2, (double)t));
    }
    __syncthreads();
    const float lr = s_sc[0], bc1 = s_sc[1], bc2 = s_sc[2];
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (zero_grads) g[i] = 0.0f;
        if (weight_decay != 0.0f) gi += weight_decay * p[i];
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;       // torch: (sqrt(v)/sqrt(bc2)) + eps
        p[i] -= step_size * (mi / denom);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(scalars + 3);
        if (atomicAdd(ticket, 1u) == gridDim.x - 1u) {            // every workgroup has read *step by now
            *step = t;
            scalars[0] = lr; scalars[1] = bc1; scalars[2] = bc2;
            atomicExch(ticket, 0u);
        }
    }
}

}  // namespace

extern "C" int pit_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                             long long* step, float lr0, float eta_min, int cosine_t_max, float beta1, float beta2,
                             float eps, float weight_decay, int zero_grads, float* scalars, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !scalars) return PIT_ERR_NULL;
    if (n <= 0) return PIT_ERR_SIZE;
    const int blocks = (int)std::min<long>((n + 255) / 256, 2048L);
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, step, lr0, eta_min, cosine_t_max, beta1, beta2, eps, weight_decay, zero_grads,
                       scalars);
    PIT_CHECK_LAUNCH();
    return 0;
}
