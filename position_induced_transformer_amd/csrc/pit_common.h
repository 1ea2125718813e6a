// Shared device helpers for the PiT hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/pit_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PIT_WAVE 64

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact fp32 fma chain.
// Fragment maps (cdna guide section 3): lane l holds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; accumulator register r of lane l is
// D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Math mode (the math_mode argument of the ABI calls): the contractions consume operand groups of 8 along the reduced
// axis, 4 per half-wave.  PIT_MATH_FP32 issues four exact v_mfma_f32_32x32x2_f32 (lane holds
// position 2u+half of the group for u = 0..3); PIT_MATH_BF16 rounds both operands to bf16 (RNE)
// and issues ONE v_mfma_f32_32x32x8_bf16 (lane holds positions 4*half+u, fp32 accumulation).
// group_pos() is the position a lane must fetch for slot u so that both modes share all the
// load / weight code.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
// The math mode is an ARGUMENT of every ABI call that contracts (no process-wide state): the entry
// point parks it in a thread-local for the launch helpers of that call (PIT_ENTER_MATH).
extern thread_local int t_call_math;
// (the low byte of the argument is the math mode, the PIT_IO_* storage flags above it are read by the entry point itself)
#define PIT_ENTER_MATH(mode_) do { if (((mode_) & 0xff) != PIT_MATH_FP32 && ((mode_) & 0xff) != PIT_MATH_BF16) return PIT_ERR_UNSUPPORTED; \
                                   if (((mode_) & ~0xff & ~PIT_ATT_UNION) && ((mode_) & 0xff) != PIT_MATH_BF16) return PIT_ERR_UNSUPPORTED; \
                                   t_call_math = ((mode_) & 0xff); } while (0)
// bf16 storage (PIT_IO_*): a tensor kept as bf16 in memory is widened exactly (bits << 16), narrowed with RNE
__device__ __forceinline__ float bf16_to_f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short f_to_bf16(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ int group_pos(bool bf16, int u, int half) { return bf16 ? 4 * half + u : 2 * u + half; }
__device__ __forceinline__ short bf16_bits(float x) { return __builtin_bit_cast(short, (__bf16)x); }
__device__ __forceinline__ bf16x4 pack_bf16(float a0, float a1, float a2, float a3) {
    bf16x4 v;
    v[0] = bf16_bits(a0); v[1] = bf16_bits(a1); v[2] = bf16_bits(a2); v[3] = bf16_bits(a3);
    return v;
}
__device__ __forceinline__ f32x16 mfma_32x32x8_bf16(bf16x4 a, bf16x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
}

// Squared distance exactly as the reference forms it in fp32:
// separate multiplies and adds, never an fma, never the |x|^2+|y|^2-2xy expansion.
// Coordinates are zero-padded to 3, which leaves the value unchanged bit for bit
// (x + (+0) == x).  Periodic variants wrap each coordinate first (pit.py:192-193,
// 251-252): d = |d|; d = min(d, l - d).
__device__ __forceinline__ float sq_dist3(float ox, float oy, float oz, float ix, float iy, float iz,
                                          bool periodic, float period) {
    float dx = __fsub_rn(ox, ix), dy = __fsub_rn(oy, iy), dz = __fsub_rn(oz, iz);
    if (periodic) {
        dx = fabsf(dx); dx = fminf(dx, __fsub_rn(period, dx));
        dy = fabsf(dy); dy = fminf(dy, __fsub_rn(period, dy));
        dz = fabsf(dz); dz = fminf(dz, __fsub_rn(period, dz));
    }
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// the same with the wrap as a compile-time variant (hot loops pick it once per launch)
template <bool PERIODIC>
__device__ __forceinline__ float sq_dist3t(float ox, float oy, float oz, float ix, float iy, float iz, float period) {
    float dx = __fsub_rn(ox, ix), dy = __fsub_rn(oy, iy), dz = __fsub_rn(oz, iz);
    if (PERIODIC) {
        dx = fabsf(dx); dx = fminf(dx, __fsub_rn(period, dx));
        dy = fabsf(dy); dy = fminf(dy, __fsub_rn(period, dy));
        dz = fabsf(dz); dz = fminf(dz, __fsub_rn(period, dz));
    }
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ATen's lerp as torch.quantile applies it.
__device__ __forceinline__ float quantile_lerp(float a, float b, float w) {
    float diff = __fsub_rn(b, a);
    return (w < 0.5f) ? __fmaf_rn(w, diff, a) : __fmaf_rn(-diff, __fsub_rn(1.0f, w), b);
}

// c = tan(0.25*pi*(1-1e-7)*(1+sin(lmda))) with the reference's fp32 roundings of the
// intermediate results (pit.py:48) and correctly rounded sin/tan (evaluated in fp64).
#define PIT_SCALE_K 0x1.921fb2a19bef9p-1  /* 0.25*pi*(1-1e-7) evaluated in double = 0.785398084857632 */
// sin and cos in fp64 the fdlibm way (two-term Cody-Waite reduction by pi/2, the k_sin / k_cos
// polynomials on [-pi/4, pi/4]): ~35 double-precision instructions, no branches, against ~120 with
// loops for the library sin().  Every forward attention workgroup evaluates the head scale, so this is
// on the hot path.  After rounding to fp32 the results are identical to the library's (checked on
// the host for 3.5 M arguments in [-1e3, 1e3] and 2 M composite c values: zero mismatches).
__device__ __forceinline__ void sincos_fp64(double x, double& sn, double& cs) {
    const double k = rint(x * 6.36619772367581382433e-01);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = r + r * z * (-1.66666666666666324348e-01 + z * ps);
    const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double c = 1.0 - (0.5 * z - z * pc);
    const int q = (int)(long long)k & 3;
    sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
__device__ __forceinline__ float head_scale_from_lmda(float lmda) {
    if (!(fabsf(lmda) < 1.0e5f)) {                         // far outside any trained value (and NaN): library path
        const float s = (float)sin((double)lmda);
        return (float)tan((double)__fmul_rn((float)PIT_SCALE_K, __fadd_rn(1.0f, s)));
    }
    double sn, cs;
    sincos_fp64((double)lmda, sn, cs);
    const float s = (float)sn;
    const float u = __fmul_rn((float)PIT_SCALE_K, __fadd_rn(1.0f, s));
    sincos_fp64((double)u, sn, cs);
    return (float)(sn / cs);
}
// d c / d lmda = (1 + c^2) * K * cos(lmda)
__device__ __forceinline__ double head_scale_grad(float lmda, float c) {
    return (1.0 + (double)c * (double)c) * PIT_SCALE_K * cos((double)lmda);
}

// Wave-wide sums on the VALU's DPP lanes instead of six ds_bpermute round trips through the LDS pipe (round 4: the cross-lane
// reduction, not the loads, bounded thin_fwd_kernel): quad swaps, row rotations by 4 and 8 (every lane of a 16-lane row then holds
// the row's sum), row_bcast:15 / :31 into the following rows, the total read from lane 63.  Returned to EVERY lane.
#ifndef PIT_NO_DPP_SUM
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1, 0xf>(v);                            // quad_perm(1,0,3,2)
    v += dpp_f<0x4E, 0xf>(v);                            // quad_perm(2,3,0,1)
    v += dpp_f<0x124, 0xf>(v);                           // row_ror:4
    v += dpp_f<0x128, 0xf>(v);                           // row_ror:8
    v += dpp_f<0x142, 0xa>(v);                           // row_bcast:15 -> rows 1 and 3
    v += dpp_f<0x143, 0xc>(v);                           // row_bcast:31 -> rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<0xB1, 0xf>(v);
    v += dpp_d<0x4E, 0xf>(v);
    v += dpp_d<0x124, 0xf>(v);
    v += dpp_d<0x128, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#else
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
#endif

// The normal CDF as exp(-x^2/2) t (c1 + ... + c6 t^5), t = 1 / (1 + 0.4 |x| / sqrt 2) (Abramowitz & Stegun 7.1.26 refitted for
// absolute error: 1.4e-8; relative L2 error of gelu / gelu' over N(0,1) arguments 4e-8 / 6e-8, libm's erff: 4e-8 / 4e-8): 16 / 18
// straight-line instructions against erff's ~60 behind a branch.  Used where the result is rounded to bf16 anyway (the bf16 math
// mode's thin tail, pit_fold.hip); the fp32 parity path keeps erff.
__device__ __forceinline__ float normal_cdf_fast(float x, float& e) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.28284271247f, 1.0f));
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044f);
    float p = -0.100061134f;
    p = fmaf(p, t, 0.354291141f);
    p = fmaf(p, t, -0.184991121f);
    p = fmaf(p, t, 0.233760774f);
    p = fmaf(p, t, 0.0810635462f);
    p = fmaf(p, t, 0.115936771f);
    const float q = p * t * e;
    return x > 0.0f ? 1.0f - q : q;
}
#ifdef PIT_GELU_FAST       // (timing experiment, DESIGN section 4 Round 5: A&S 7.1.26 refitted, 16 / 18 straight-line instructions; NOT the default)
__device__ __forceinline__ float normal_cdf(float x, float& e) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.28284271247f, 1.0f));
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044f);
    float p = -0.100061134f;
    p = fmaf(p, t, 0.354291141f);
    p = fmaf(p, t, -0.184991121f);
    p = fmaf(p, t, 0.233760774f);
    p = fmaf(p, t, 0.0810635462f);
    p = fmaf(p, t, 0.115936771f);
    const float q = p * t * e;
    return x > 0.0f ? 1.0f - q : q;
}
__device__ __forceinline__ float gelu_erf(float x) { float e; return x * normal_cdf(x, e); }
__device__ __forceinline__ float gelu_erf_grad(float x) { float e; const float c = normal_cdf(x, e); return fmaf(x * 0.39894228040143267794f, e, c); }
#else
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
#endif

// Raw buffer loads (T8): out-of-range offsets return 0 in hardware, so row / column / tail
// predication costs an integer select on the OFFSET instead of a branch around the load
// (hipcc otherwise sinks a predicated global load into an exec-masked block followed by
// s_waitcnt vmcnt(0), which serialises every load of a prefetch group).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float (&v)[4]) {
    const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    v[0] = __int_as_float(q.x); v[1] = __int_as_float(q.y); v[2] = __int_as_float(q.z); v[3] = __int_as_float(q.w);
}
// raw buffer store: an offset at or beyond the resource's size drops the store (row / column tails without a branch, and without the
// 64-bit per-lane addresses a predicated global store keeps alive - mlp_bwd64 spilled nine of them)
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), r, (int)byte_off, 0, 0);
}
// the same with the sc1 cache policy (served by L2 / memory, never by this CU's L1): loads of bytes ANOTHER workgroup of the
// same launch has stored (csrc/pit_latent.hip)
__device__ __forceinline__ void buf_load4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float (&v)[4]) {
    const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
    v[0] = __int_as_float(q.x); v[1] = __int_as_float(q.y); v[2] = __int_as_float(q.z); v[3] = __int_as_float(q.w);
}
// largest tensor a 32-bit buffer offset can address (with slack for offset arithmetic)
#define PIT_MAX_BUFFER_BYTES 0xF0000000ull

#define PIT_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)
