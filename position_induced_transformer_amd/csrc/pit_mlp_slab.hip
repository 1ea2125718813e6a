// Fused kaiming_mlp kernels for the LARGE regime at hid 64 (round 4): 64-row slabs, rows staged through LDS.
//
// pit.py:13-26 (+ the trailing gelu of pit.py:111,121) at 65 536 rows x (192 -> 64 -> 64) - the processor MLP of Darcy at
// batch 256 - ran as two streaming GEMM launches forward (27 + 27 us for 117 MB: X read, Z1 / H written, H read again, Z2 / Y
// written) and three backward (10 + 21 + 36 us for 134 MB); the persistent register-direct variant of round 3 (mlp_fwd16
// with 16-row slabs, every lane fetching its own A fragments) was bound by those row loads.  Here a 256-thread workgroup
// walks 64-row slabs:
//   forward   X slab (64 x n0, one contiguous block when ldx = n0) -> LDS by coalesced 16-B loads, requested one slab AHEAD
//             into registers; GEMM1 (wave w owns hidden columns [16w, 16w+16): its W1 rows live in registers for the whole
//             launch) + bias + erf-GELU -> Z1, H to memory and H to LDS; GEMM2 + bias (+ GELU) from LDS -> Z2, Y.
//             H is never re-read from memory, the weights are fetched once per workgroup.
//   backward  (data path) dY slab -> dZ2 = dY gelu'(Z2) -> LDS; dZ1 = (dZ2 W2) gelu'(Z1) -> LDS; dX = dZ1 W1; dZ1 / dZ2 also
//             go to the scratch the weight-gradient reductions (gemm_rr_kernel) read.
// v_mfma_f32_16x16x4_f32, exact fp32 products: the arithmetic of mlp_fwd16_kernel / mlp_bwd16_kernel (same fragment order,
// same two-accumulator split), so results are bit-identical to the small-regime kernels and equal to the GEMM path to
// summation order.  2.15 GFLOP per MLP forward = 13.7 us of the fp32 MFMA peak, 117 MB = 23 us of HBM at 5 TB/s.
#include "pit_common.h"
#include "pit_block_dev.h"
#include <cstdlib>

namespace {

struct SlabFwdArgs {
    const float* x; long ldx; int rows, n0, n2;
    const float *w1, *b1, *w2, *b2; int out_gelu;
    float *z1, *h, *z2, *y; long ldy;
};

template <int KS, bool THIN>                            // n0 = 16 KS; THIN: out_dim <= 4 (pit.py:106 `de`): row dots, no output tile
__global__ __launch_bounds__(256, 2) void mlp_fwd64_kernel(SlabFwdArgs g) {
    constexpr int N0 = 16 * KS, XP = N0 + 4, HP = BD + 4, Q = 64 * N0 / 4 / 256;      // Q: 16-B pieces of a slab per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                   // [64][XP]
    float* hs = smem + 64 * XP;                         // [64][HP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int c1 = wave * 16 + l15;                     // this lane's hidden / output column
    const int nslabs = (g.rows + 63) / 64;
    // this wave's weight fragments: fetched once
    float4 bv[KS], w2v[4];
#pragma unroll
    for (int s = 0; s < KS; ++s) bv[s] = *reinterpret_cast<const float4*>(g.w1 + (long)c1 * N0 + 16 * s + 4 * kq);
#pragma unroll
    for (int s = 0; s < 4; ++s)
        w2v[s] = THIN ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(g.w2 + (long)c1 * BD + 16 * s + 4 * kq);
    const float bias = g.b1[c1], bias2 = THIN ? 0.0f : g.b2[c1];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.x, (unsigned)(((long)(g.rows - 1) * g.ldx + N0) * 4));
    const unsigned x_bytes = (unsigned)(((long)(g.rows - 1) * g.ldx + N0) * 4);
    float pre[Q][4];
    auto fetch = [&](int slab) {                        // the slab's X rows -> registers (rows beyond the end read as 0)
#pragma unroll
        for (int p = 0; p < Q; ++p) {
            const int q = p * 256 + tid, r = q / (N0 / 4), c4 = q % (N0 / 4);
            const long row = (long)slab * 64 + r;
            buf_load4(rx, row < g.rows ? (unsigned)((row * g.ldx + 4 * c4) * 4) : x_bytes, pre[p]);
        }
    };
    auto park_x = [&]() {
#pragma unroll
        for (int p = 0; p < Q; ++p) {
            const int q = p * 256 + tid, r = q / (N0 / 4), c4 = q % (N0 / 4);
            *reinterpret_cast<float4*>(xs + r * XP + 4 * c4) = make_float4(pre[p][0], pre[p][1], pre[p][2], pre[p][3]);
        }
    };
    // the accumulator-shaped stores (Z1, H, Z2, Y) through buffer resources sized to the tensors: 32-bit offsets, row tails dropped
    // by the hardware (mlp_bwd64 below: what the 64-bit per-lane addresses of predicated global stores cost)
    const unsigned rows64 = (unsigned)g.rows * (BD * 4u), ldy4 = (unsigned)g.ldy * 4u;
    const __amdgpu_buffer_rsrc_t rz1 = make_rsrc(g.z1, rows64), rh = make_rsrc(g.h, rows64), rz2 = make_rsrc(g.z2, g.z2 ? rows64 : 0u);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(g.y, THIN ? 0u : ((unsigned)(g.rows - 1) * (unsigned)g.ldy + BD) * 4u);
    unsigned row_c1[4], row_y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        row_c1[i] = (unsigned)((4 * kq + i) * BD + c1) * 4u;
        row_y[i] = (unsigned)(4 * kq + i) * ldy4 + (unsigned)c1 * 4u;
    }
    int slab = blockIdx.x;
    if (slab < nslabs) fetch(slab);
    for (; slab < nslabs; slab += gridDim.x) {
        const long m0 = (long)slab * 64;
        const unsigned slab_off1 = (unsigned)m0 * (BD * 4u);
        park_x();
        __syncthreads();
        if (slab + (int)gridDim.x < nslabs) fetch(slab + gridDim.x);       // the next slab travels while this one is contracted
        // ---- GEMM1 + bias + gelu: four 16-row tiles
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(xs + (rt * 16 + l15) * XP + 16 * s + 4 * kq);
                a0 = mfma_16x16x4(a.x, bv[s].x, a0);
                a1 = mfma_16x16x4(a.y, bv[s].y, a1);
                a0 = mfma_16x16x4(a.z, bv[s].z, a0);
                a1 = mfma_16x16x4(a.w, bv[s].w, a1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rt * 16 + 4 * kq + i;
                const float z = a0[i] + a1[i] + bias;
                const float hv = gelu_erf(z);
                hs[r * HP + c1] = hv;
                const unsigned off = slab_off1 + row_c1[i] + rt * (16 * BD * 4);     // (rows beyond the last: dropped by the resources' size)
                buf_store(rz1, off, z);
#ifndef PIT_SLAB_NO_H      // (timing experiment: the hidden activation not stored)
                buf_store(rh, off, hv);
#endif
            }
        }
        __syncthreads();
        if (THIN) {
            // thin output layer: a row dot per output - four lanes per row, 16 hidden values each
            const int r = tid >> 2, part = tid & 3;
            for (int o = 0; o < g.n2; ++o) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; k += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(hs + r * HP + part * 16 + k);
                    const float4 w = *reinterpret_cast<const float4*>(g.w2 + (long)o * BD + part * 16 + k);
                    acc += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
                }
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                if (part == 0 && m0 + r < g.rows) {
                    float v = acc + g.b2[o];
                    if (g.out_gelu) { g.z2[(m0 + r) * g.n2 + o] = v; v = gelu_erf(v); }
                    g.y[(m0 + r) * g.ldy + o] = v;
                }
            }
        } else {
        // ---- GEMM2 + bias (+ gelu)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 a = *reinterpret_cast<const float4*>(hs + (rt * 16 + l15) * HP + 16 * s + 4 * kq);
                    o0 = mfma_16x16x4(a.x, w2v[s].x, o0);
                    o1 = mfma_16x16x4(a.y, w2v[s].y, o1);
                    o0 = mfma_16x16x4(a.z, w2v[s].z, o0);
                    o1 = mfma_16x16x4(a.w, w2v[s].w, o1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = o0[i] + o1[i] + bias2;
                    if (g.out_gelu) { buf_store(rz2, slab_off1 + row_c1[i] + rt * (16 * BD * 4), v); v = gelu_erf(v); }
                    buf_store(ry, ((unsigned)m0 + rt * 16) * ldy4 + row_y[i], v);
                }
            }
        }
        __syncthreads();                                // xs / hs are rewritten by the next slab
    }
}

struct SlabBwdArgs {
    int rows, n0, n2, out_gelu;
    const float *w1, *w2, *z1, *z2, *d_y; long ld_dy;
    float* d_x; long ld_dx;
    float *dz1, *dz2;
};

template <int TPW, bool THIN>                           // dX column tiles (16 wide) per wave: n0 <= 64 TPW; THIN: out_dim <= 4
__global__ __launch_bounds__(256, 2) void mlp_bwd64_kernel(SlabBwdArgs g) {
    constexpr int P1 = BD + 4;
    __shared__ __attribute__((aligned(16))) float ds2[64 * P1];
    __shared__ __attribute__((aligned(16))) float ds1[64 * P1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int c1 = wave * 16 + l15;
    const int nslabs = (g.rows + 63) / 64;
    // weights once: W2 fragments of this wave's dZ1 tile, W1 fragments of its dX tiles (B(k, n) = w[k][n])
    float w2v[4][4], w1v[TPW][4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (THIN) w2v[s][e] = (s == 0 && e < g.n2) ? g.w2[(long)e * BD + c1] : 0.0f;     // w2[o][c1], o < n2 <= 4, in w2v[0][o]
            else w2v[s][e] = g.w2[(long)(16 * s + 4 * kq + e) * BD + c1];
        }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int col = (wave + 4 * t) * 16 + l15;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) w1v[t][s][e] = (col < g.n0) ? g.w1[(long)(16 * s + 4 * kq + e) * g.n0 + col] : 0.0f;
    }
    // the accumulator-shaped accesses (Z1 in, dZ1 / dX out) go through buffer resources sized to the tensors: 32-bit offsets, row
    // tails dropped by the hardware.  (As predicated global accesses their 64-bit per-lane addresses were hoisted out of the slab
    // loop and nine of them spilled: reloaded in front of the dX stores, each reload a full s_waitcnt vmcnt(0).)
    const __amdgpu_buffer_rsrc_t rz1 = make_rsrc(g.z1, (unsigned)g.rows * (BD * 4u)), rdz1 = make_rsrc(g.dz1, (unsigned)g.rows * (BD * 4u));
    const unsigned ldx4 = (unsigned)g.ld_dx * 4u;
    const __amdgpu_buffer_rsrc_t rdx = make_rsrc(g.d_x, g.d_x ? ((unsigned)(g.rows - 1) * (unsigned)g.ld_dx + (unsigned)g.n0) * 4u : 0u);
    unsigned row_c1[4], row_dx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        row_c1[i] = (unsigned)((4 * kq + i) * BD + c1) * 4u;
        row_dx[i] = (unsigned)(4 * kq + i) * ldx4;
    }
    // a thread owns 4 x float4 of the slab's 64 x 64 dY tile: rows (tid >> 4) + 16 p, columns 4 (tid & 15)
    const int pr = tid >> 4, pc = 4 * (tid & 15);
    float4 dyv[4], z2v[4];
    auto fetch = [&](int slab) {
        if (THIN) {                                     // (64 rows x n2 <= 4 values: thread r < 64 takes row r)
            const long r = (long)slab * 64 + tid;
            const bool ok = tid < 64 && r < g.rows;
            float* d = &dyv[0].x;
            float* z = &z2v[0].x;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                d[o] = (ok && o < g.n2) ? g.d_y[r * g.ld_dy + o] : 0.0f;
                z[o] = (ok && o < g.n2 && g.out_gelu) ? g.z2[r * g.n2 + o] : 0.0f;
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long r = (long)slab * 64 + pr + 16 * p;
            const bool ok = r < g.rows;
            dyv[p] = ok ? *reinterpret_cast<const float4*>(g.d_y + r * g.ld_dy + pc) : make_float4(0.f, 0.f, 0.f, 0.f);
            z2v[p] = (ok && g.out_gelu) ? *reinterpret_cast<const float4*>(g.z2 + r * BD + pc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int slab = blockIdx.x;
    if (slab < nslabs) fetch(slab);
    for (; slab < nslabs; slab += gridDim.x) {
        const long m0 = (long)slab * 64;
        // ---- dZ2 = dY * gelu'(Z2) -> LDS (+ scratch)
        if (THIN) {
            if (tid < 64) {
                float4 v = dyv[0];
                if (g.out_gelu) {
                    v.x *= gelu_erf_grad(z2v[0].x); v.y *= gelu_erf_grad(z2v[0].y);
                    v.z *= gelu_erf_grad(z2v[0].z); v.w *= gelu_erf_grad(z2v[0].w);
                    const float* vv = &v.x;
                    for (int o = 0; o < g.n2; ++o)
                        if (m0 + tid < g.rows) g.dz2[(m0 + tid) * g.n2 + o] = vv[o];
                }
                *reinterpret_cast<float4*>(ds2 + tid * P1) = v;
            }
        } else
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 v = dyv[p];
            const long r = m0 + pr + 16 * p;
            if (g.out_gelu) {
                v.x *= gelu_erf_grad(z2v[p].x); v.y *= gelu_erf_grad(z2v[p].y);
                v.z *= gelu_erf_grad(z2v[p].z); v.w *= gelu_erf_grad(z2v[p].w);
                if (r < g.rows) *reinterpret_cast<float4*>(g.dz2 + r * BD + pc) = v;
            }
            *reinterpret_cast<float4*>(ds2 + (pr + 16 * p) * P1 + pc) = v;
        }
        // gelu' argument of this wave's dZ1 tiles: requested before the barrier
        float z1v[4][4];
        const unsigned slab_off1 = (unsigned)m0 * (BD * 4u);       // byte offset of the slab's first row in the (rows, 64) tensors
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) z1v[rt][i] = buf_load(rz1, slab_off1 + row_c1[i] + rt * (16 * BD * 4));
        __syncthreads();
        if (slab + (int)gridDim.x < nslabs) fetch(slab + gridDim.x);
        // ---- dZ1 = (dZ2 W2) * gelu'(Z1)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            if (THIN) {                                 // an outer product per output, no MFMA tile
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 d = *reinterpret_cast<const float4*>(ds2 + (rt * 16 + 4 * kq + i) * P1);
                    a0[i] = (d.x * w2v[0][0] + d.y * w2v[0][1]) + (d.z * w2v[0][2] + d.w * w2v[0][3]);
                }
            } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(ds2 + (rt * 16 + l15) * P1 + 16 * s + 4 * kq);
                a0 = mfma_16x16x4(a.x, w2v[s][0], a0);
                a1 = mfma_16x16x4(a.y, w2v[s][1], a1);
                a0 = mfma_16x16x4(a.z, w2v[s][2], a0);
                a1 = mfma_16x16x4(a.w, w2v[s][3], a1);
            }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rt * 16 + 4 * kq + i;
                const float v = (a0[i] + a1[i]) * gelu_erf_grad(z1v[rt][i]);
                ds1[r * P1 + c1] = v;
                buf_store(rdz1, slab_off1 + row_c1[i] + rt * (16 * BD * 4), v);      // (rows beyond the last: dropped by the resource's size)
            }
        }
        __syncthreads();
        // ---- dX = dZ1 W1
        if (g.d_x) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int col = (wave + 4 * t) * 16 + l15;
                if ((wave + 4 * t) * 16 >= g.n0) break;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float4 a = *reinterpret_cast<const float4*>(ds1 + (rt * 16 + l15) * P1 + 16 * s + 4 * kq);
                        o0 = mfma_16x16x4(a.x, w1v[t][s][0], o0);
                        o1 = mfma_16x16x4(a.y, w1v[t][s][1], o1);
                        o0 = mfma_16x16x4(a.z, w1v[t][s][2], o0);
                        o1 = mfma_16x16x4(a.w, w1v[t][s][3], o1);
                    }
                    const unsigned tile_off = (col < g.n0) ? ((unsigned)m0 + rt * 16) * ldx4 + col * 4u : 0xF8000000u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) buf_store(rdx, tile_off + row_dx[i], o0[i] + o1[i]);
                }
            }
        }
        __syncthreads();
    }
}

int slab_grid_size(int nslabs) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    return std::min(nslabs, 2 * cus);                   // two workgroups per compute unit, each walking its share of the slabs
}

}  // namespace

// eligibility of the slab kernels (pit_mlp.hip consults it above the small regime): hid 64, full output tile, K in whole
// 16-steps, 16-B-aligned rows.  PIT_NO_SLAB_MLP=1: the two / three GEMM launches (A/B measurements).
// (thin output layers - out_dim <= 4, the decoder MLP - run on the slab variants only at very many rows: same-box A/B per step,
// Darcy b=256 (473 k rows) 1.876 vs 1.890 ms, but b=64 0.710 vs 0.693, b=32 0.423 vs 0.407, b=16 0.288 vs 0.272 against the GEMM +
// thin_* launches; PIT_NO_SLAB_THIN=1 for that form everywhere)
bool pit_mlp_slab_eligible(int rows, int n0, int n1, int n2) {
    static const bool off = getenv("PIT_NO_SLAB_MLP") != nullptr;
    static const bool thin = getenv("PIT_NO_SLAB_THIN") == nullptr;
    const bool thin_ok = thin && n2 >= 1 && n2 <= 4 && rows >= 262144;
    return !off && n1 == BD && (n2 == BD || thin_ok) && n0 % 16 == 0 && n0 >= 16 && n0 <= 256 && rows >= 4096;
}
// Do the slab kernels also replace the SMALL-regime fused kernels (mlp_fwd16 / mlp_bwd16: 16-row slabs, every workgroup
// re-reading the weights)?  Measured: NO at the Darcy decoder MLP of batch 8 (14 792 rows = 232 slabs, one per compute unit:
// 0.206 vs 0.195 ms per step - a 64-row slab is a serial ~8 us, the 925 16-row workgroups overlap better).  Only from four
// slabs per compute unit.
bool pit_mlp_slab_preferred(int rows, int n0, int n1, int n2) { return pit_mlp_slab_eligible(rows, n0, n1, n2) && rows >= 65536; }

bool try_launch_mlp_fwd64(const float* x, long ldx, int rows, int n0, int n1, int n2, const float* w1, const float* b1,
                          const float* w2, const float* b2, int out_gelu, float* z1, float* h, float* z2, float* y, long ldy,
                          hipStream_t s) {
    if (!pit_mlp_slab_eligible(rows, n0, n1, n2)) return false;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || ldx % 4 || (reinterpret_cast<uintptr_t>(w1) & 15) || (reinterpret_cast<uintptr_t>(w2) & 15)) return false;
    const bool thin = n2 <= 4;
    if (((unsigned long long)(rows - 1) * ldx + n0) * 4ull > PIT_MAX_BUFFER_BYTES) return false;
    if ((unsigned long long)rows * std::max<long>(ldy, BD) * 4ull > PIT_MAX_BUFFER_BYTES) return false;
    SlabFwdArgs g;
    g.x = x; g.ldx = ldx; g.rows = rows; g.n0 = n0; g.n2 = n2; g.w1 = w1; g.b1 = b1; g.w2 = w2; g.b2 = b2; g.out_gelu = out_gelu;
    g.z1 = z1; g.h = h; g.z2 = z2; g.y = y; g.ldy = ldy;
    const int nslabs = (rows + 63) / 64;
    const dim3 grid((unsigned)slab_grid_size(nslabs)), block(256);
    const int ks = n0 / 16;
    const size_t sm = (size_t)(64 * (n0 + 4) + 64 * (BD + 4)) * sizeof(float);
#define PIT_SLAB_F(KS_) case KS_: {                                                                                        \
        static bool once = ((void)hipFuncSetAttribute((const void*)mlp_fwd64_kernel<KS_, false>,                           \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304),                        \
                            (void)hipFuncSetAttribute((const void*)mlp_fwd64_kernel<KS_, true>,                            \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 98304), true);                 \
        (void)once;                                                                                                        \
        if (thin) hipLaunchKernelGGL((mlp_fwd64_kernel<KS_, true>), grid, block, sm, s, g);                                \
        else hipLaunchKernelGGL((mlp_fwd64_kernel<KS_, false>), grid, block, sm, s, g);                                    \
        break; }
    switch (ks) {
        PIT_SLAB_F(1) PIT_SLAB_F(2) PIT_SLAB_F(3) PIT_SLAB_F(4) PIT_SLAB_F(5) PIT_SLAB_F(6) PIT_SLAB_F(7) PIT_SLAB_F(8)
        PIT_SLAB_F(9) PIT_SLAB_F(10) PIT_SLAB_F(11) PIT_SLAB_F(12) PIT_SLAB_F(13) PIT_SLAB_F(14) PIT_SLAB_F(15) PIT_SLAB_F(16)
        default: return false;
    }
#undef PIT_SLAB_F
    return true;
}

bool try_launch_mlp_bwd64(int rows, int n0, int n1, int n2, const float* w1, const float* w2, const float* z1, const float* z2,
                          int out_gelu, const float* d_y, long ld_dy, float* d_x, long ld_dx, float* dz1, float* dz2, hipStream_t s) {
    if (!pit_mlp_slab_eligible(rows, n0, n1, n2)) return false;
    const bool thin = n2 <= 4;
    if ((unsigned long long)rows * std::max<long>(ld_dx, BD) * 4ull > PIT_MAX_BUFFER_BYTES) return false;      // (32-bit buffer offsets)
    if (!thin && ((reinterpret_cast<uintptr_t>(d_y) & 15) || ld_dy % 4 || (z2 && (reinterpret_cast<uintptr_t>(z2) & 15)) ||
                  (reinterpret_cast<uintptr_t>(dz2) & 15))) return false;
    SlabBwdArgs g;
    g.rows = rows; g.n0 = n0; g.n2 = n2; g.out_gelu = out_gelu; g.w1 = w1; g.w2 = w2; g.z1 = z1; g.z2 = z2; g.d_y = d_y; g.ld_dy = ld_dy;
    g.d_x = d_x; g.ld_dx = ld_dx; g.dz1 = dz1; g.dz2 = dz2;
    const int nslabs = (rows + 63) / 64;
    const dim3 grid((unsigned)slab_grid_size(nslabs)), block(256);
    const int tpw = (n0 + 63) / 64;
#define PIT_SLAB_B(T_) do { if (thin) hipLaunchKernelGGL((mlp_bwd64_kernel<T_, true>), grid, block, 0, s, g);                  \
                           else hipLaunchKernelGGL((mlp_bwd64_kernel<T_, false>), grid, block, 0, s, g); } while (0)
    if (tpw == 1) PIT_SLAB_B(1); else if (tpw == 2) PIT_SLAB_B(2); else if (tpw == 3) PIT_SLAB_B(3); else PIT_SLAB_B(4);
#undef PIT_SLAB_B
    return true;
}
