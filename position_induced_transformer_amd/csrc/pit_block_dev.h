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

// ---------------------------------------------------------------------------------------------- weights
struct WeightsArgs {
    const float* mesh; int L, sdim, periodic; float period;
    int n_layers, n_head, head_is_scale;
    const float* head[MAX_LAYERS];
    float *e, *q, *inv, *rowstat, *scale_out;
};

// coordinates of the 4 consecutive mesh points j0 .. j0+3 (j0 % 4 == 0) with 16-B loads: 4*sdim contiguous floats
// (32 scalar loads per lane made the kernel address-rate bound: 5.8 us for 2 M weights)
__device__ __forceinline__ void load4pts(const float* __restrict__ mesh, int j0, int sdim, float (&x)[4][3]) {
    float raw[12];
    const float4* p = reinterpret_cast<const float4*>(mesh + (long)j0 * sdim);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < sdim) t = p[v];
        raw[4 * v] = t.x; raw[4 * v + 1] = t.y; raw[4 * v + 2] = t.z; raw[4 * v + 3] = t.w;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        x[u][0] = sdim == 1 ? raw[u] : (sdim == 2 ? raw[2 * u] : raw[3 * u]);
        x[u][1] = sdim == 1 ? 0.0f : (sdim == 2 ? raw[2 * u + 1] : raw[3 * u + 1]);
        x[u][2] = sdim == 3 ? raw[3 * u + 2] : 0.0f;
    }
}

// one wave per (layer, head, row): keys 4*lane + 256*r, 16-B stores; `wg`: the workgroup's index among ceil(rows / 4)
__device__ __forceinline__ void block_weights_body(const WeightsArgs& a, long wg) {
    const int lane = threadIdx.x & 63;
    const long row_id = wg * 4 + (threadIdx.x >> 6);          // (layer, head, n)
    const long rows_total = (long)a.n_layers * a.n_head * a.L;
    if (row_id >= rows_total) return;
    const int n = (int)(row_id % a.L);
    const int lh = (int)(row_id / a.L);
    const int l = lh / a.n_head, h = lh % a.n_head;
    const float hv = a.head[l][h];
    const float c = a.head_is_scale ? hv : head_scale_from_lmda(hv);
    const float* xo = a.mesh + (long)n * a.sdim;
    const float ox = xo[0], oy = a.sdim > 1 ? xo[1] : 0.0f, oz = a.sdim > 2 ? xo[2] : 0.0f;
    const bool per = a.periodic != 0;
    float rsum = 0.0f, qsum = 0.0f;
    float* erow = a.e + row_id * a.L;
    float* qrow = a.q ? a.q + row_id * a.L : nullptr;
    auto weight = [&](int j, float& m) {
        const float* xi = a.mesh + (long)j * a.sdim;
        m = sq_dist3(ox, oy, oz, xi[0], a.sdim > 1 ? xi[1] : 0.0f, a.sdim > 2 ? xi[2] : 0.0f, per, a.period);
        return __expf(-__fmul_rn(m, c));                                    // S_min = 0: the row holds its own point
    };
    const bool vec_mesh = (reinterpret_cast<uintptr_t>(a.mesh) & 15) == 0;
    if (a.L <= 1024) {                                                      // the row stays in registers: one pass
        float pv[4][4], mv[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j0 = 4 * lane + 256 * r;
            float xk[4][3];
            if (j0 < a.L && vec_mesh) load4pts(a.mesh, j0, a.sdim, xk);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pv[r][u] = 0.0f; mv[r][u] = 0.0f;
                if (j0 < a.L) {
                    if (vec_mesh) {
                        mv[r][u] = sq_dist3(ox, oy, oz, xk[u][0], xk[u][1], xk[u][2], per, a.period);
                        pv[r][u] = __expf(-__fmul_rn(mv[r][u], c));
                    } else {
                        pv[r][u] = weight(j0 + u, mv[r][u]);
                    }
                    rsum += pv[r][u];
                    qsum += pv[r][u] * mv[r][u];
                }
            }
        }
        rsum = wave_sum(rsum);
        qsum = wave_sum(qsum);
        const float inv = rsum > 0.0f ? 1.0f / rsum : 0.0f;
        const float mbar = qsum * inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j0 = 4 * lane + 256 * r;
            if (j0 < a.L) {
                *reinterpret_cast<float4*>(erow + j0) = make_float4(pv[r][0], pv[r][1], pv[r][2], pv[r][3]);
                if (qrow)
                    *reinterpret_cast<float4*>(qrow + j0) = make_float4(pv[r][0] * (mv[r][0] - mbar) * inv, pv[r][1] * (mv[r][1] - mbar) * inv,
                                                                        pv[r][2] * (mv[r][2] - mbar) * inv, pv[r][3] * (mv[r][3] - mbar) * inv);
            }
        }
        if (lane == 0) {
            a.inv[row_id] = inv;
            float4 st; st.x = __builtin_inff(); st.y = 0.0f; st.z = inv; st.w = mbar;
            *reinterpret_cast<float4*>(a.rowstat + row_id * 4) = st;
            if (n == 0) a.scale_out[lh] = c;
        }
        return;
    }
    // longer rows: pass 1 row sums, pass 2 re-forms the weights (an exp is cheaper than parking them)
    for (int j0 = 4 * lane; j0 < a.L; j0 += 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float m;
            const float p = weight(j0 + u, m);
            rsum += p;
            qsum += p * m;
        }
    }
    rsum = wave_sum(rsum);
    qsum = wave_sum(qsum);
    const float inv = rsum > 0.0f ? 1.0f / rsum : 0.0f;
    const float mbar = qsum * inv;
    for (int j0 = 4 * lane; j0 < a.L; j0 += 256) {
        float4 ev, qv;
        float* ep = &ev.x;
        float* qp = &qv.x;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float m;
            const float p = weight(j0 + u, m);
            ep[u] = p;
            qp[u] = p * (m - mbar) * inv;
        }
        *reinterpret_cast<float4*>(erow + j0) = ev;
        if (qrow) *reinterpret_cast<float4*>(qrow + j0) = qv;
    }
    if (lane == 0) {
        a.inv[row_id] = inv;
        float4 st; st.x = __builtin_inff(); st.y = 0.0f; st.z = inv; st.w = mbar;
        *reinterpret_cast<float4*>(a.rowstat + row_id * 4) = st;
        if (n == 0) a.scale_out[lh] = c;
    }
}
__host__ inline int block_weights_wgs(int n_layers, int n_head, int L) { return (int)(((long)n_layers * n_head * L + 3) / 4); }
// pit_block_weights' argument checks and kernel arguments (include/pit_hip.h)
__host__ inline int fill_weights_args(WeightsArgs& a, const float* mesh, int n_pts, int space_dim, int metric, float period,
                                      int n_layers, const float* const* heads, int head_is_scale, int n_head, float* e, float* q,
                                      float* inv, float* rowstat, float* scale_out) {
    if (!mesh || !heads || !e || !inv || !rowstat || !scale_out) return PIT_ERR_NULL;      // (q may be NULL: forward only)
    if (n_pts <= 0 || n_pts % 4 != 0 || space_dim < 1 || space_dim > 3 || n_layers < 1 || n_layers > MAX_LAYERS ||
        n_head < 1) return PIT_ERR_SIZE;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    if (!aligned16(e) || (q && !aligned16(q)) || !aligned16(rowstat)) return PIT_ERR_SIZE;
    a.mesh = mesh; a.L = n_pts; a.sdim = space_dim; a.periodic = metric != PIT_METRIC_EUCLID; a.period = period;
    a.n_layers = n_layers; a.n_head = n_head; a.head_is_scale = head_is_scale;
    for (int l = 0; l < n_layers; ++l) {
        if (!heads[l]) return PIT_ERR_NULL;
        a.head[l] = heads[l];
    }
    a.e = e; a.q = q; a.inv = inv; a.rowstat = rowstat; a.scale_out = scale_out;
    return 0;
}


}  // namespace
