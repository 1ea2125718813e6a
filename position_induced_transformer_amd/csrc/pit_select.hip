// Selection pre-pass: per-row order statistics of the UNSCALED squared distance.
//
// Replaces the full row sort inside torch.quantile (pit.py:49,136,197,255).  Squared
// distances are >= +0, so their fp32 bit patterns order like unsigned integers; the
// k-th smallest key is found by an MSB-first bitwise search (31 counting passes), the
// (k+1)-th by one more pass.  SURVEY appendix A.4: because x -> fl(c*x) is monotone the
// scaled order statistics the reference interpolates are fl(c*m_(k)), fl(c*m_(k+1)).
//
// Two kernels: `select_rows_reg` keeps a row (n_in <= 64*ITEMS) in registers, one
// wavefront per row, counting with ballots + scalar popcounts; `select_rows_stream`
// (long rows, e.g. zero-shot super-resolution J = 177k) recomputes the distances from
// the coordinates on every pass with one 256-thread workgroup per row.
#include "pit_common.h"

namespace {

struct SelectArgs {
    const float* mesh_out;
    const float* mesh_in;
    float* stats;       // [3][rows]
    int mesh_batch, n_out, n_in, sdim;
    int periodic;       // 0 euclid, 1 periodic
    int coords_used;    // coordinates entering the distance (periodic1d: 1)
    float period;
    int rank_k, need_kth;
};

__device__ __forceinline__ void load_point(const float* p, int sdim, int used, float& x, float& y, float& z) {
    x = p[0];
    y = (used > 1) ? p[1] : 0.0f;
    z = (used > 2) ? p[2] : 0.0f;
    (void)sdim;
}

template <int ITEMS>
__global__ __launch_bounds__(256) void select_rows_reg(SelectArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long rows = (long)a.mesh_batch * a.n_out;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;                      // whole wave exits together
    const int mb = (int)(row / a.n_out);
    const float* po = a.mesh_out + row * a.sdim;
    const float* pin = a.mesh_in + (long)mb * a.n_in * a.sdim;
    float ox, oy, oz;
    load_point(po, a.sdim, a.coords_used, ox, oy, oz);

    uint32_t key[ITEMS];
    uint32_t kmin = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int j = lane + 64 * i;
        uint32_t k = 0xFFFFFFFFu;                 // padding sorts last
        if (j < a.n_in) {
            float ix, iy, iz;
            load_point(pin + (long)j * a.sdim, a.sdim, a.coords_used, ix, iy, iz);
            k = __float_as_uint(sq_dist3(ox, oy, oz, ix, iy, iz, a.periodic != 0, a.period));
        }
        key[i] = k;
        kmin = min(kmin, k);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o));

    uint32_t vk = kmin, vk1 = kmin;
    if (a.need_kth) {
        const int k = a.rank_k;
        uint32_t prefix = 0;
        for (int bit = 30; bit >= 0; --bit) {       // bit 31 (sign) is never set on real keys
            const uint32_t cand = prefix | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i)
                cnt += __popcll(__builtin_amdgcn_ballot_w64(key[i] < cand));
            if (cnt <= k) prefix = cand;            // k-th key has this bit set
        }
        vk = prefix;
        int cnt_le = 0;
        uint32_t next = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            cnt_le += __popcll(__builtin_amdgcn_ballot_w64(key[i] <= vk));
            if (key[i] > vk) next = min(next, key[i]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) next = min(next, (uint32_t)__shfl_xor((int)next, o));
        // (k+1)-th order statistic, clipped to the last element (torch.quantile's upper index)
        vk1 = (cnt_le >= k + 2 || k + 1 > a.n_in - 1) ? vk : next;
    }
    if (lane == 0) {
        a.stats[row] = __uint_as_float(vk);
        a.stats[rows + row] = __uint_as_float(vk1);
        a.stats[2 * rows + row] = __uint_as_float(kmin);
    }
}

// one workgroup per row, distances recomputed per pass
__global__ __launch_bounds__(256) void select_rows_stream(SelectArgs a) {
    __shared__ int s_cnt[4];
    __shared__ uint32_t s_min[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long rows = (long)a.mesh_batch * a.n_out;
    const long row = blockIdx.x;
    const int mb = (int)(row / a.n_out);
    const float* po = a.mesh_out + row * a.sdim;
    const float* pin = a.mesh_in + (long)mb * a.n_in * a.sdim;
    float ox, oy, oz;
    load_point(po, a.sdim, a.coords_used, ox, oy, oz);

    auto key_at = [&](int j) -> uint32_t {
        float ix, iy, iz;
        load_point(pin + (long)j * a.sdim, a.sdim, a.coords_used, ix, iy, iz);
        return __float_as_uint(sq_dist3(ox, oy, oz, ix, iy, iz, a.periodic != 0, a.period));
    };
    // block-wide count of keys < cand (strict) or <= cand, and min of keys > bound
    auto block_count = [&](uint32_t cand, bool inclusive) -> int {
        int c = 0;
        for (int j = threadIdx.x; j < a.n_in; j += 256) {
            const uint32_t k = key_at(j);
            c += inclusive ? (k <= cand) : (k < cand);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        __syncthreads();
        if (lane == 0) s_cnt[wave] = c;
        __syncthreads();
        return s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    };
    auto block_min_above = [&](uint32_t bound, bool strictly_above) -> uint32_t {
        uint32_t m = 0xFFFFFFFFu;
        for (int j = threadIdx.x; j < a.n_in; j += 256) {
            const uint32_t k = key_at(j);
            if (!strictly_above || k > bound) m = min(m, k);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
        __syncthreads();
        if (lane == 0) s_min[wave] = m;
        __syncthreads();
        return min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
    };

    const uint32_t kmin = block_min_above(0, false);
    uint32_t vk = kmin, vk1 = kmin;
    if (a.need_kth) {
        const int k = a.rank_k;
        uint32_t prefix = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t cand = prefix | (1u << bit);
            if (block_count(cand, false) <= k) prefix = cand;
        }
        vk = prefix;
        const int cnt_le = block_count(vk, true);
        const uint32_t next = block_min_above(vk, true);
        vk1 = (cnt_le >= k + 2 || k + 1 > a.n_in - 1) ? vk : next;
    }
    if (threadIdx.x == 0) {
        a.stats[row] = __uint_as_float(vk);
        a.stats[rows + row] = __uint_as_float(vk1);
        a.stats[2 * rows + row] = __uint_as_float(kmin);
    }
}

template <int ITEMS>
void launch_reg(const SelectArgs& a, hipStream_t s) {
    const long rows = (long)a.mesh_batch * a.n_out;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipLaunchKernelGGL(select_rows_reg<ITEMS>, dim3(grid), dim3(256), 0, s, a);
}

__global__ void head_scale_kernel(const float* lmda, int n_head, float* out) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < n_head) out[h] = head_scale_from_lmda(lmda[h]);
}

}  // namespace

extern "C" int pit_head_scale(const float* lmda, int n_head, float* scale_out, void* stream) {
    if (!lmda || !scale_out) return PIT_ERR_NULL;
    if (n_head <= 0) return PIT_ERR_SIZE;
    hipLaunchKernelGGL(head_scale_kernel, dim3((n_head + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       lmda, n_head, scale_out);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_select_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                              int space_dim, int metric, float period, int rank_k, int need_kth,
                              float* stats, void* stream) {
    if (!mesh_out || !mesh_in || !stats) return PIT_ERR_NULL;
    if (mesh_batch <= 0 || n_out <= 0 || n_in <= 0 || space_dim < 1 || space_dim > 3) return PIT_ERR_SIZE;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    if (need_kth && (rank_k < 0 || rank_k > n_in - 1)) return PIT_ERR_SIZE;
    SelectArgs a;
    a.mesh_out = mesh_out; a.mesh_in = mesh_in; a.stats = stats;
    a.mesh_batch = mesh_batch; a.n_out = n_out; a.n_in = n_in; a.sdim = space_dim;
    a.periodic = (metric != PIT_METRIC_EUCLID);
    a.coords_used = (metric == PIT_METRIC_PERIODIC1D) ? 1 : space_dim;
    a.period = period; a.rank_k = rank_k; a.need_kth = need_kth;
    hipStream_t s = (hipStream_t)stream;
    const int items = (n_in + 63) / 64;
    if (items <= 1) launch_reg<1>(a, s);
    else if (items <= 2) launch_reg<2>(a, s);
    else if (items <= 4) launch_reg<4>(a, s);
    else if (items <= 8) launch_reg<8>(a, s);
    else if (items <= 16) launch_reg<16>(a, s);
    else if (items <= 32) launch_reg<32>(a, s);
    else if (items <= 64) launch_reg<64>(a, s);
    else {
        const long rows = (long)mesh_batch * n_out;
        hipLaunchKernelGGL(select_rows_stream, dim3((unsigned)rows), dim3(256), 0, s, a);
    }
    PIT_CHECK_LAUNCH();
    return 0;
}
