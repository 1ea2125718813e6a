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
#include <cstdlib>

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

// MSB-first bitwise search over NI keys per lane: the key of 0-based rank k in the wave's multiset
template <int NI>
__device__ __forceinline__ uint32_t wave_kth(const uint32_t (&key)[NI], int k) {
    uint32_t prefix = 0;
    for (int bit = 30; bit >= 0; --bit) {       // bit 31 (sign) is never set on real keys
        const uint32_t cand = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) cnt += __popcll(__builtin_amdgcn_ballot_w64(key[i] < cand));
        if (cnt <= k) prefix = cand;            // the k-th key has this bit set
    }
    return prefix;
}

// Selection AND candidate lists in one pass over the row (pit_plan_fwd): the keys stay in registers,
// so the lists cost one more compare per key instead of a second distance pass.  The search itself
// is narrowed first: the (k+2)-th smallest of the 64 lane minima is an upper bound U of m_(k+1)
// (at least k+2 keys are <= U), typically only a few more than k+2 keys are <= U, and when they fit
// one per lane the exact order statistics come from a search over ONE key per lane (31 passes x 1
// compare instead of x ITEMS).
template <int ITEMS>
__global__ __launch_bounds__(256) void plan_rows_reg(SelectArgs a, int cap, int* __restrict__ nbr_idx,
                                                     int* __restrict__ nbr_cnt, int* __restrict__ counts) {
    __shared__ uint32_t s_cand[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long rows = (long)a.mesh_batch * a.n_out;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;                      // whole wave exits together (no block barrier below)
    const int mb = (int)(row / a.n_out);
    const float* po = a.mesh_out + row * a.sdim;
    const float* pin = a.mesh_in + (long)mb * a.n_in * a.sdim;
    float ox, oy, oz;
    load_point(po, a.sdim, a.coords_used, ox, oy, oz);

    uint32_t key[ITEMS];
    uint32_t lmin = 0xFFFFFFFFu;
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
        lmin = min(lmin, k);
    }
    uint32_t kmin = lmin;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o));

    const int k = a.rank_k;
    uint32_t vk;
    bool narrowed = false;
    if (ITEMS >= 4 && k + 2 <= 32) {
        const uint32_t one[1] = {lmin};
        const uint32_t U = wave_kth<1>(one, k + 1);            // (k+2)-th smallest lane minimum >= m_(k+1)
        // (a 64-lane bitonic sort instead of the 31 counting passes was measured slower: 460 vs 412 us on
        // the NACA decoder - 21 dependent cross-lane permutes)
        int total = 0;
        bool fits = (U != 0xFFFFFFFFu);
        if (fits) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) total += __popcll(__builtin_amdgcn_ballot_w64(key[i] <= U));
            fits = total <= 64;
        }
        if (fits) {                                            // wave-uniform
            s_cand[wave][lane] = 0xFFFFFFFFu;
            int base = 0;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const bool in = key[i] <= U;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                if (in) s_cand[wave][base + __popcll(m & ((1ull << lane) - 1ull))] = key[i];
                base += __popcll(m);
            }
            // same-wave LDS traffic only: the LDS pipeline executes a wave's accesses in program order
            const uint32_t cnd[1] = {s_cand[wave][lane]};
            vk = wave_kth<1>(cnd, k);
            narrowed = true;
        }
    }
    if (!narrowed) vk = wave_kth<ITEMS>(key, k);
    int cnt_le = 0;
    uint32_t next = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        cnt_le += __popcll(__builtin_amdgcn_ballot_w64(key[i] <= vk));
        if (key[i] > vk) next = min(next, key[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) next = min(next, (uint32_t)__shfl_xor((int)next, o));
    const uint32_t vk1 = (cnt_le >= k + 2 || k + 1 > a.n_in - 1) ? vk : next;
    if (lane == 0) {
        a.stats[row] = __uint_as_float(vk);
        a.stats[rows + row] = __uint_as_float(vk1);
        a.stats[2 * rows + row] = __uint_as_float(kmin);
    }
    // ---- candidate list: keys with m <= m_(k+1) * (1 + 2^-21), in key order (as neighbors_kernel)
    const float bound = __uint_as_float(vk1) * 1.00000047683715820312f;
    int total = 0;
    int* out = nbr_idx + row * cap;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int j = lane + 64 * i;
        const bool in = (j < a.n_in) && (__uint_as_float(key[i]) <= bound);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
        const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
        if (in && pos < cap) {
            out[pos] = j;
            if (counts) atomicAdd(counts + (long)mb * a.n_in + j, 1);
        }
        total += __popcll(mask);
    }
    if (lane == 0) nbr_cnt[row] = total;
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

// ------------------------------------------------------------------------------------
// Candidate lists for the masked (locality < 1) layers.
//
// kept  <=>  fl(c*m) <= T  and  T <= fl(c*m_(k+1)),  so every kept key satisfies
// fl(c*m) <= fl(c*m_(k+1)); with m > m_(k+1) that needs both products to round to the same
// float, i.e. m <= m_(k+1)*(1+2^-22).  The list of keys with m <= m_(k+1)*(1+2^-21) is
// therefore a superset of the kept set for EVERY head scale c, depends only on the meshes, and
// has k+2 entries plus ties.  The sparse attention kernels evaluate the exact mask on it.
// A row whose count exceeds `cap` keeps its true count (list truncated): consumers treat
// count > cap as "scan all keys".
__global__ __launch_bounds__(256) void neighbors_kernel(SelectArgs a, int cap, int* __restrict__ nbr_idx,
                                                        int* __restrict__ nbr_cnt, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long rows = (long)a.mesh_batch * a.n_out;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int mb = (int)(row / a.n_out);
    const float* po = a.mesh_out + row * a.sdim;
    const float* pin = a.mesh_in + (long)mb * a.n_in * a.sdim;
    float ox, oy, oz;
    load_point(po, a.sdim, a.coords_used, ox, oy, oz);
    const float bound = a.stats[rows + row] * 1.00000047683715820312f;     // m_(k+1) * (1 + 2^-21)
    int total = 0;
    int* out = nbr_idx + row * cap;
    for (int j0 = 0; j0 < a.n_in; j0 += 64) {
        const int j = j0 + lane;
        bool in = false;
        if (j < a.n_in) {
            float ix, iy, iz;
            load_point(pin + (long)j * a.sdim, a.sdim, a.coords_used, ix, iy, iz);
            in = sq_dist3(ox, oy, oz, ix, iy, iz, a.periodic != 0, a.period) <= bound;
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
        const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
        if (in && pos < cap) {
            out[pos] = j;
            if (counts) atomicAdd(counts + (long)mb * a.n_in + j, 1);     // per-key counts for the transpose
        }
        total += __popcll(mask);
    }
    if (lane == 0) nbr_cnt[row] = total;
}

// reverse lists (key -> rows that list it), CSR per mesh sample: counts come from
// neighbors_kernel, then scan and fill (slots of overflowed rows stay -1)
__global__ __launch_bounds__(256) void nbr_scan_kernel(const int* __restrict__ counts, int n_in,
                                                       int* __restrict__ rev_ptr, int* __restrict__ cursor) {
    // one workgroup per mesh sample: exclusive scan of counts[mb][0..n_in) -> rev_ptr[mb][0..n_in]
    __shared__ int s_part[256];
    __shared__ int s_carry;
    const int mb = blockIdx.x;
    const int* c = counts + (long)mb * n_in;
    int* p = rev_ptr + (long)mb * (n_in + 1);
    int* cur = cursor + (long)mb * n_in;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_in; base += 256) {
        const int j = base + threadIdx.x;
        const int v = (j < n_in) ? c[j] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {          // Hillis-Steele inclusive scan
            const int t = (threadIdx.x >= off) ? s_part[threadIdx.x - off] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = s_part[threadIdx.x];
        const int carry = s_carry;
        if (j < n_in) { p[j] = carry + incl - v; cur[j] = carry + incl - v; }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) p[n_in] = s_carry;
}

// four list slots per thread, all four returning atomics in flight before any result is used
__global__ void nbr_fill_kernel(const int* __restrict__ nbr_idx, const int* __restrict__ nbr_cnt, long rows,
                                int n_out, int n_in, int cap, int* __restrict__ cursor, int* __restrict__ rev_row,
                                long rev_stride) {
    const long total = rows * cap;
    const long stride = (long)gridDim.x * blockDim.x;
    const long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int pos[4], local[4], mbs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long e = e0 + r * stride;
        pos[r] = -1;
        if (e >= total) continue;
        const long row = e / cap;
        const int i = (int)(e - row * cap);
        const int cnt = nbr_cnt[row];
        if (cnt > cap || i >= cnt) continue;
        const int mb = (int)(row / n_out);
        mbs[r] = mb;
        local[r] = (int)(row - (long)mb * n_out);
        pos[r] = atomicAdd(cursor + (long)mb * n_in + nbr_idx[e], 1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (pos[r] >= 0) rev_row[(long)mbs[r] * rev_stride + pos[r]] = local[r];
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

// plain fill kernel used instead of hipMemsetAsync: a memset NODE of a captured hipGraph is not reliably
// ordered against the neighbouring kernel nodes on ROCm 7.2 once other work ran between two replays
// (observed: counts not yet zero when plan_rows_reg adds to them -> cursors past the lists -> wild writes)
__global__ __launch_bounds__(256) void fill_int_kernel(int* __restrict__ p, long n, int v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
static void fill_int(int* p, long n, int v, hipStream_t s) {
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 1024L);
    hipLaunchKernelGGL(fill_int_kernel, dim3(std::max(1u, blocks)), dim3(256), 0, s, p, n, v);
}

// Transposed lists with workgroup-level aggregation (n_in <= NBR_LDS_KEYS): a key of a small key set is
// listed by thousands of rows (NACA decoder: 11 271 rows x ~16 listed keys over 728 latent points), so one
// global atomic per list ENTRY on counts[key] / cursor[key] is thousands of adds per address.  A workgroup
// takes NBR_ROWS consecutive rows of one mesh sample, histograms their entries in LDS and touches every
// global counter once: counting (nbr_count_lds) and - after the scan - slot reservation + fill (nbr_fill_lds).
constexpr int NBR_LDS_KEYS = 4096, NBR_ROWS = 128;

__global__ __launch_bounds__(256) void nbr_count_lds(const int* __restrict__ nbr_idx, const int* __restrict__ nbr_cnt,
                                                      int n_out, int n_in, int cap, int* __restrict__ counts) {
    __shared__ int hist[NBR_LDS_KEYS];
    const int mb = blockIdx.y, r0 = blockIdx.x * NBR_ROWS;
    for (int k = threadIdx.x; k < n_in; k += 256) hist[k] = 0;
    __syncthreads();
    const int nrows = min(NBR_ROWS, n_out - r0);
    for (int e = threadIdx.x; e < nrows * cap; e += 256) {
        const int r = e / cap, i = e - r * cap;
        const long row = (long)mb * n_out + r0 + r;
        const int cnt = nbr_cnt[row];
        if (i < min(cnt, cap)) atomicAdd(&hist[nbr_idx[row * cap + i]], 1);      // (overflowed rows count their first cap keys, as before)
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_in; k += 256)
        if (hist[k]) atomicAdd(counts + (long)mb * n_in + k, hist[k]);
}

__global__ __launch_bounds__(256) void nbr_fill_lds(const int* __restrict__ nbr_idx, const int* __restrict__ nbr_cnt,
                                                     int n_out, int n_in, int cap, int* __restrict__ cursor,
                                                     int* __restrict__ rev_row, long rev_stride) {
    __shared__ int hist[NBR_LDS_KEYS];      // entries of this workgroup per key, then the running slot inside the reservation
    __shared__ int base[NBR_LDS_KEYS];
    const int mb = blockIdx.y, r0 = blockIdx.x * NBR_ROWS;
    for (int k = threadIdx.x; k < n_in; k += 256) hist[k] = 0;
    __syncthreads();
    const int nrows = min(NBR_ROWS, n_out - r0);
    for (int e = threadIdx.x; e < nrows * cap; e += 256) {
        const int r = e / cap, i = e - r * cap;
        const long row = (long)mb * n_out + r0 + r;
        const int cnt = nbr_cnt[row];
        if (cnt <= cap && i < cnt) atomicAdd(&hist[nbr_idx[row * cap + i]], 1);   // overflowed rows stay out of the transpose
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_in; k += 256) {
        const int c = hist[k];
        base[k] = c ? atomicAdd(cursor + (long)mb * n_in + k, c) : 0;             // ONE reservation per key and workgroup
        hist[k] = 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nrows * cap; e += 256) {
        const int r = e / cap, i = e - r * cap;
        const long row = (long)mb * n_out + r0 + r;
        const int cnt = nbr_cnt[row];
        if (cnt <= cap && i < cnt) {
            const int key = nbr_idx[row * cap + i];
            const int pos = base[key] + atomicAdd(&hist[key], 1);
            rev_row[(long)mb * rev_stride + pos] = r0 + r;
        }
    }
}

// transposed lists from nbr_idx / counts: scan + fill.  counts_done = the counts were already accumulated (by
// the list-emitting kernel's own atomics); otherwise they are counted here with workgroup aggregation.
static int launch_transpose(const int* nbr_idx, const int* nbr_cnt, int mesh_batch, int n_out, int n_in, int cap,
                     int* rev_ptr, int* rev_row, int* counts, int* cursor, bool counts_done, hipStream_t s) {
    const long rows = (long)mesh_batch * n_out;
    const bool lds = n_in <= NBR_LDS_KEYS;
    const dim3 agrid((n_out + NBR_ROWS - 1) / NBR_ROWS, mesh_batch);
    if (!counts_done) {
        if (!lds) return PIT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(nbr_count_lds, agrid, dim3(256), 0, s, nbr_idx, nbr_cnt, n_out, n_in, cap, counts);
        PIT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(nbr_scan_kernel, dim3(mesh_batch), dim3(256), 0, s, counts, n_in, rev_ptr, cursor);
    PIT_CHECK_LAUNCH();
    if (lds) {
        hipLaunchKernelGGL(nbr_fill_lds, agrid, dim3(256), 0, s, nbr_idx, nbr_cnt, n_out, n_in, cap, cursor, rev_row,
                           (long)n_out * cap);
    } else {
        const unsigned blocks = (unsigned)((rows * cap + 1023) / 1024);         // 4 slots per thread
        hipLaunchKernelGGL(nbr_fill_kernel, dim3(blocks), dim3(256), 0, s, nbr_idx, nbr_cnt, rows, n_out, n_in, cap, cursor,
                           rev_row, (long)n_out * cap);
    }
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_plan_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                            int space_dim, int metric, float period, int rank_k, float* stats, int cap,
                            int* nbr_idx, int* nbr_cnt, int* rev_ptr, int* rev_row, int* workspace, void* stream) {
    if (!mesh_out || !mesh_in || !stats || !nbr_idx || !nbr_cnt) return PIT_ERR_NULL;
    if (rev_ptr && (!rev_row || !workspace)) return PIT_ERR_NULL;
    if (mesh_batch <= 0 || n_out <= 0 || n_in <= 0 || space_dim < 1 || space_dim > 3 || cap <= 0) return PIT_ERR_SIZE;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    if (rank_k < 0 || rank_k > n_in - 1) return PIT_ERR_SIZE;
    const int items = (n_in + 63) / 64;
    if (items > 64 || getenv("PIT_NO_FUSED_PLAN")) {          // long rows: the two streaming passes
        int rc = pit_select_fwd(mesh_out, mesh_in, mesh_batch, n_out, n_in, space_dim, metric, period, rank_k, 1, stats,
                                stream);
        if (rc) return rc;
        return pit_neighbors_fwd(mesh_out, mesh_in, mesh_batch, n_out, n_in, space_dim, metric, period, stats, cap,
                                 nbr_idx, nbr_cnt, rev_ptr, rev_row, workspace, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    SelectArgs a;
    a.mesh_out = mesh_out; a.mesh_in = mesh_in; a.stats = stats;
    a.mesh_batch = mesh_batch; a.n_out = n_out; a.n_in = n_in; a.sdim = space_dim;
    a.periodic = (metric != PIT_METRIC_EUCLID);
    a.coords_used = (metric == PIT_METRIC_PERIODIC1D) ? 1 : space_dim;
    a.period = period; a.rank_k = rank_k; a.need_kth = 1;
    const long rows = (long)mesh_batch * n_out;
    int* counts = rev_ptr ? workspace : nullptr;
    int* cursor = rev_ptr ? workspace + (long)mesh_batch * n_in : nullptr;
    if (rev_ptr) {
        fill_int(counts, (long)mesh_batch * n_in, 0, s);
        fill_int(rev_row, rows * cap, -1, s);
    }
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const bool agg = rev_ptr && n_in <= NBR_LDS_KEYS;             // counts by workgroup aggregation (launch_transpose)
    int* kcounts = agg ? nullptr : counts;
#define PIT_PLAN(I_) hipLaunchKernelGGL(plan_rows_reg<I_>, grid, block, 0, s, a, cap, nbr_idx, nbr_cnt, kcounts)
    if (items <= 1) PIT_PLAN(1);
    else if (items <= 2) PIT_PLAN(2);
    else if (items <= 4) PIT_PLAN(4);
    else if (items <= 8) PIT_PLAN(8);
    else if (items <= 12) PIT_PLAN(12);
    else if (items <= 16) PIT_PLAN(16);
    else if (items <= 24) PIT_PLAN(24);
    else if (items <= 32) PIT_PLAN(32);
    else PIT_PLAN(64);
#undef PIT_PLAN
    PIT_CHECK_LAUNCH();
    if (rev_ptr) return launch_transpose(nbr_idx, nbr_cnt, mesh_batch, n_out, n_in, cap, rev_ptr, rev_row, counts, cursor, !agg, s);
    return 0;
}

extern "C" int pit_neighbors_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                                 int space_dim, int metric, float period, const float* stats, int cap,
                                 int* nbr_idx, int* nbr_cnt, int* rev_ptr, int* rev_row, int* workspace,
                                 void* stream) {
    if (!mesh_out || !mesh_in || !stats || !nbr_idx || !nbr_cnt) return PIT_ERR_NULL;
    if (rev_ptr && (!rev_row || !workspace)) return PIT_ERR_NULL;
    if (mesh_batch <= 0 || n_out <= 0 || n_in <= 0 || space_dim < 1 || space_dim > 3 || cap <= 0) return PIT_ERR_SIZE;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    hipStream_t s = (hipStream_t)stream;
    SelectArgs a;
    a.mesh_out = mesh_out; a.mesh_in = mesh_in; a.stats = const_cast<float*>(stats);
    a.mesh_batch = mesh_batch; a.n_out = n_out; a.n_in = n_in; a.sdim = space_dim;
    a.periodic = (metric != PIT_METRIC_EUCLID);
    a.coords_used = (metric == PIT_METRIC_PERIODIC1D) ? 1 : space_dim;
    a.period = period; a.rank_k = 0; a.need_kth = 1;
    const long rows = (long)mesh_batch * n_out;
    int* counts = rev_ptr ? workspace : nullptr;                       // mesh_batch * n_in
    int* cursor = rev_ptr ? workspace + (long)mesh_batch * n_in : nullptr;
    if (rev_ptr) {
        fill_int(counts, (long)mesh_batch * n_in, 0, s);
        fill_int(rev_row, rows * cap, -1, s);
    }
    const bool agg = rev_ptr && n_in <= NBR_LDS_KEYS;
    hipLaunchKernelGGL(neighbors_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a, cap, nbr_idx, nbr_cnt,
                       agg ? nullptr : counts);
    PIT_CHECK_LAUNCH();
    if (rev_ptr) return launch_transpose(nbr_idx, nbr_cnt, mesh_batch, n_out, n_in, cap, rev_ptr, rev_row, counts, cursor, !agg, s);
    return 0;
}
