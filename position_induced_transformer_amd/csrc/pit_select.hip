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
__device__ __forceinline__ void plan_row_wave(const SelectArgs& a, long row, int cap, int* __restrict__ nbr_idx,
                                              int* __restrict__ nbr_cnt, int* __restrict__ counts, uint32_t (*s_cand)[64]) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long rows = (long)a.mesh_batch * a.n_out;
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

template <int ITEMS>
__global__ __launch_bounds__(256) void plan_rows_reg(SelectArgs a, int cap, int* __restrict__ nbr_idx,
                                                     int* __restrict__ nbr_cnt, int* __restrict__ counts) {
    __shared__ uint32_t s_cand[4][64];
    const long rows = (long)a.mesh_batch * a.n_out;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                      // whole wave exits together (no block barrier below)
    plan_row_wave<ITEMS>(a, row, cap, nbr_idx, nbr_cnt, counts, s_cand);
}

// rows plan_rows_lane flagged (count = -1: more candidates than a lane's column holds): a wavefront each, the algorithm above
template <int ITEMS>
__global__ __launch_bounds__(256) void plan_rows_fix(SelectArgs a, int cap, int* __restrict__ nbr_idx,
                                                     int* __restrict__ nbr_cnt, int* __restrict__ counts) {
    __shared__ uint32_t s_cand[4][64];
    const long rows = (long)a.mesh_batch * a.n_out;
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (long base = wid * 64; base < rows; base += nw * 64) {         // a wave scans 64 counts at a time
        const long rr = base + lane;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(rr < rows && nbr_cnt[rr] < 0);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            plan_row_wave<ITEMS>(a, base + l, cap, nbr_idx, nbr_cnt, counts, s_cand);
        }
    }
}

// ------------------------------------------------------------------------------------
// Round 4: selection AND candidate lists with ONE ROW PER LANE (per-sample meshes: the plan is rebuilt every step -
// train_naca.py:62-65, train_elasticity.py:46).  plan_rows_reg gives a whole wavefront to each row: ~650 wave
// instructions per row, 70 % of them the two 31-pass bitwise searches (ballot + scalar popcount per pass), 237 us for the
// NACA decoder's 225 k rows x 728 keys at 126 GB/s.  Here a lane owns a row and walks ALL keys of its sample (staged once
// per workgroup in LDS, read as broadcasts), so every wave instruction serves 64 rows and nothing crosses lanes:
//   pass 1   64 block minima (key j belongs to block j % 64): 6 instructions per key;
//   bound    the (k+2)-th smallest block minimum U >= m_(k+1) (k + 2 distinct blocks hold a key <= U): a 64-input sorting
//            network on the lane's own registers; on average only ~2.5 more than k+2 keys are <= U;
//   pass 2   distances again, the keys with m <= U (1 + 2^-20) appended (in key order) to the lane's LDS column;
//   exact    those <= LN_CAPB candidates sorted in registers: m_min, m_(k), m_(k+1); the list = candidates with
//            m <= m_(k+1) (1 + 2^-21), compacted in place, copied out coalesced; per-key counts for the transposed lists
//            through an LDS histogram (one global atomic per key and workgroup - nbr_count_lds's job).
// ~210 wave instructions per row instead of ~650, none of them a dependent cross-lane chain.  Distances are formed by the
// same sq_dist3 as everywhere else, order statistics on their bit patterns: results are bit-identical to plan_rows_reg
// (tests/test_gpu_ops.py, test_gpu_round4.py).  A lane whose candidates overflow its column (massive ties) flags its row
// (count = -1); plan_rows_fix then gives such rows a wavefront each with the old algorithm.
constexpr int LN_NB = 64;               // most blocks a lane keeps (needs rank_k + 2 <= blocks)
constexpr int LN_CAPB = 40;             // candidates per lane (2-byte key indices: 20 KB per workgroup)
__host__ __device__ constexpr int ln_capb(int nb) { return nb < LN_CAPB ? nb : LN_CAPB; }

template <int N, int LOGN>
__device__ __forceinline__ void sort_regs(uint32_t (&v)[N]) {           // bitonic network, every index compile-time
    static_assert((1 << LOGN) == N, "power of two");
#pragma unroll
    for (int lk = 1; lk <= LOGN; ++lk)
#pragma unroll
        for (int lj = lk - 1; lj >= 0; --lj)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int k = 1 << lk, j = 1 << lj, l = i ^ j;
                if (l > i) {
                    const uint32_t lo = min(v[i], v[l]), hi = max(v[i], v[l]);
                    const bool up = (i & k) == 0;
                    v[i] = up ? lo : hi;
                    v[l] = up ? hi : lo;
                }
            }
}
// v[idx] and v[idx + 1] for a wave-uniform idx (a chain of selects on registers)
template <int N>
__device__ __forceinline__ void pick2_reg(const uint32_t (&v)[N], int idx, uint32_t& a0, uint32_t& a1) {
    a0 = v[0]; a1 = v[N - 1];
#pragma unroll
    for (int i = 1; i < N; ++i) {
        a0 = (idx == i) ? v[i] : a0;
        a1 = (idx + 1 == i) ? v[i] : a1;
    }
}

#ifndef PL_G
#define PL_G 6          // keys of a block whose distances are formed together in pass 2 (NACA plan: 8 / 6 / 4 -> 102.8 / 97.5 / 98.5 us)
#endif
// NB blocks (32 when rank_k + 2 <= 16: half the registers and a third of the sorting network; 64 otherwise)
template <bool SD2, bool PER, int NB, int LOGNB>   // SD2: at most two coordinates enter the distance (keys as float2); PER: periodic wrap
__global__ __launch_bounds__(256) void plan_rows_lane(SelectArgs a, int cap, int* __restrict__ nbr_idx,
                                                      int* __restrict__ nbr_cnt, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int KW = SD2 ? 2 : 4;                                     // floats per staged key
    constexpr int CAPB = ln_capb(NB);
    const int npad = (a.n_in + NB - 1) / NB * NB;
    float* keys = reinterpret_cast<float*>(lds_raw);                    // [npad][KW]
    unsigned short* cbuf = reinterpret_cast<unsigned short*>(keys + (long)npad * KW);       // [CAPB][256]
    int* hist = reinterpret_cast<int*>(cbuf + CAPB * 256);              // [n_in] (counts != null)
    const int tid = threadIdx.x;
    const int mb = blockIdx.y, r0 = blockIdx.x * 256;
    const long rows = (long)a.mesh_batch * a.n_out;
    const float* pin = a.mesh_in + (long)mb * a.n_in * a.sdim;
    // ---- stage the sample's keys; the padding keys are infinitely far away (m = +inf: never selected)
    for (int j = tid; j < npad; j += 256) {
        float x = __builtin_inff(), y = 0.0f, z = 0.0f;
        if (j < a.n_in) load_point(pin + (long)j * a.sdim, a.sdim, a.coords_used, x, y, z);
        if (SD2) { keys[2 * j] = x; keys[2 * j + 1] = y; }
        else *reinterpret_cast<float4*>(keys + 4 * j) = make_float4(x, y, z, 0.0f);
    }
    if (counts) for (int j = tid; j < a.n_in; j += 256) hist[j] = 0;
    const int r = r0 + tid;
    const bool valid = r < a.n_out;
    const long row = (long)mb * a.n_out + (valid ? r : a.n_out - 1);
    float ox, oy, oz;
    load_point(a.mesh_out + row * a.sdim, a.sdim, a.coords_used, ox, oy, oz);
    __syncthreads();
    auto dist_bits = [&](int j) -> uint32_t {
        float ix, iy, iz = 0.0f;
        if (SD2) { const float2 q = *reinterpret_cast<const float2*>(keys + 2 * j); ix = q.x; iy = q.y; }
        else { const float4 q = *reinterpret_cast<const float4*>(keys + 4 * j); ix = q.x; iy = q.y; iz = q.z; }
        return __float_as_uint(sq_dist3t<PER>(ox, oy, oz, ix, iy, iz, a.period));
    };
    const int k = a.rank_k;
    uint32_t Ub;
    unsigned long long bmask;                                           // blocks pass 2 visits
    bool no_bound;
    {   // ---- pass 1: block minima, eight keys in flight (the scheduler would otherwise request a whole block row at once)
        uint32_t bmin[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) bmin[u] = 0xFFFFFFFFu;
        for (int j0 = 0; j0 < npad; j0 += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                bmin[u] = min(bmin[u], dist_bits(j0 + u));
                if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the minima are sorted WITH their block index in the low LOGNB bits (the minimum truncated by as many bits: still a
        // lower bound of the block): the sorted prefix below the bound then names the blocks pass 2 has to visit.  (The empty asm
        // statements pin each value in its register: without them hipcc keeps both forms of all 64 minima alive - 185 VGPRs for 95.)
#pragma unroll
        for (int u = 0; u < NB; ++u) { asm volatile("" : "+v"(bmin[u])); bmin[u] = (bmin[u] & ~(uint32_t)(NB - 1)) | (uint32_t)u; asm volatile("" : "+v"(bmin[u])); }
        sort_regs<NB, LOGNB>(bmin);
        uint32_t U, unused;
        pick2_reg<NB>(bmin, k + 1, U, unused);                          // (k+2)-th smallest block minimum >= m_(k+1)
        U |= (uint32_t)(NB - 1);                                        // (at least the untruncated minimum of that block)
        no_bound = U >= 0x7F800000u;                                    // (cannot happen with n_in >= k + 2 real keys)
        Ub = __float_as_uint(__fmul_rn(__uint_as_float(U), 1.00000095367431640625f));     // U (1 + 2^-20)
        bmask = 0ull;
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if ((bmin[u] & ~(uint32_t)(NB - 1)) <= Ub) bmask |= 1ull << (bmin[u] & (uint32_t)(NB - 1));
        if (no_bound) bmask = ~0ull;
    }
    // ---- pass 2: candidates (key order); eight keys' distances are formed before the (divergent) appends.  (A branch-free
    // form - every key written to the lane's next slot, the slot advancing only for a qualifying key - measured slower:
    // 78 vs 70 us, the LDS write traffic outweighs the saved branches.)
    // Only the blocks whose minimum is below the bound can hold candidates (typically 20-30 of the 64): each lane walks ITS
    // blocks (keys u, u + NB, u + 2 NB, ... - per-lane LDS addresses instead of broadcasts), the wavefront loops until the
    // lane with the most blocks is done.  (All keys in key order, as before: 728 distances per row instead of ~330.)
    int cnt = 0;
#pragma unroll 1
    while (__builtin_amdgcn_ballot_w64(bmask != 0ull) != 0ull) {
        const bool live = bmask != 0ull;
        const int u = live ? (int)__builtin_ctzll(bmask) : 0;
        bmask &= bmask - 1ull;
#pragma unroll 1
        for (int j0 = u; j0 < npad; j0 += PL_G * NB) {
            uint32_t m8[PL_G];
#pragma unroll
            for (int t = 0; t < PL_G; ++t) m8[t] = dist_bits(min(j0 + t * NB, npad - 1));
#pragma unroll
            for (int t = 0; t < PL_G; ++t) {
                if (live && j0 + t * NB < npad && m8[t] <= Ub) {
                    if (cnt < CAPB) cbuf[cnt * 256 + tid] = (unsigned short)(j0 + t * NB);
                    ++cnt;
                }
            }
        }
    }
    const bool overflow = cnt > CAPB || no_bound;
    // ---- exact order statistics among the candidates (all reads of a group of eight requested before they are used: the
    // slots beyond cnt hold stale indices - clamped - and count as +inf)
    uint32_t kmin, vk, vk1;
    {
        uint32_t cm[NB];
#pragma unroll
        for (int g0 = 0; g0 < NB; g0 += 8) {
            int jj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s2 = g0 + u;
                jj[u] = (s2 < CAPB) ? (int)cbuf[(s2 < CAPB ? s2 : 0) * 256 + tid] : 0;
                jj[u] = (s2 < cnt && jj[u] < npad) ? jj[u] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s2 = g0 + u;
                const uint32_t m = dist_bits(jj[u]);
                cm[s2] = (s2 < CAPB && s2 < cnt) ? m : 0xFFFFFFFFu;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        sort_regs<NB, LOGNB>(cm);
        kmin = cm[0];
        pick2_reg<NB>(cm, k, vk, vk1);
        if (k + 1 > a.n_in - 1) vk1 = vk;
    }
    // ---- the list: candidates with m <= m_(k+1) (1 + 2^-21), compacted in place (still in key order: a slot is only ever
    // written at or below the slot being read)
    const float bound = __uint_as_float(vk1) * 1.00000047683715820312f;
    int total = 0;
    if (!overflow) {
#pragma unroll
        for (int g0 = 0; g0 < CAPB; g0 += 8) {
            if (g0 < cnt) {
                int jj[8];
                uint32_t mm[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    jj[u] = (int)cbuf[(g0 + u) * 256 + tid];
                    jj[u] = (g0 + u < cnt && jj[u] < npad) ? jj[u] : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) mm[u] = dist_bits(jj[u]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (g0 + u < cnt && __uint_as_float(mm[u]) <= bound) {
                        cbuf[total * 256 + tid] = (unsigned short)jj[u];
                        if (counts && valid && total < cap) atomicAdd(&hist[jj[u]], 1);
                        ++total;
                    }
                }
            }
        }
    }
    if (valid) {
        if (!overflow) {
            a.stats[row] = __uint_as_float(vk);
            a.stats[rows + row] = __uint_as_float(vk1);
            a.stats[2 * rows + row] = __uint_as_float(kmin);
        }
        nbr_cnt[row] = overflow ? -1 : total;                          // -1: plan_rows_fix redoes this row
    }
    // (a list longer than cap keeps its true count and is truncated, as plan_rows_reg does)
    __syncthreads();
    const int nrows = min(256, a.n_out - r0);
    int* out = nbr_idx + ((long)mb * a.n_out + r0) * cap;
    const int ncopy = min(cap, CAPB);
    for (int e = tid; e < nrows * ncopy; e += 256) {
        const int rr = e / ncopy, sl = e - rr * ncopy;
        out[(long)rr * cap + sl] = (int)cbuf[sl * 256 + rr];
    }
    if (counts)
        for (int j = tid; j < a.n_in; j += 256)
            if (hist[j]) atomicAdd(counts + (long)mb * a.n_in + j, hist[j]);
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

extern "C" int pit_lists_transpose(const int* nbr_idx, const int* nbr_cnt, int mesh_batch, int n_out, int n_in, int cap,
                                   int* rev_ptr, int* rev_row, int* workspace, void* stream) {
    if (!nbr_idx || !nbr_cnt || !rev_ptr || !rev_row || !workspace) return PIT_ERR_NULL;
    if (mesh_batch <= 0 || n_out <= 0 || n_in <= 0 || cap <= 0) return PIT_ERR_SIZE;
    if (n_in > NBR_LDS_KEYS) return PIT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int* counts = workspace;
    int* cursor = workspace + (long)mesh_batch * n_in;
    fill_int(counts, (long)mesh_batch * n_in, 0, s);
    fill_int(rev_row, (long)mesh_batch * n_out * cap, -1, s);
    return launch_transpose(nbr_idx, nbr_cnt, mesh_batch, n_out, n_in, cap, rev_ptr, rev_row, counts, cursor, false, s);
}

extern "C" int pit_plan_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                            int space_dim, int metric, float period, int rank_k, float* stats, int cap,
                            int* nbr_idx, int* nbr_cnt, int* rev_ptr, int* rev_row, int* workspace, int flags, void* stream) {
    if (!mesh_out || !mesh_in || !stats || !nbr_idx || !nbr_cnt) return PIT_ERR_NULL;
    if (rev_ptr && (!rev_row || !workspace)) return PIT_ERR_NULL;
    if (mesh_batch <= 0 || n_out <= 0 || n_in <= 0 || space_dim < 1 || space_dim > 3 || cap <= 0) return PIT_ERR_SIZE;
    if (metric < PIT_METRIC_EUCLID || metric > PIT_METRIC_PERIODIC2D) return PIT_ERR_METRIC;
    if (rank_k < 0 || rank_k > n_in - 1) return PIT_ERR_SIZE;
    const int items = (n_in + 63) / 64;
    if (items > 64 || (flags & PIT_PLAN_TWO_PASSES)) {        // long rows: the two streaming passes
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
    // per-sample meshes (rebuilt every step), short rows: one row per lane (plan_rows_lane); rows it cannot hold in a lane's
    // column (massive ties) are redone by plan_rows_fix.  Batch-free meshes (one-off, cached plans; regular grids with
    // large tie shells) keep the wave-per-row kernel.
    // (a lane per row needs rows: 64 per wave - below ~500 waves the wave-per-row kernel fills the chip better: Elasticity, 9 720 rows,
    // 2.59 vs 2.45 ms per step)
    const bool lane_ok = mesh_batch > 1 && rank_k + 2 <= LN_NB && n_in <= NBR_LDS_KEYS && n_in < 65536 && rows >= 32768 &&
                         (!rev_ptr || agg) && !(flags & PIT_PLAN_WAVE_PER_ROW);
    const bool sd2 = a.coords_used <= 2;
    // (dynamic LDS of the lane kernel: the sample's keys + the lanes' columns (+ the per-key histogram): 3-d meshes of 2 816 ..
    // 4 096 keys would need 66-102 KB - beyond the 64 KB a launch gets without raising the kernel's limit: those keep the wave-per-row kernel)
    const size_t lane_sm = (size_t)((n_in + 63) / 64 * 64) * (sd2 ? 2 : 4) * sizeof(float) + (size_t)ln_capb(64) * 256 * sizeof(unsigned short) +
                           (rev_ptr ? (size_t)n_in * sizeof(int) : 0);
    if (lane_ok && lane_sm <= 65536) {
        const int nb = 64;       // (32 blocks: a third of the sorting network, but 144 VGPRs against 95 and a looser bound - 0.2 % of the NACA rows then
                                 // overflow a 32-slot column and the clustered repairs cost 30 us: 64 blocks measured faster everywhere)
        const int npad = (n_in + nb - 1) / nb * nb;
        const size_t sm = (size_t)npad * (sd2 ? 2 : 4) * sizeof(float) + (size_t)ln_capb(nb) * 256 * sizeof(unsigned short) +
                          (rev_ptr ? (size_t)n_in * sizeof(int) : 0);
        const dim3 lgrid((unsigned)((n_out + 255) / 256), (unsigned)mesh_batch);
        int* lcounts = rev_ptr ? counts : nullptr;
#define PIT_LANE(SD_, PER_, NB_, LG_) hipLaunchKernelGGL((plan_rows_lane<SD_, PER_, NB_, LG_>), lgrid, block, sm, s, a, cap, nbr_idx, nbr_cnt, lcounts)
#define PIT_LANE_P(SD_, NB_, LG_) do { if (a.periodic) PIT_LANE(SD_, true, NB_, LG_); else PIT_LANE(SD_, false, NB_, LG_); } while (0)
        if (sd2) { if (nb == 32) PIT_LANE_P(true, 32, 5); else PIT_LANE_P(true, 64, 6); }
        else { if (nb == 32) PIT_LANE_P(false, 32, 5); else PIT_LANE_P(false, 64, 6); }
#undef PIT_LANE_P
#undef PIT_LANE
        PIT_CHECK_LAUNCH();
        const dim3 fgrid(256);
#define PIT_FIX(I_) hipLaunchKernelGGL(plan_rows_fix<I_>, fgrid, block, 0, s, a, cap, nbr_idx, nbr_cnt, rev_ptr ? counts : nullptr)
        if (items <= 4) PIT_FIX(4);
        else if (items <= 16) PIT_FIX(16);
        else PIT_FIX(64);
#undef PIT_FIX
        PIT_CHECK_LAUNCH();
        if (rev_ptr) return launch_transpose(nbr_idx, nbr_cnt, mesh_batch, n_out, n_in, cap, rev_ptr, rev_row, counts, cursor, true, s);
        return 0;
    }
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
