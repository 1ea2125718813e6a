// Register-direct GEMM body (v_mfma_f32_32x32x2_f32, in-workgroup split-K, reduce-scatter epilogue) shared by the
// MLP kernels (pit_mlp.hip) and by the attention-backward launch that carries an MLP's weight-gradient reductions
// along (pit_posatt.hip: posatt_bwd_pair_dw_kernel).  See pit_mlp.hip for the design notes.
#pragma once
#include "pit_common.h"
#include <type_traits>

// (timing experiment -DPIT_RR_NO_ATOMICS: the split-K partial tiles leave as plain stores - results void)
#ifdef PIT_RR_NO_ATOMICS
#define PIT_RR_ADD(p_, v_) (*(p_) = (v_))
#else
#define PIT_RR_ADD(p_, v_) atomicAdd((p_), (v_))
#endif

namespace pit_detail {

struct GemmArgs {
    const float* A; long a_rs, a_cs;     // A(m,k) = A[m*a_rs + k*a_cs]
    const float* B; long b_rs, b_cs;     // B(k,n) = B[k*b_rs + n*b_cs]
    int M, N, K;
    unsigned a_bytes, b_bytes;           // extents for the raw-buffer descriptors
    int a_vec, b_vec;                    // 16-B fragment loads legal (unit k stride, aligned rows)
    int k_slab;                          // K range per blockIdx.z
    const float* a_gz;                   // optional prologue: A(m,k) *= gelu'(a_gz[same index as A])
    float* a_out; long a_out_rs, a_out_cs;  // optional: write the prologue result (blockIdx.x == 0 only)
    const float* bias;                   // [N] or null
    float* C; long ldc;
    float* Z; long ldz;                  // optional pre-activation copy
    const float* G; long ldg;            // optional: multiply result by gelu'(G[m,n])
    int epi;                             // EPI_* epilogue flavour
    int atomic;                          // 1: row-reducing GEMM, also split over blockIdx.z
    int ones_col;                        // >= 0: B(k, ones_col) == 1, that output column goes to C2[m]
    float* C2;
    int bf16;                            // PIT_MATH_BF16: one v_mfma_f32_32x32x8_bf16 per 4 k instead of 4 fp32 MFMAs
    int seq_epi;                         // experiments (PIT_GEMM_RD_SEQ_EPI): one accumulator register per epilogue trip
    // bf16 STORAGE (PIT_IO_*; gemm_bfl_kernel and the thin kernels only): the tensor behind A / B / C / Z / G holds bf16
    // elements; strides stay in elements
    int a16, b16, c16, z16, g16;
    // XCD-aware tile order of the LDS-staged kernels (non-atomic kinds): a 1-D grid whose workgroup id -> (column block,
    // row tile) map puts all column blocks of a row tile on ONE XCD (ids are dealt round-robin to the 8 XCDs), so the A
    // rows are fetched from HBM once per tile instead of once per column block; 0 = the plain (x, y) grid
    int remap_gx, remap_gy;
};

// both weight-gradient reductions of one MLP (dW2|db2 and dW1|db1), prepared for gemm_rd_body<1, EPI_ATOMIC>:
// workgroups [0, n1) run g1 on a (gx1, gy1, .) grid, [n1, n1 + n2) run g2
// rr1 / rr2 (pit_block.hip's riders): that reduction runs as gemm_rr_tile tiles - workgroup id -> (slab = id / tiles, tile = id % tiles),
// slab s = chunks [s nchunks / slabs, (s+1) nchunks / slabs) of RR_BK rows
struct DwPair { GemmArgs g1, g2; int n1, n2, gx1, gy1, gx2, gy2; int rr1, rr2, tx1, tiles1, slabs1, tx2, tiles2, slabs2, nchunks; };
constexpr int RR_BK = 64;

// fills `out` and returns true when the job is small enough to ride along in another launch (pit_mlp.hip);
// `waves` = waves per workgroup of the carrying launch (the row slabs are sized so that a wave reduces as many rows
// as in the reductions' own 8-wave launch)
// `target_wgs`: workgroups each of the two reductions aims for (its K range is split accordingly; 768 = the reductions' own
// launch)
// `allow_rr`: the caller's kernel runs rr1 / rr2 reductions through gemm_rr_tile (>= 64 KiB of LDS, >= 4 waves)
bool plan_dw_pair(const pit_mlp_params_job& job, int waves, DwPair* out, int target_wgs = 768, bool allow_rr = false);
// both reductions as gemm_rr_tile tiles (fp32 or - bf16 math mode - bf16 MFMA operands) for a carrying launch that has 64 KiB of LDS
// and at least four waves (pit_satt.hip's backward launch): any size, `budget_wgs` workgroups in all
bool plan_rr_rider(const pit_mlp_params_job& job, DwPair* out, int budget_wgs);

}  // namespace pit_detail

namespace {

using pit_detail::GemmArgs;

// 4 k-values of one operand for this lane: X(i, kk+e), e = 0..3, through a raw buffer
// (invalid i / k -> offset out of range -> 0).  VEC: one 16-B load (k is the unit-stride axis
// and rows are 16-B aligned); FULL: the 4 k are known to be in range.
template <bool FULL>
__device__ __forceinline__ void load_frag(__amdgpu_buffer_rsrc_t r, unsigned bytes, unsigned ibase, bool ivalid,
                                          unsigned kstride4, bool vec, int kk, int kend, float (&v)[4]) {
    if (FULL && vec) {
        buf_load4(r, ivalid ? ibase + (unsigned)kk * 4u : bytes, v);
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = ivalid && (FULL || kk + e < kend);
        v[e] = buf_load(r, ok ? ibase + (unsigned)(kk + e) * kstride4 : bytes);
    }
}

// epilogue flavours (compile-time, so the per-element code is branch-free)
constexpr int EPI_STORE = 0;      // C = acc
constexpr int EPI_BIAS = 1;       // C = acc + bias
constexpr int EPI_BIAS_GELU = 2;  // Z = acc + bias; C = gelu(Z)
constexpr int EPI_MUL_GELU_GRAD = 3;  // C = acc * gelu'(G)
constexpr int EPI_ATOMIC = 4;     // C += acc (fp32 atomics), ones column -> C2

template <int TN, int EPI>
__device__ __forceinline__ void gemm_rd_body(const GemmArgs& g, int bx, int by, int bz) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = by * 32, n0 = bx * 32 * TN;
    const int kbeg = bz * g.k_slab;
    const int kend = min(g.K, kbeg + g.k_slab);
    // this wave's k range: multiples of 8
    const int span = kend - kbeg;
    const int per_wave = ((span + nwaves * 8 - 1) / (nwaves * 8)) * 8;
    const int wk0 = kbeg + wave * per_wave;
    const int wk1 = min(kend, wk0 + per_wave);

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, g.a_bytes);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(g.a_gz ? g.a_gz : g.A, g.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(g.B, g.b_bytes);
    const int m = m0 + l31;
    const bool mvalid = m < g.M;
    const unsigned abase = (unsigned)m * (unsigned)g.a_rs * 4u;
    const unsigned akstride = (unsigned)g.a_cs * 4u, bkstride = (unsigned)g.b_rs * 4u;
    int ncol[TN];
    bool nvalid[TN], nones[TN];
    unsigned bbase[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        ncol[t] = n0 + t * 32 + l31;
        nones[t] = (EPI == EPI_ATOMIC) && ncol[t] == g.ones_col;
        nvalid[t] = ncol[t] < g.N && !nones[t];
        bbase[t] = (unsigned)ncol[t] * (unsigned)g.b_cs * 4u;
    }

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    auto step = [&](int k0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int kk = k0 + 4 * half;            // this half-wave's 4 k values
        float av[4], bv[TN][4];
        load_frag<FULL>(ra, g.a_bytes, abase, mvalid, akstride, g.a_vec != 0, kk, wk1, av);
        if (g.a_gz) {
            float zv[4];
            load_frag<FULL>(rz, g.a_bytes, abase, mvalid, akstride, g.a_vec != 0, kk, wk1, zv);
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] *= gelu_erf_grad(zv[e]);
            if (g.a_out && bx == 0 && mvalid) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (FULL || kk + e < wk1) g.a_out[(long)m * g.a_out_rs + (long)(kk + e) * g.a_out_cs] = av[e];
            }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            load_frag<FULL>(rb, g.b_bytes, bbase[t], nvalid[t], bkstride, g.b_vec != 0, kk, wk1, bv[t]);
            if (EPI == EPI_ATOMIC && nones[t]) {      // virtual all-ones column (bias gradient)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[t][e] = (FULL || kk + e < wk1) ? 1.0f : 0.0f;
            }
        }
        if (g.bf16) {                            // the lane's 4 consecutive k are exactly one bf16 fragment
            const bf16x4 ap = pack_bf16(av[0], av[1], av[2], av[3]);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                acc[t] = mfma_32x32x8_bf16(ap, pack_bf16(bv[t][0], bv[t][1], bv[t][2], bv[t][3]), acc[t]);
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[t] = mfma_32x32x2(av[e], bv[t][e], acc[t]);
    };

    int k0 = wk0;
    for (; k0 + 16 <= wk1; k0 += 16) {           // two steps per trip: 4+ fragment loads in flight
        step(k0, std::true_type{});
        step(k0 + 8, std::true_type{});
    }
    if (k0 + 8 <= wk1) { step(k0, std::true_type{}); k0 += 8; }
    if (k0 < wk1) step(k0, std::false_type{});

    // ---- in-workgroup split-K reduction as a reduce-scatter through LDS: every wave parks its
    // partial tile, then each wave sums and finishes its own share of the 16*TN accumulator
    // registers, so the (erf-heavy) epilogue is spread over all waves and costs one barrier.
    constexpr int NQ = TN * 16;
    const int share = NQ / nwaves;               // nwaves in {1,2,4,8} divides 16
    {
        float* dst = red + (long)wave * NQ * 64 + lane;
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64] = acc[t][r];
        __syncthreads();
    }
    const int q0 = wave * share;
    // U accumulator registers per trip: their LDS reads (U x nwaves) and the epilogue's global loads (bias, the
    // gelu' argument) are all requested before the first add - one register per trip with a runtime wave loop
    // made every trip a chain of dependent LDS and memory round trips
    auto finish = [&](auto u_tag, auto nw_tag, int qfirst) {
        constexpr int U = decltype(u_tag)::value, NW = decltype(nw_tag)::value;
        float part[U][NW], aux[U], bia[U];
        int col[U], row[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int w = 0; w < NW; ++w) part[u][w] = red[((long)w * NQ + qfirst + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = qfirst + u;
            const int t = q >> 4, r = q & 15;
            col[u] = n0 + t * 32 + l31;
            row[u] = m0 + acc_row(r, half);
            ok[u] = col[u] < g.N && row[u] < g.M;
            bia[u] = 0.0f; aux[u] = 0.0f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bia[u] = ok[u] ? g.bias[col[u]] : 0.0f;
            if (EPI == EPI_MUL_GELU_GRAD) aux[u] = ok[u] ? g.G[(long)row[u] * g.ldg + col[u]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += part[u][w];
            if (!ok[u]) continue;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) v += bia[u];
            if (EPI == EPI_BIAS_GELU) { g.Z[(long)row[u] * g.ldz + col[u]] = v; v = gelu_erf(v); }
            if (EPI == EPI_MUL_GELU_GRAD) v *= gelu_erf_grad(aux[u]);
            if (EPI == EPI_ATOMIC) {
                if (col[u] == g.ones_col) PIT_RR_ADD(g.C2 + row[u], v);
                else PIT_RR_ADD(g.C + (long)row[u] * g.ldc + col[u], v);
            } else {
                g.C[(long)row[u] * g.ldc + col[u]] = v;
            }
        }
    };
    auto finish_all = [&](auto nw_tag) {
        int qq = 0;
        if (EPI != EPI_ATOMIC && !g.seq_epi) {       // (the row-reducing GEMMs end in atomics, nothing to wait for: batching them measured slower)
            for (; qq + 4 <= share; qq += 4) finish(std::integral_constant<int, 4>{}, nw_tag, q0 + qq);
            for (; qq + 2 <= share; qq += 2) finish(std::integral_constant<int, 2>{}, nw_tag, q0 + qq);
        }
        for (; qq < share; ++qq) finish(std::integral_constant<int, 1>{}, nw_tag, q0 + qq);
    };
    switch (nwaves) {
        case 2: finish_all(std::integral_constant<int, 2>{}); break;
        case 4: finish_all(std::integral_constant<int, 4>{}); break;
        default: finish_all(std::integral_constant<int, 8>{}); break;
    }
}



// ------------------------------------------------------------------------------------
// One tile of the row-reducing GEMM of pit_mlp.hip (gemm_rr_kernel; design notes there): C[m][n] += sum over k in
// [kbeg, kend) of A[k][m] B[k][n] for the (64 RM) x (64 RN) tile (bx, by), by the FIRST FOUR waves of the calling workgroup
// (callers with more waves let the others leave before the call: ended waves do not take part in the barriers).
// As_ / Bs_: 2 x BK x 64 RM and 2 x BK x 64 RN floats of LDS.  Shared with the fused block backward (pit_block.hip), whose
// riders are these tiles.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// SC1A: the A operand was stored by other workgroups of THIS launch (the persistent latent backward): sc1 loads
template <int RM, int RN, int BK, bool BF, bool SC1A = false>
__device__ __forceinline__ void gemm_rr_tile(const GemmArgs& g, int bx, int by, int kbeg, int kend, float* As_, float* Bs_) {
    constexpr int BM = 64 * RM, BN = 64 * RN;
    constexpr int PA = BK * BM / 1024, PB = BK * BN / 1024;         // 16-B loads per thread and chunk
    constexpr int NA = (RM * RN == 1) ? 2 : 1;     // a lone tile alternates between two accumulators: no dependent MFMA chain
    static_assert(PA >= 1 && PB >= 1 && BK % (BF ? 32 : 4) == 0, "chunk shape");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = by * BM, n0 = bx * BN;
    const int wm = (wave & 1) * 32 * RM, wn = (wave >> 1) * 32 * RN;
    const int n_real = g.ones_col >= 0 ? g.N - 1 : g.N;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, g.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(g.B, g.b_bytes);

    float sa[PA][4], sb[PB][4];
    auto gload = [&](int kc) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int q = p * 256 + tid, k = kc + q / (BM / 4), m = m0 + (q % (BM / 4)) * 4;
            const bool ok = k < kend && m < g.M;
            if constexpr (SC1A) buf_load4_sc1(ra, ok ? ((unsigned)k * (unsigned)g.a_cs + (unsigned)m) * 4u : g.a_bytes, sa[p]);
            else buf_load4(ra, ok ? ((unsigned)k * (unsigned)g.a_cs + (unsigned)m) * 4u : g.a_bytes, sa[p]);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int q = p * 256 + tid, k = kc + q / (BN / 4), n = n0 + (q % (BN / 4)) * 4;
            const bool ok = k < kend && n < n_real;
            buf_load4(rb, ok ? ((unsigned)k * (unsigned)g.b_rs + (unsigned)n) * 4u : g.b_bytes, sb[p]);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PA; ++p)
            *reinterpret_cast<float4*>(&As_[buf * BK * BM + (p * 256 + tid) * 4]) = make_float4(sa[p][0], sa[p][1], sa[p][2], sa[p][3]);
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<float4*>(&Bs_[buf * BK * BN + (p * 256 + tid) * 4]) = make_float4(sb[p][0], sb[p][1], sb[p][2], sb[p][3]);
    };

    f32x16 acc[RM][RN][NA];
#pragma unroll
    for (int t = 0; t < RM; ++t)
#pragma unroll
        for (int u = 0; u < RN; ++u)
#pragma unroll
            for (int x = 0; x < NA; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][x][r] = 0.0f;
    const bool want_rowsum = g.ones_col >= 0 && bx == 0 && wn == 0;
    float rsum[RM];
#pragma unroll
    for (int t = 0; t < RM; ++t) rsum[t] = 0.0f;
    gload(kbeg);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kc = kbeg; kc < kend; kc += BK) {
        const bool more = kc + BK < kend;
        if (more) gload(kc + BK);
        if constexpr (BF) {
            const float* as = &As_[cur * BK * BM + 8 * half * BM + wm + RM * l31];      // this half-wave's k of a step: 16 st + 8 half + e
            const float* bs = &Bs_[cur * BK * BN + 8 * half * BN + wn + l31];
#pragma unroll
            for (int st = 0; st < BK / 16; ++st) {
                bf16x8_t a8[RM], b8[RN];
#pragma unroll
                for (int t = 0; t < RM; ++t) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = as[(16 * st + e) * BM + t];
                    if (want_rowsum) rsum[t] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    const uint4 q = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
                    a8[t] = __builtin_bit_cast(bf16x8_t, q);
                }
#pragma unroll
                for (int u = 0; u < RN; ++u) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bs[(16 * st + e) * BN + 32 * u];
                    const uint4 q = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
                    b8[u] = __builtin_bit_cast(bf16x8_t, q);
                }
#pragma unroll
                for (int t = 0; t < RM; ++t)
#pragma unroll
                    for (int u = 0; u < RN; ++u) acc[t][u][st % NA] = mfma_32x32x16_bf16(a8[t], b8[u], acc[t][u][st % NA]);
            }
        } else {
        const float* as = &As_[cur * BK * BM + half * BM + wm + RM * l31];          // this half-wave's k of a step: 2 st + half
        const float* bs = &Bs_[cur * BK * BN + half * BN + wn + l31];
        float av[2][RM], bv[2][RN];
        auto fetch = [&](int st, float (&a_)[RM], float (&b_)[RN]) {
#pragma unroll
            for (int t = 0; t < RM; ++t) a_[t] = as[2 * st * BM + t];
#pragma unroll
            for (int u = 0; u < RN; ++u) b_[u] = bs[2 * st * BN + 32 * u];
        };
        fetch(0, av[0], bv[0]);
#pragma unroll
        for (int st = 0; st < BK / 2; ++st) {
            if (st + 1 < BK / 2) fetch(st + 1, av[(st + 1) & 1], bv[(st + 1) & 1]);
            if (want_rowsum) {
#pragma unroll
                for (int t = 0; t < RM; ++t) rsum[t] += av[st & 1][t];
            }
#pragma unroll
            for (int t = 0; t < RM; ++t)
#pragma unroll
                for (int u = 0; u < RN; ++u)
                    acc[t][u][st % NA] = mfma_32x32x2(av[st & 1][t], bv[st & 1][u], acc[t][u][st % NA]);
        }
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    if (want_rowsum) {
#pragma unroll
        for (int t = 0; t < RM; ++t) {
            const float v = rsum[t] + __shfl_xor(rsum[t], 32);
            const int row = m0 + wm + RM * l31 + t;
            if (half == 0 && row < g.M) PIT_RR_ADD(g.C2 + row, v);
        }
    }
#pragma unroll
    for (int t = 0; t < RM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + RM * acc_row(r, half) + t;
            if (row >= g.M) continue;
#pragma unroll
            for (int u = 0; u < RN; ++u) {
                const int col = n0 + wn + 32 * u + l31;
                const float v = NA == 2 ? acc[t][u][0][r] + acc[t][u][NA - 1][r] : acc[t][u][0][r];
                if (col < n_real) PIT_RR_ADD(g.C + (long)row * g.ldc + col, v);
            }
        }
}

// the launch's dynamic LDS (every `extern __shared__` array of a kernel names the same base)
__device__ __forceinline__ float* pit_dyn_smem() {
    extern __shared__ __attribute__((aligned(16))) float pit_dyn_lds[];
    return pit_dyn_lds;
}
// one reduction of a rider as a gemm_rr_tile tile (DwPair::rr1 / rr2): the first four waves, 64 KiB of the launch's LDS
__device__ __forceinline__ void rr_rider(const pit_detail::GemmArgs& g, int id, int tx, int tiles, int slabs, int nchunks, float* smem) {
    if (threadIdx.x >= 256) return;                       // (ended waves do not take part in the tile's barriers)
    const int slab = id / tiles, tile = id % tiles;
    const int kbeg = (int)((long)slab * nchunks / slabs) * pit_detail::RR_BK;
    const int kend = min(g.K, (int)((long)(slab + 1) * nchunks / slabs) * pit_detail::RR_BK);
    gemm_rr_tile<1, 1, pit_detail::RR_BK, false>(g, tile % tx, tile / tx, kbeg, kend, smem, smem + 2 * pit_detail::RR_BK * 64);
}
// workgroup `id` of a plan_rr_rider pair (both reductions are tiles; bf16 MFMA operands when the job's math mode says so)
__device__ __forceinline__ void rr_rider_pair(const pit_detail::DwPair& w, int id, float* smem) {
    if (threadIdx.x >= 256) return;
    const bool second = id >= w.n1;
    if (second) id -= w.n1;
    const GemmArgs& g = second ? w.g2 : w.g1;
    const int tx = second ? w.tx2 : w.tx1, tiles = second ? w.tiles2 : w.tiles1, slabs = second ? w.slabs2 : w.slabs1;
    const int slab = id / tiles, tile = id % tiles;
    const int kbeg = (int)((long)slab * w.nchunks / slabs) * pit_detail::RR_BK;
    const int kend = min(g.K, (int)((long)(slab + 1) * w.nchunks / slabs) * pit_detail::RR_BK);
    if (g.bf16) gemm_rr_tile<1, 1, pit_detail::RR_BK, true>(g, tile % tx, tile / tx, kbeg, kend, smem, smem + 2 * pit_detail::RR_BK * 64);
    else gemm_rr_tile<1, 1, pit_detail::RR_BK, false>(g, tile % tx, tile / tx, kbeg, kend, smem, smem + 2 * pit_detail::RR_BK * 64);
}
// workgroup `id` of a carried pair of weight-gradient reductions; `smem`: the launch's dynamic LDS (64 KiB when the plan
// allowed rr reductions, see plan_dw_pair)
// (ONE instance of each reduction body per call site: the job's arguments are selected first.  With a body per reduction -
// and block_bwd_kernel calling this for two pairs - that kernel carried eight inlined reduction bodies, 14.8 k instructions)
__device__ __forceinline__ void dw_pair_body(const pit_detail::DwPair& w, int id, float* smem) {
    const bool second = id >= w.n1;
    if (second) id -= w.n1;
    const GemmArgs& g = second ? w.g2 : w.g1;
    const int rr = second ? w.rr2 : w.rr1, tx = second ? w.tx2 : w.tx1, tiles = second ? w.tiles2 : w.tiles1;
    const int slabs = second ? w.slabs2 : w.slabs1, gx = second ? w.gx2 : w.gx1, gy = second ? w.gy2 : w.gy1;
    if (rr) rr_rider(g, id, tx, tiles, slabs, w.nchunks, smem);
    else gemm_rd_body<1, EPI_ATOMIC>(g, id % gx, (id / gx) % gy, id / (gx * gy));
}


}  // namespace
