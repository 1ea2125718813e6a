// One-launch kaiming_mlp chains for the mid-sized processors in the bf16 math mode (round 6; pit.py:21-26 + the gelu of
// pit.py:111,121): hid 128 / 256 on a few thousand rows (Vorticity 5 120 x 768 -> 256 -> 256, Elasticity 9 720, NACA 14 560 x
// 256 -> 128 -> 128).
//
// Such an MLP ran as two GEMM launches forward (12.7 us each at Vorticity: 2.5 % MFMA busy, 45 vector instructions per MFMA - the
// fp32 activations are rounded in-kernel, one round trip per 64-k chunk, 160-320 workgroups that each pull their weights through
// the L1) and three launches for the backward data path (gelu' pass 5.4 + dZ1 15.4 + dX 12.0 us): 58 of a block's 122 us.  Here
// each direction is ONE launch on 32-row (or 48-row: chain_slab_rows) slabs:
//   mlp_chain_fwd_kernel   X slab -> LDS (bf16) once; W1 | W2 - bf16 copies formed once per weight version by the host side -
//                          stream through LDS as 64-k panels, double-buffered, one barrier per panel, the next panel's loads in
//                          flight under the current panel's v_mfma_f32_16x16x32_bf16; Z1 / H / Z2 / Y leave from the accumulators
//                          (H also stays in LDS as GEMM2's A operand).
//   mlp_chain_bwd_kernel   dZ2 = dY gelu'(Z2) -> LDS; dZ1 = (dZ2 W2) gelu'(Z1) -> LDS; dX = dZ1 W1 in n0 / n1 column chunks.  The
//                          weights are the SAME row-major bf16 copies read as [k][n] images through ds_read_b64_tr_b16.
// A workgroup's 512 KB of weights (hid 256) at the ~70 GB/s a CU pulls from L2 is the launch: ~8 us, whatever the slab height.
// gelu / gelu' are the polynomial CDF of pit_common.h (the mode's operands are bf16: 8 bits): the fp32 mode never comes here.
#include "pit_common.h"

namespace {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));

// rows per workgroup: the CR template parameter, 32 or 48 (chain_slab_rows below)
constexpr int PADE = 8;      // pad (bf16 elements) of every LDS row: 16 bytes
constexpr int NT = 512;      // threads: eight waves, wave w owns rows x columns [w N1 / 8, (w + 1) N1 / 8)
constexpr int NSETS = 4;     // weight panels in flight in registers (first version: one, 256 threads - every panel waited a full L2
                             // round trip behind 16 MFMAs: 27.7 / 23.6 us per launch where the two-GEMM path took 25.4 / 32.8)

#ifndef PIT_CHAIN_DBG
#define PIT_CHAIN_DBG 0      // diagnostic builds of mlp_chain_fwd_kernel: 1 no X slab loads, 2 no output stores, 4 no MFMAs, 8 no panel loads
                             // in the loop (results are void, times are not)
#endif
struct ChainArgs {
    int rows, n0, n1;
    const float* x; long ldx;
    const unsigned short *w1b, *w2b;          // bf16 copies: (n1, n0) and (n1, n1), row-major
    const float *b1, *b2;
    float *z1, *h, *z2, *y; long ldy;
    const float* d_y; long ld_dy;
    const float *z1r, *z2r;
    float* d_x; long ld_dx;
    float *dz1, *dz2;
    unsigned short* y16;                       // forward, optional: bf16(y), (rows, n1) - the next layer's pit_satt_fwd reads it (no prep launch)
    unsigned short* g16; const float* rowstat; int pts, mesh_batch;      // backward, optional: what pit_satt_bwd's prep would form from d_x
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void* p) { return make_rsrc(p, 0x7ffffff0u); }
constexpr unsigned OOB = 0x7ffffff8u;
__device__ __forceinline__ float4 ldg4_if(const float* p, long i, bool ok) {
    float v[4];
    buf_load4(wide_rsrc(p), ok ? (unsigned)(i * 4) : OOB, v);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 pack4(float4 v) {
    uint2 pk;
    pk.x = (unsigned)f_to_bf16(v.x) | ((unsigned)f_to_bf16(v.y) << 16);
    pk.y = (unsigned)f_to_bf16(v.z) | ((unsigned)f_to_bf16(v.w) << 16);
    return pk;
}
// eight bf16 of row `row` starting at k0 of a [row][k] image (one ds_read_b128)
__device__ __forceinline__ v8s_t frag_row(const unsigned short* t, int pitch, int row, int k0) {
    return *reinterpret_cast<const v8s_t*>(t + row * pitch + k0);
}
// the lane's eight k's (k0 + 8 kq ..) of column n0 + l15 of a [k][n] image: two ds_read_b64_tr_b16 (cdna guide T10; EXEC all ones)
__device__ __forceinline__ v8s_t frag_tr(const unsigned short* t, int pitch, int k0, int n0, int l15, int kq) {
    const unsigned short* a0 = t + (k0 + 8 * kq + (l15 >> 2)) * pitch + n0 + 4 * (l15 & 3);
    typedef v4s_t __attribute__((address_space(3))) * lds_v4;
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0 + 4 * pitch));
    v8s_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
__device__ __forceinline__ f32x4 mma(v8s_t a, v8s_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, a), __builtin_bit_cast(v8bf_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ float gelu_f(float t) { float e; return t * normal_cdf_fast(t, e); }
__device__ __forceinline__ float gelu_grad_f(float t) { float e; const float c = normal_cdf_fast(t, e); return fmaf(t * 0.39894228040143267794f, e, c); }

// A weight panel as 16-byte pieces (8 bf16) in registers: NPC pieces per thread.  (ext_vector_type, not HIP's uint4 struct: an
// array of the struct type that is filled from memory and consumed behind a barrier is kept in scratch by hipcc - DESIGN section 4.)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int NPC> struct Panel { u32x4_t v[NPC]; };
template <int N1> struct Kp { static constexpr int v = N1 / 4; };     // k per weight panel: 64 (hid 256), 32 (hid 128)

// forward panels: N1 rows (output neurons) x KP k's of w (row stride ldw elements, k offset k0): image [n][KP + 8]
template <int N1>
__device__ __forceinline__ void fwd_panel_load(const unsigned short* w, int ldw, int k0, int tid, Panel<N1 * Kp<N1>::v / 8 / NT>& p) {
    constexpr int KP = Kp<N1>::v;
#pragma unroll
    for (int u = 0; u < N1 * KP / 8 / NT; ++u) {
        const int e = tid + NT * u, n = e / (KP / 8), kc = e % (KP / 8);
        p.v[u] = *reinterpret_cast<const u32x4_t*>(w + (long)n * ldw + k0 + 8 * kc);
    }
}
template <int N1>
__device__ __forceinline__ void fwd_panel_park(unsigned short* dst, int tid, const Panel<N1 * Kp<N1>::v / 8 / NT>& p) {
    constexpr int KP = Kp<N1>::v;
#pragma unroll
    for (int u = 0; u < N1 * KP / 8 / NT; ++u) {
        const int e = tid + NT * u, n = e / (KP / 8), kc = e % (KP / 8);
        *reinterpret_cast<u32x4_t*>(dst + n * (KP + PADE) + 8 * kc) = p.v[u];
    }
}
// backward panels: KP rows (k) x N1 columns of w (row stride ldw, row offset k0, column offset c0): image [k][N1 + 8]
template <int N1>
__device__ __forceinline__ void bwd_panel_load(const unsigned short* w, int ldw, int k0, int c0, int tid, Panel<N1 * Kp<N1>::v / 8 / NT>& p) {
    constexpr int KP = Kp<N1>::v;
#pragma unroll
    for (int u = 0; u < N1 * KP / 8 / NT; ++u) {
        const int e = tid + NT * u, k = e / (N1 / 8), nc = e % (N1 / 8);
        p.v[u] = *reinterpret_cast<const u32x4_t*>(w + (long)(k0 + k) * ldw + c0 + 8 * nc);
    }
}
template <int N1>
__device__ __forceinline__ void bwd_panel_park(unsigned short* dst, int tid, const Panel<N1 * Kp<N1>::v / 8 / NT>& p) {
    constexpr int KP = Kp<N1>::v;
#pragma unroll
    for (int u = 0; u < N1 * KP / 8 / NT; ++u) {
        const int e = tid + NT * u, k = e / (N1 / 8), nc = e % (N1 / 8);
        *reinterpret_cast<u32x4_t*>(dst + k * (N1 + PADE) + 8 * nc) = p.v[u];
    }
}

template <int N1, int CR>
__global__ __launch_bounds__(NT) void mlp_chain_fwd_kernel(ChainArgs g) {
    constexpr int KP = Kp<N1>::v, NC = N1 / 8, CT = NC / 16, RT = CR / 16, WP = KP + PADE, HP = N1 + PADE;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int XP = g.n0 + PADE;
    unsigned short* xs = smem;                                   // [CR][XP]   X slab; later [CR][HP] H
    unsigned short* wp = smem + CR * (XP > HP ? XP : HP);        // [2][N1][WP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int row0 = (int)blockIdx.x * CR;
    const int p1 = g.n0 / KP, np = p1 + N1 / KP;                 // panels of W1 (a multiple of NSETS), then NSETS of W2
    // (panel p: a select on the ADDRESS, never a branch around the loads: the loops below carry NO conditional load or store, so
    // that the compiler's counted s_waitcnt leaves the younger panels in flight - with a conditional park in the loop it waited for
    // everything: 20.8 us; panels beyond the last re-read panel 0 and are parked into a buffer nobody reads)
#define PIT_CHAIN_LOAD(p_, set_) do { const int pp_ = (p_) < np ? (p_) : 0; \
        fwd_panel_load<N1>(pp_ < p1 ? g.w1b : g.w2b, pp_ < p1 ? g.n0 : N1, (pp_ < p1 ? pp_ : pp_ - p1) * KP, tid, reg[set_]); } while (0)
#define PIT_CHAIN_STEP(p_, j_, A_PITCH_, A_K0_)                                                                       \
    do {                                                                                                              \
        __syncthreads();                                                                                              \
        fwd_panel_park<N1>(wp + (((p_) + 1) & 1) * N1 * WP, tid, reg[((j_) + 1) % NSETS]);                            \
        if (!(PIT_CHAIN_DBG & 8)) PIT_CHAIN_LOAD((p_) + 1 + NSETS, ((j_) + 1) % NSETS);                                \
        const unsigned short* wb_ = wp + ((p_) & 1) * N1 * WP;                                                        \
        if (!(PIT_CHAIN_DBG & 4))                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < KP / 32; ++ks) {                                                      \
            v8s_t a_[RT];                                                                                             \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) a_[rt] = frag_row(xs, A_PITCH_, 16 * rt + l15, (A_K0_) + 32 * ks + 8 * kq); \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                       \
                const v8s_t b_ = frag_row(wb_, WP, wave * NC + 16 * ct + l15, 32 * ks + 8 * kq);                      \
                _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mma(b_, a_[rt], acc[rt][ct]);         \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    Panel<N1 * KP / 8 / NT> reg[NSETS];
#pragma unroll
    for (int j = 0; j < NSETS; ++j) PIT_CHAIN_LOAD(j, j);
    // the X slab: fp32 rows -> bf16 image, all pieces of a thread requested at once
    {
        const int q4 = g.n0 / 4, total = CR * q4;
        for (int base = 0; base < total; base += 12 * NT) {
            float4 v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int e = base + tid + NT * u, r = e / q4, c = e - r * q4;
                v[u] = (PIT_CHAIN_DBG & 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : ldg4_if(g.x, (long)(row0 + r) * g.ldx + 4 * c, e < total && row0 + r < g.rows);
            }
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int e = base + tid + NT * u, r = e / q4, c = e - r * q4;
                if (e < total) *reinterpret_cast<uint2*>(xs + r * XP + 4 * c) = pack4(v[u]);
            }
        }
    }
    fwd_panel_park<N1>(wp, tid, reg[0]);
    PIT_CHAIN_LOAD(NSETS, 0);
    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // At the top of step p the sets hold panels p + 1 .. p + NSETS (set (p + j) % NSETS holds panel p + j): the oldest is parked into the
    // LDS buffer panel p - 1 just left and its set re-used for panel p + 1 + NSETS.  Rounds of NSETS steps: set indices are compile-time.
    for (int pb = 0; pb < p1; pb += NSETS) {
#pragma unroll
        for (int j = 0; j < NSETS; ++j) PIT_CHAIN_STEP(pb + j, j, XP, (pb + j) * KP);
    }
    // GEMM1 done: Z1 = acc + b1, H = gelu(Z1) -> memory (the backward reads them) and, as bf16, the A operand of GEMM2.
    // The products are formed TRANSPOSED (the weight fragment is the MFMA's A operand, the activation fragment its B: both have the
    // same register format): accumulator register i of lane (l15, kq) is out[row 16 rt + l15][column 16 ct + 4 kq + i] - four ADJACENT
    // columns of one row per lane, i.e. 16-byte stores (8-byte for bf16) instead of a 4-byte store per element.  (Diagnostic builds,
    // -DPIT_CHAIN_DBG, put 6.1 of the forward's 21.1 us on the stores, 5.5 on the MFMAs, 1.7 on the X slab and 0.7 on the weight
    // panels; the wider stores gave 21.1 -> 20.0 only: what costs is not the instruction count but that the Z1 / H stores sit in front
    // of GEMM2's counted waits - vmcnt counts stores too - and the Z2 / Y stores drain at the kernel's end, one workgroup per CU.
    // Keeping Z1 in registers and storing Z1 / H behind GEMM2 instead measured 20.0 -> 20.9 us: a CU's store rate is the limit, and in
    // front of GEMM2 the stores at least overlap its MFMAs.)
    __syncthreads();                                             // every wave is through with the X image
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col0 = wave * NC + 16 * ct + 4 * kq;
        const float4 bias = *reinterpret_cast<const float4*>(g.b1 + col0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = 16 * rt + l15;
            const float4 z = make_float4(acc[rt][ct][0] + bias.x, acc[rt][ct][1] + bias.y, acc[rt][ct][2] + bias.z, acc[rt][ct][3] + bias.w);
            const float4 hv = make_float4(gelu_f(z.x), gelu_f(z.y), gelu_f(z.z), gelu_f(z.w));
            *reinterpret_cast<uint2*>(xs + r * HP + col0) = pack4(hv);
            if (row0 + r < g.rows && !(PIT_CHAIN_DBG & 2)) {
                *reinterpret_cast<float4*>(g.z1 + (long)(row0 + r) * N1 + col0) = z;
                *reinterpret_cast<float4*>(g.h + (long)(row0 + r) * N1 + col0) = hv;
            }
            acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int j = 0; j < NSETS; ++j) PIT_CHAIN_STEP(p1 + j, j, HP, j * KP);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col0 = wave * NC + 16 * ct + 4 * kq;
        const float4 bias = *reinterpret_cast<const float4*>(g.b2 + col0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = 16 * rt + l15;
            if (row0 + r < g.rows && (!(PIT_CHAIN_DBG & 2) || acc[rt][ct][0] == 123.456f)) {
                const float4 z = make_float4(acc[rt][ct][0] + bias.x, acc[rt][ct][1] + bias.y, acc[rt][ct][2] + bias.z, acc[rt][ct][3] + bias.w);
                const float4 yv = make_float4(gelu_f(z.x), gelu_f(z.y), gelu_f(z.z), gelu_f(z.w));
                *reinterpret_cast<float4*>(g.z2 + (long)(row0 + r) * N1 + col0) = z;
                *reinterpret_cast<float4*>(g.y + (long)(row0 + r) * g.ldy + col0) = yv;
                if (g.y16) *reinterpret_cast<uint2*>(g.y16 + (long)(row0 + r) * N1 + col0) = pack4(yv);
            }
        }
    }
}
#undef PIT_CHAIN_STEP

#undef PIT_CHAIN_LOAD

template <int N1, int CR>
__global__ __launch_bounds__(NT) void mlp_chain_bwd_kernel(ChainArgs g) {
    constexpr int KP = Kp<N1>::v, NC = N1 / 8, CT = NC / 16, RT = CR / 16, AP = N1 + PADE, WP = N1 + PADE;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* a2s = smem;                       // [CR][AP]  dZ2
    unsigned short* a1s = a2s + CR * AP;              // [CR][AP]  dZ1
    unsigned short* wp = a1s + CR * AP;               // [2][KP][WP]
    float* ivs = reinterpret_cast<float*>(wp + 2 * KP * WP);     // [3][CR]  g16: 1 / rowsum of the slab's rows, per head (n0 <= 4 n1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int row0 = (int)blockIdx.x * CR;
    if (g.g16 && tid < (g.n0 / N1 - 1) * CR) {        // (read behind several barriers, in the dX epilogues)
        const int hd = tid / CR, r = min(row0 + tid % CR, g.rows - 1), bb = r / g.pts, n = r - bb * g.pts;
        ivs[tid] = g.rowstat[(((long)(g.mesh_batch == 1 ? 0 : bb) * (g.n0 / N1 - 1) + hd) * g.pts + n) * 4 + 2];
    }
    constexpr int PW = N1 / KP;                       // panels per [N1 x N1] product (= NSETS)
    static_assert(PW == NSETS, "a product is one round of the panel pipeline");
    const int chunks = g.n0 / N1, np = PW * (1 + chunks);
#define PIT_CHAIN_LOAD(p_, set_) do { const int pp_ = (p_) < np ? (p_) : 0; \
        bwd_panel_load<N1>(pp_ < PW ? g.w2b : g.w1b, pp_ < PW ? N1 : g.n0, (pp_ % PW) * KP, pp_ < PW ? 0 : ((pp_ - PW) / PW) * N1, tid, reg[set_]); } while (0)
#define PIT_CHAIN_STEP(p_, j_, A_IMG_)                                                                                \
    do {                                                                                                              \
        __syncthreads();                                                                                              \
        bwd_panel_park<N1>(wp + (((p_) + 1) & 1) * KP * WP, tid, reg[((j_) + 1) % NSETS]);                            \
        PIT_CHAIN_LOAD((p_) + 1 + NSETS, ((j_) + 1) % NSETS);                                                          \
        const unsigned short* wb_ = wp + ((p_) & 1) * KP * WP;                                                        \
        _Pragma("unroll") for (int ks = 0; ks < KP / 32; ++ks) {                                                      \
            v8s_t a_[RT];                                                                                             \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) a_[rt] = frag_row(A_IMG_, AP, 16 * rt + l15, (j_) * KP + 32 * ks + 8 * kq); \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                       \
                const v8s_t b_ = frag_tr(wb_, WP, 32 * ks, wave * NC + 16 * ct, l15, kq);                             \
                _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mma(b_, a_[rt], acc[rt][ct]);         \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    Panel<N1 * KP / 8 / NT> reg[NSETS];
#pragma unroll
    for (int j = 0; j < NSETS; ++j) PIT_CHAIN_LOAD(j, j);
    // dZ2 = dY gelu'(Z2): memory (the weight-gradient reductions read it) and the bf16 A image
    {
        constexpr int Q4 = N1 / 4, TOTAL = CR * Q4;
        float4 dy[TOTAL / NT], zz[TOTAL / NT];
#pragma unroll
        for (int u = 0; u < TOTAL / NT; ++u) {
            const int e = tid + NT * u, r = e / Q4, c = e % Q4;
            const bool ok = row0 + r < g.rows;
            dy[u] = ldg4_if(g.d_y, (long)(row0 + r) * g.ld_dy + 4 * c, ok);
            zz[u] = ldg4_if(g.z2r, (long)(row0 + r) * N1 + 4 * c, ok);
        }
#pragma unroll
        for (int u = 0; u < TOTAL / NT; ++u) {
            const int e = tid + NT * u, r = e / Q4, c = e % Q4;
            const float4 v = make_float4(dy[u].x * gelu_grad_f(zz[u].x), dy[u].y * gelu_grad_f(zz[u].y),
                                         dy[u].z * gelu_grad_f(zz[u].z), dy[u].w * gelu_grad_f(zz[u].w));
            *reinterpret_cast<uint2*>(a2s + r * AP + 4 * c) = pack4(v);
            if (row0 + r < g.rows) *reinterpret_cast<float4*>(g.dz2 + (long)(row0 + r) * N1 + 4 * c) = v;
        }
    }
    // Z1 at the accumulator positions of the first product's epilogue (transposed products, as the forward: row 16 rt + l15, four adjacent
    // columns from 16 ct + 4 kq), consumed after the first round
    float z1v[RT][CT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            buf_load4(wide_rsrc(g.z1r), row0 + 16 * rt + l15 < g.rows ? (unsigned)(((long)(row0 + 16 * rt + l15) * N1 + wave * NC + 16 * ct + 4 * kq) * 4) : OOB, z1v[rt][ct]);
    bwd_panel_park<N1>(wp, tid, reg[0]);
    PIT_CHAIN_LOAD(NSETS, 0);
    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // round 0: dZ1 = (dZ2 W2) gelu'(Z1) -> memory and the bf16 A image of the dX products
#pragma unroll
    for (int j = 0; j < NSETS; ++j) PIT_CHAIN_STEP(j, j, a2s);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col0 = wave * NC + 16 * ct + 4 * kq;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = 16 * rt + l15;
            const float4 v = make_float4(acc[rt][ct][0] * gelu_grad_f(z1v[rt][ct][0]), acc[rt][ct][1] * gelu_grad_f(z1v[rt][ct][1]),
                                         acc[rt][ct][2] * gelu_grad_f(z1v[rt][ct][2]), acc[rt][ct][3] * gelu_grad_f(z1v[rt][ct][3]));
            *reinterpret_cast<uint2*>(a1s + r * AP + col0) = pack4(v);          // (nobody reads a1s before the next step's barrier)
            if (row0 + r < g.rows) *reinterpret_cast<float4*>(g.dz1 + (long)(row0 + r) * N1 + col0) = v;
            acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // rounds 1 .. chunks: a column chunk of dX = dZ1 W1 each
    for (int c = 0; c < chunks; ++c) {
        const int pb = PW * (1 + c);
#pragma unroll
        for (int j = 0; j < NSETS; ++j) PIT_CHAIN_STEP(pb + j, j, a1s);
        if (g.d_x) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col0 = c * N1 + wave * NC + 16 * ct + 4 * kq;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r = 16 * rt + l15;
                    if (row0 + r < g.rows)
                        *reinterpret_cast<float4*>(g.d_x + (long)(row0 + r) * g.ld_dx + col0) = make_float4(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]);
                }
            }
            // the columns of head c - 1 of a self-attention layer's concat buffer: G_h = d_x / rowsum_h as bf16, (batch, H, pts, N1)
            if (g.g16 && c >= 1) {
                const int hd = c - 1, nh = chunks - 1;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r = row0 + 16 * rt + l15;
                    if (r >= g.rows) continue;
                    const int bb = r / g.pts, n = r - bb * g.pts;
                    const float iv = ivs[hd * CR + 16 * rt + l15];
                    unsigned short* dst = g.g16 + (((long)bb * nh + hd) * g.pts + n) * N1 + wave * NC + 4 * kq;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        *reinterpret_cast<uint2*>(dst + 16 * ct) = pack4(make_float4(acc[rt][ct][0] * iv, acc[rt][ct][1] * iv, acc[rt][ct][2] * iv, acc[rt][ct][3] * iv));
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
#undef PIT_CHAIN_STEP

#undef PIT_CHAIN_LOAD

// bf16 copies of up to 32 weight matrices in ONE launch (the chains read their weights as bf16; the copies are formed once per
// weight version - inside a captured step once per replay - by the host side, ops.prepare_chain_weights)
constexpr int CAST_MAX = 32;
struct CastArgs { const float* src[CAST_MAX]; unsigned short* dst[CAST_MAX]; long n4[CAST_MAX]; int n; };
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(CastArgs g) {
    const int t = blockIdx.y;
    const long n4 = g.n4[t];
    const float4* s = reinterpret_cast<const float4*>(g.src[t]);
    uint2* d = reinterpret_cast<uint2*>(g.dst[t]);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) d[i] = pack4(s[i]);
}

bool chain_shape_ok(int rows, int n0, int n1, int n2) {
    return rows > 0 && n1 == n2 && (n1 == 128 || n1 == 256) && n0 >= n1 && n0 % n1 == 0 && n0 <= 1024 &&
           (long)rows * n0 * 4 < (1L << 31) - 65536;
}
size_t chain_fwd_smem(int cr, int n0, int n1) { return (size_t)(cr * (std::max(n0, n1) + PADE) + 2 * n1 * (n1 / 4 + PADE)) * 2; }
size_t chain_bwd_smem(int cr, int n1) { return (size_t)(2 * cr * (n1 + PADE) + 2 * (n1 / 4) * (n1 + PADE)) * 2 + (size_t)3 * cr * 4; }
// Rows per workgroup.  A hid-256 workgroup fills a CU's LDS (one per CU) and takes ~19 us whatever its height (its 0.5 MB of weights
// at the rate one CU pulls from L2): Elasticity's 9 720 rows are 304 slabs of 32 = two rounds on 256 CUs (38 / 35 us measured), 203
// slabs of 48 = one.  48 when that saves a round and the images still fit; hid 128 runs several workgroups per CU: 32.
int chain_slab_rows(int rows, int n0, int n1) {
    if (n1 != 256 || chain_fwd_smem(48, n0, n1) > 160 * 1024) return 32;
    const int cus = 256, r32 = ((rows + 31) / 32 + cus - 1) / cus, r48 = ((rows + 47) / 48 + cus - 1) / cus;
    return r48 < r32 ? 48 : 32;
}
template <int N1, int CR>
void chain_launch_fwd(const ChainArgs& g, hipStream_t s) {
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_fwd_kernel<N1, CR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL((mlp_chain_fwd_kernel<N1, CR>), dim3((unsigned)((g.rows + CR - 1) / CR)), dim3(NT), chain_fwd_smem(CR, g.n0, N1), s, g);
}
template <int N1, int CR>
void chain_launch_bwd(const ChainArgs& g, hipStream_t s) {
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_bwd_kernel<N1, CR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL((mlp_chain_bwd_kernel<N1, CR>), dim3((unsigned)((g.rows + CR - 1) / CR)), dim3(NT), chain_bwd_smem(CR, N1), s, g);
}
bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// 1 when pit_mlp_chain_fwd / _bwd cover kaiming_mlp (n0 -> n1 -> n2) followed by gelu on `rows` rows: hid 128 / 256, n0 a multiple
// of the hidden width (the processor's (1 + H) hid), a few thousand rows (below: the fused small-regime kernels; far above: the
// tiled GEMMs, whose workgroups then amortise their weights)
extern "C" int pit_mlp_chain_supported(int rows, int n0, int n1, int n2) {
    return chain_shape_ok(rows, n0, n1, n2) && rows >= 1024 && rows <= 65536;
}

extern "C" int pit_mlp_chain_fwd(const float* x, long ldx, int rows, int n0, int n1, const unsigned short* w1_bf16, const float* b1,
                                 const unsigned short* w2_bf16, const float* b2, float* z1, float* h, float* z2, float* y, long ldy,
                                 unsigned short* y16, void* stream) {
    if (!x || !w1_bf16 || !b1 || !w2_bf16 || !b2 || !z1 || !h || !z2 || !y) return PIT_ERR_NULL;
    if (!chain_shape_ok(rows, n0, n1, n1) || ldx < n0 || ldy < n1) return PIT_ERR_UNSUPPORTED;
    if (ldx % 4 || ldy % 4 || !a16(x) || !a16(w1_bf16) || !a16(w2_bf16) || !a16(b1) || !a16(b2) || !a16(z1) || !a16(h) || !a16(z2) || !a16(y) ||
        (y16 && (reinterpret_cast<uintptr_t>(y16) & 7))) return PIT_ERR_SIZE;
    ChainArgs g = ChainArgs();
    g.rows = rows; g.n0 = n0; g.n1 = n1; g.x = x; g.ldx = ldx; g.w1b = w1_bf16; g.w2b = w2_bf16; g.b1 = b1; g.b2 = b2;
    g.z1 = z1; g.h = h; g.z2 = z2; g.y = y; g.ldy = ldy; g.y16 = y16;
    hipStream_t s = (hipStream_t)stream;
    if (n1 == 128) chain_launch_fwd<128, 32>(g, s);
    else if (chain_slab_rows(rows, n0, n1) == 48) chain_launch_fwd<256, 48>(g, s);
    else chain_launch_fwd<256, 32>(g, s);
    PIT_CHECK_LAUNCH();
    return 0;
}

// The data path of the backward: scratch = dZ1 (rows*n1) | dZ2 (rows*n1) fp32 - the layout pit_mlp_bwd_params reads - and d_x (NULL:
// not needed).  d_y rows ld_dy apart.
extern "C" int pit_mlp_chain_bwd(int rows, int n0, int n1, const unsigned short* w1_bf16, const unsigned short* w2_bf16,
                                 const float* z1, const float* z2, const float* d_y, long ld_dy, float* d_x, long ld_dx,
                                 float* scratch, unsigned short* g16, const float* rowstat, int pts, int mesh_batch, void* stream) {
    if (!w1_bf16 || !w2_bf16 || !z1 || !z2 || !d_y || !scratch) return PIT_ERR_NULL;
    if (!chain_shape_ok(rows, n0, n1, n1) || ld_dy < n1 || (d_x && ld_dx < n0)) return PIT_ERR_UNSUPPORTED;
    if (ld_dy % 4 || ld_dx % 4 || !a16(d_y) || !a16(z1) || !a16(z2) || !a16(w1_bf16) || !a16(w2_bf16) || !a16(scratch) || (d_x && !a16(d_x)) ||
        (g16 && (reinterpret_cast<uintptr_t>(g16) & 7))) return PIT_ERR_SIZE;
    ChainArgs g = ChainArgs();
    g.rows = rows; g.n0 = n0; g.n1 = n1; g.w1b = w1_bf16; g.w2b = w2_bf16; g.z1r = z1; g.z2r = z2; g.d_y = d_y; g.ld_dy = ld_dy;
    g.d_x = d_x; g.ld_dx = ld_dx; g.dz1 = scratch; g.dz2 = scratch + (long)rows * n1;
    if (g16) {
        if (!rowstat || !d_x) return PIT_ERR_NULL;
        if (pts <= 0 || rows % pts || n0 < 2 * n1 || (mesh_batch != 1 && mesh_batch != rows / pts)) return PIT_ERR_SIZE;
        g.g16 = g16; g.rowstat = rowstat; g.pts = pts; g.mesh_batch = mesh_batch;
    }
    hipStream_t s = (hipStream_t)stream;
    if (n1 == 128) chain_launch_bwd<128, 32>(g, s);
    else if (chain_slab_rows(rows, n0, n1) == 48) chain_launch_bwd<256, 48>(g, s);
    else chain_launch_bwd<256, 32>(g, s);
    PIT_CHECK_LAUNCH();
    return 0;
}

// dst[i] = bf16(src[i]) (round to nearest even) for n tensors of count[i] floats each (multiples of 4, 16-byte aligned): one launch
extern "C" int pit_cast_bf16_multi(int n, const float* const* src, unsigned short* const* dst, const long* count, void* stream) {
    if (n <= 0 || n > CAST_MAX || !src || !dst || !count) return PIT_ERR_SIZE;
    CastArgs g = CastArgs();
    long most = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i]) return PIT_ERR_NULL;
        if (count[i] <= 0 || count[i] % 4 || !a16(src[i]) || (reinterpret_cast<uintptr_t>(dst[i]) & 7)) return PIT_ERR_SIZE;
        g.src[i] = src[i]; g.dst[i] = dst[i]; g.n4[i] = count[i] / 4;
        most = std::max(most, g.n4[i]);
    }
    g.n = n;
    const unsigned gx = (unsigned)std::max<long>(1, std::min<long>((most + 255) / 256, 64));
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, g);
    PIT_CHECK_LAUNCH();
    return 0;
}
