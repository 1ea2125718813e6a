// The decoder side with the decoder MLP's first layer FOLDED INTO THE VALUES (round 6).
//
// pit.decoder (pit.py:124-127) is  de(up(values)):  X = concat_h(P_h V)  (batch, n_out, H*hid),  Z1 = X W1^T + b1,
// y = gelu(Z1) W2^T + b2.  Nothing non-linear sits between the up-projection and the first Linear, so
//
//     Z1 = sum_h P_h (V W1_h^T) + b1,        W1_h = W1[:, h*hid : (h+1)*hid]
//
// with the small product  VW[:, n H + h] = (V W1_h^T)[:, n]  formed on the LATENT points (n_in rows per sample) instead of the first
// Linear on the OUTPUT points (n_out rows: 16x as many at Vorticity, 15x at NACA, 5x at Cylinder).  The (batch, n_out, H*hid)
// tensor - 168 MB of fp32 at Vorticity b=20 - and the three GEMMs on its 81 920 rows (forward, dX, dW1: 64 GF of the step's
// 183) do not exist any more; the backward is the same identity transposed:
//
//     d(VW_h) = P_h^T dZ1,      d(scale_h) = -(Q_h VW_h) . dZ1,      dV = sum_h d(VW_h) W1_h,    dW1_h = sum_b d(VW_h)^T V
//
// (the last two are small GEMMs on n_in rows again: pit_linear_bwd).  Same function, re-associated products: within 1e-6 of the
// reference's order in fp32 (tests/test_gpu_round6.py compares every launch with the oracle).
//
// Kernels, for batch-free mesh pairs with a slab plan whose slabs are 64 / 128 / 256 rows tall (pit_slab_plan.rows):
//   fold_fwd_kernel   a workgroup owns (sample, slab, 64 output columns): the slab's union value rows of BOTH heads' VW go to
//                     LDS once, then the slab is walked in passes of 32 rows: Z[32 x 64] = sum_h P_h[32 x U] VW_h[U x 64].
//   fold_bwd_kernel   the same walk with d(VW_h)[U x 64] = sum_rows P_h^T dZ kept in MFMA accumulators over the WHOLE slab and
//                     added to memory once per slab (fp32 atomics) - the tall slabs are what makes that cheap: at Vorticity a
//                     256-row slab (four grid lines) has a union of 64 latent keys where its sixteen 16-row slabs have 324
//                     between them, i.e. 5x fewer atomic adds than union_att_bwd_kernel issues (which they bound: 82 of 120 us).
//   fp32 math mode: v_mfma_f32_16x16x4_f32 on fp32 tiles.  bf16 math mode: tiles rounded to bf16 on their way into LDS,
//   v_mfma_f32_16x16x32_bf16 with the [k][n] images read through ds_read_b64_tr_b16 (cdna guide T10), fp32 accumulation.
//   thin_tail_fwd / _bwd_kernel   the rest of `de` for out_dim <= 4: y = gelu(Z + b1) W2^T + b2 as row dots; backward
//                     dZ = (d_y W2) * gelu'(Z + b1) with db1, dW2, db2 reduced in the same pass (no saved activations at all:
//                     gelu is recomputed from Z).
#include "pit_common.h"
#include "pit_block_dev.h"

namespace {

constexpr int EU = PIT_SLAB_UNION_MAX;     // union keys a slab tile holds (64)
constexpr int SR = 32;                     // rows per pass
constexpr int CW = 64;                     // output columns per workgroup

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void* p) { return make_rsrc(p, 0x7ffffff0u); }
constexpr unsigned OOB = 0x7ffffff8u;
__device__ __forceinline__ float4 ldg4_if(const float* p, long i, bool ok) {
    float v[4];
    buf_load4(wide_rsrc(p), ok ? (unsigned)(i * 4) : OOB, v);
    return make_float4(v[0], v[1], v[2], v[3]);
}
// four consecutive bf16 (8 bytes) widened to fp32
__device__ __forceinline__ float4 ldh4_if(const void* p, long i, bool ok) {
    const __amdgpu_buffer_rsrc_t r = wide_rsrc(p);
    const unsigned off = ok ? (unsigned)(i * 2) : OOB;
    const unsigned lo = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
    const unsigned hi = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)(ok ? off + 4u : OOB), 0, 0);
    return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                       __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ uint2 pack4_bf16(float4 v) {
    uint2 pk;
    pk.x = (unsigned)f_to_bf16(v.x) | ((unsigned)f_to_bf16(v.y) << 16);
    pk.y = (unsigned)f_to_bf16(v.z) | ((unsigned)f_to_bf16(v.w) << 16);
    return pk;
}

// ---- the two MFMA flavours.  A "k group" is what one mma() contracts: 16 k's as four v_mfma_f32_16x16x4_f32 (lane quarter kq
// holds k = 16 kg + 4 kq + j for instruction j: operands read from a [row][k] image are then ONE 16-byte LDS read) or 32 k's as
// one v_mfma_f32_16x16x32_bf16 (lane quarter kq holds k = 32 kg + 8 kq + j, j = 0..7).  Accumulator register i of lane l is
// D[row 4 (l >> 4) + i][column l & 15] in both.
template <bool BF> struct Fl;
template <> struct Fl<false> {
    typedef float T;
    typedef f32x4_t Frag;
    static constexpr int KG = 16, PADE = 4;
    // from a [i][k] image: row i0 + l15, the lane's four k's
    static __device__ __forceinline__ Frag rowfrag(const T* t, int pitch, int i0, int kg, int l15, int kq) {
        return *reinterpret_cast<const f32x4_t*>(t + (i0 + l15) * pitch + 16 * kg + 4 * kq);
    }
    // from a [k][n] image: column n0 + l15, the lane's four k's (rows 4 kq .. 4 kq + 3 of the group: 4 * pitch = 16 mod 32 banks
    // apart between lane quarters - conflict-free with pitch = width + 4)
    static __device__ __forceinline__ Frag kfrag(const T* t, int pitch, int kg, int n0, int l15, int kq) {
        const T* p = t + (16 * kg + 4 * kq) * pitch + n0 + l15;
        Frag f;
        f[0] = p[0]; f[1] = p[pitch]; f[2] = p[2 * pitch]; f[3] = p[3 * pitch];
        return f;
    }
    static __device__ __forceinline__ f32x4_t mma(Frag a, Frag b, f32x4_t c) {
        c = mfma_16x16x4(a[0], b[0], c);
        c = mfma_16x16x4(a[1], b[1], c);
        c = mfma_16x16x4(a[2], b[2], c);
        c = mfma_16x16x4(a[3], b[3], c);
        return c;
    }
    static __device__ __forceinline__ void put4(T* dst, float4 v) { *reinterpret_cast<float4*>(dst) = v; }
    static __device__ __forceinline__ float get(const T* p) { return *p; }
};
template <> struct Fl<true> {
    typedef unsigned short T;
    typedef v8s_t Frag;
    static constexpr int KG = 32, PADE = 8;
    static __device__ __forceinline__ Frag rowfrag(const T* t, int pitch, int i0, int kg, int l15, int kq) {
        return *reinterpret_cast<const v8s_t*>(t + (i0 + l15) * pitch + 32 * kg + 8 * kq);
    }
    // ds_read_b64_tr_b16 (cdna guide T10): within the 16 lanes of quarter kq, lane 4 q + p supplies the address of row q, columns
    // 4 p .. 4 p + 3 of a 4 x 16 block; lane i receives column i of the four rows.  Two blocks (rows +0..3, +4..7) give the lane
    // its eight k's.  EXEC must be all ones: called from wave-uniform control flow only.
    static __device__ __forceinline__ Frag kfrag(const T* t, int pitch, int kg, int n0, int l15, int kq) {
        const T* a0 = t + (32 * kg + 8 * kq + (l15 >> 2)) * pitch + n0 + 4 * (l15 & 3);
        typedef v4s_t __attribute__((address_space(3))) * lds_v4;
        const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0));
        const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0 + 4 * pitch));
        Frag f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    }
    static __device__ __forceinline__ f32x4_t mma(Frag a, Frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, a), __builtin_bit_cast(v8bf_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void put4(T* dst, float4 v) { *reinterpret_cast<uint2*>(dst) = pack4_bf16(v); }
    static __device__ __forceinline__ float get(const T* p) { return bf16_to_f(*p); }
};

#ifndef PIT_FOLD_DBG
#define PIT_FOLD_DBG 0       // diagnostic builds (tools/variant_build.sh) of fold_bwd_kernel: 1 no d(scale) contraction, 2 no d(VW) contraction,
                             // 4 no epilogue, 8 tiles not refreshed in the loop, 16 no union staging (results are void, times are not)
#endif
struct FoldArgs {
    pit_slab_plan p; int um, batch, dim, chunks;
    const float* vw; long ld_vw, vw_bstride;             // (batch, n_in, dim * H), head-interleaved: column n * H + h = head h's value for output column n
    const float *pw, *qw;                                // (n_slabs * H, rows, um): the step's weights (pit_fold_weights)
    const unsigned short *pw16, *qw16;                   // the same as bf16 (what the bf16 flavour reads: half the bytes per pass, nothing to round)
    void* z; long ld_z, z_bstride; int z16;              // forward: out (batch, n_out, dim)
    const void* dz; long ld_dz, dz_bstride; int dz16;    // backward: its gradient
    float* d_vw; long ld_dvw, dvw_bstride;               // backward: ADDED to (fp32 atomics), layout of vw - or, with `tiles`, written
    double* dscale;
    float* tiles;                                        // backward, optional: (batch, n_slabs, 64 slots, dim * H) per-slab sums (fold_reduce_kernel)
    const int *rev_ptr, *rev_ent;                        // key -> the (slab * 64 + slot) entries that hold it (CSR over n_in keys)
};

// workgroup id -> (sample, slab, chunk): all workgroups of a slab on ONE XCD (ids are dealt to the XCDs round-robin) - what they
// share, the slab's weight tiles, is the larger stream (the weights are the same for every sample); a sample's VW rows are small.
// (Darcy's last slab is short - 1849 rows = 7 x 256 + 57 - and its XCD gets a quarter of the others' work; dealing sample b slab
// (x + b) mod 8 instead measured 245.3 against 245.4 us: the dispatcher already fills idle CUs with the later ids.)
__device__ __forceinline__ bool fold_ids(int id, const FoldArgs& g, int& b, int& slab, int& chunk) {
    const int x = id & 7, k = id >> 3, per = g.batch * g.chunks;
    slab = x + 8 * (k / per);
    const int rem = k % per;
    b = rem / g.chunks;
    chunk = rem - b * g.chunks;
    return slab < g.p.n_slabs;
}

// The slab's union value rows of every head.  VW is HEAD-INTERLEAVED: column n * H + h holds head h's value for output column n
// (that order makes the fold a plain Linear with W1's own memory read as an (H dim, dim) matrix - pit_linear_fwd - and its weight
// gradient lands in W1.grad's layout).  A workgroup's 64 output columns are therefore ONE run of 64 H floats per key row: thread
// t owns 16-byte piece e = t + 256 u of the (um rows x 16 H pieces) block and de-interleaves it on its way into the per-head
// tiles.  The plan pads its key lists with key 0 (a valid row); slots beyond the union load out of range (zeros) -
// unconditional loads, nothing consumed before all of them are requested.
template <bool BF, int H, int NT = 256>
__device__ __forceinline__ void stage_union(const FoldArgs& g, int b, int slab, int chunk, int nk, int umk, typename Fl<BF>::T* vt,
                                            int tid) {
    typedef typename Fl<BF>::T T;
    constexpr int VP = CW + Fl<BF>::PADE, PPR = 16 * H, NU = EU * PPR / NT;      // pieces per row, pieces per thread
    const int* kp = g.p.keys + (long)slab * g.p.umax;
    int key[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) key[u] = kp[(tid + NT * u) / PPR];
    const float* vb = g.vw + (long)b * g.vw_bstride + (long)chunk * CW * H;
    float4 v[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int e = tid + NT * u, row = e / PPR, pq = e % PPR;
        v[u] = ldg4_if(vb, (long)key[u] * g.ld_vw + 4 * pq, row < nk);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int e = tid + NT * u, row = e / PPR, pq = e % PPR;
        if (row >= umk) continue;
        if (H == 1) {
            Fl<BF>::put4(vt + row * VP + 4 * pq, v[u]);
        } else {                                  // (n, h) pairs: x = (2 pq, 0), y = (2 pq, 1), z = (2 pq + 1, 0), w = (2 pq + 1, 1)
            T* d0 = vt + row * VP + 2 * pq;
            T* d1 = vt + (umk + row) * VP + 2 * pq;
            if (BF) {
                *reinterpret_cast<unsigned*>(d0) = (unsigned)f_to_bf16(v[u].x) | ((unsigned)f_to_bf16(v[u].z) << 16);
                *reinterpret_cast<unsigned*>(d1) = (unsigned)f_to_bf16(v[u].y) | ((unsigned)f_to_bf16(v[u].w) << 16);
            } else {
                *reinterpret_cast<float2*>(d0) = make_float2(v[u].x, v[u].z);
                *reinterpret_cast<float2*>(d1) = make_float2(v[u].y, v[u].w);
            }
        }
    }
}

// A 32-row pass of a weight tensor: (H, 32, um) floats, each head's 32 x um block contiguous in memory, as 16-byte pieces
template <int H, int NT = 256> struct WReq { float4 v[H * SR * EU / 4 / NT]; };
template <int H, int NT = 256>
__device__ __forceinline__ void wreq(const float* w, const FoldArgs& g, int slab, int sub, int tid, WReq<H, NT>& r) {
    const int perh = SR * g.um / 4, npc = H * perh;
#pragma unroll
    for (int u = 0; u < H * SR * EU / 4 / NT; ++u) {
        const int e = tid + NT * u, h = e / perh, rem = e - h * perh;
        r.v[u] = ldg4_if(w, ((long)(slab * H + h) * g.p.rows + sub * SR) * g.um + 4L * rem, e < npc);
    }
}
template <bool BF, int H, int NT = 256>
__device__ __forceinline__ void wpark(typename Fl<BF>::T* tile, int pitch, const FoldArgs& g, int tid, const WReq<H, NT>& r) {
    const int um4 = g.um / 4, perh = SR * um4, npc = H * perh;
#pragma unroll
    for (int u = 0; u < H * SR * EU / 4 / NT; ++u) {
        const int e = tid + NT * u, h = e / perh, rem = e - h * perh, row = rem / um4, c4 = rem - row * um4;
        if (e < npc) Fl<BF>::put4(tile + (h * SR + row) * pitch + 4 * c4, r.v[u]);
    }
}
// the bf16 flavour's weight pass from the bf16 tiles of pit_fold_weights: 16-byte pieces of 8 weights, parked as they are
typedef unsigned u32x4w_t __attribute__((ext_vector_type(4)));
template <int H, int NT> struct WReq16 { u32x4w_t v[(H * SR * EU / 8 + NT - 1) / NT]; };
template <int H, int NT>
__device__ __forceinline__ void wreq16(const unsigned short* w, const FoldArgs& g, int slab, int sub, int tid, WReq16<H, NT>& r) {
    const int perh = SR * g.um / 8, npc = H * perh;
    const __amdgpu_buffer_rsrc_t rw = wide_rsrc(w);
#pragma unroll
    for (int u = 0; u < (H * SR * EU / 8 + NT - 1) / NT; ++u) {
        const int e = tid + NT * u, h = e / perh, rem = e - h * perh;
        const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(e < npc ? (unsigned)((((long)(slab * H + h) * g.p.rows + sub * SR) * g.um + 8L * rem) * 2) : OOB), 0, 0);
        r.v[u] = u32x4w_t{(unsigned)q.x, (unsigned)q.y, (unsigned)q.z, (unsigned)q.w};
    }
}
template <int H, int NT>
__device__ __forceinline__ void wpark16(unsigned short* tile, int pitch, const FoldArgs& g, int tid, const WReq16<H, NT>& r) {
    const int um8 = g.um / 8, perh = SR * um8, npc = H * perh;
#pragma unroll
    for (int u = 0; u < (H * SR * EU / 8 + NT - 1) / NT; ++u) {
        const int e = tid + NT * u, h = e / perh, rem = e - h * perh, row = rem / um8, c8 = rem - row * um8;
        if (e < npc) *reinterpret_cast<u32x4w_t*>(tile + (h * SR + row) * pitch + 8 * c8) = r.v[u];
    }
}
// one pass's weight request / park of either flavour (fp32 tiles -> fp32 LDS image; bf16 tiles -> bf16 LDS image)
template <bool BF, int H, int NT> struct WPass { typedef WReq<H, NT> T; };
template <int H, int NT> struct WPass<true, H, NT> { typedef WReq16<H, NT> T; };
template <bool BF, int H, int NT>
__device__ __forceinline__ void wrequest(const FoldArgs& g, bool q, int slab, int sub, int tid, typename WPass<BF, H, NT>::T& r) {
    if constexpr (BF) wreq16<H, NT>(q ? g.qw16 : g.pw16, g, slab, sub, tid, r);
    else wreq<H, NT>(q ? g.qw : g.pw, g, slab, sub, tid, r);
}
template <bool BF, int H, int NT>
__device__ __forceinline__ void wstore(typename Fl<BF>::T* tile, int pitch, const FoldArgs& g, int tid, const typename WPass<BF, H, NT>::T& r) {
    if constexpr (BF) wpark16<H, NT>(tile, pitch, g, tid, r);
    else wpark<BF, H, NT>(tile, pitch, g, tid, r);
}
// zero the columns [um, umk) of a weight tile (bf16 flavour, um = 48: the k groups are 32 wide)
template <bool BF, int H, int NT = 256>
__device__ __forceinline__ void wpad(typename Fl<BF>::T* tile, int pitch, int um, int umk, int tid) {
    const int padc = umk - um;
    for (int e = tid; e < H * SR * padc; e += NT) tile[(e / padc) * pitch + um + e % padc] = 0;
}

// 32 rows x 64 columns of the activation gradient (fp32 or bf16 in memory): 4-column pieces, two (one) per thread
template <int NT = 256> struct DReq { float4 v[SR * CW / 4 / NT]; };
template <int NT = 256>
__device__ __forceinline__ void dreq(const FoldArgs& g, int b, int slab, int chunk, int sub, int tid, DReq<NT>& r) {
#pragma unroll
    for (int u = 0; u < SR * CW / 4 / NT; ++u) {
        const int e = tid + NT * u, row = e >> 4, q = e & 15;
        const int n = slab * g.p.rows + sub * SR + row;
        const long o = (long)b * g.dz_bstride + (long)n * g.ld_dz + chunk * CW + 4 * q;
        r.v[u] = g.dz16 ? ldh4_if(g.dz, o, n < g.p.n_out) : ldg4_if(reinterpret_cast<const float*>(g.dz), o, n < g.p.n_out);
    }
}
template <bool BF, int NT = 256>
__device__ __forceinline__ void dpark(typename Fl<BF>::T* tile, int tid, const DReq<NT>& r) {
    constexpr int VP = CW + Fl<BF>::PADE;
#pragma unroll
    for (int u = 0; u < SR * CW / 4 / NT; ++u) {
        const int e = tid + NT * u;
        Fl<BF>::put4(tile + (e >> 4) * VP + 4 * (e & 15), r.v[u]);
    }
}

constexpr int NTB = 512;     // threads of the fold kernels: eight wavefronts
// wave = 4 gsel + w4 owns output columns [16 w4, 16 w4 + 16) of a pass, both heads; NT = 512: rows [16 gsel, 16 gsel + 16) (fp32 flavour:
// Darcy b=256 134 -> 130 us, Vorticity 67.6 -> 63.0), NT = 256: all 32 rows (bf16 flavour: eight waves measured 35.3 us against 32.5)
template <int H, bool BF, int NT>
__global__ __launch_bounds__(NT, 4) void fold_fwd_kernel(FoldArgs g) {
    typedef Fl<BF> F;
    typedef typename F::T T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int um = g.um, umk = BF ? ((um + 31) & ~31) : um;
    const int PP = umk + F::PADE;
    constexpr int VP = CW + F::PADE, ZP = CW + 4;
    T* vt = reinterpret_cast<T*>(smem_raw);            // [H][umk][VP]   union value rows
    T* pt = vt + H * umk * VP;                          // [H][SR][PP]    this pass's weights
    float* zs = reinterpret_cast<float*>(pt + H * SR * PP);      // [SR][ZP]  the pass's output tile on its way to memory
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int gsel = wave >> 2, w4 = wave & 3;
    int b, slab, chunk;
    if (!fold_ids((int)blockIdx.x, g, b, slab, chunk)) return;
    const pit_slab_plan& p = g.p;
    const int nsub = p.rows / SR;
    const int nk = min(p.nkeys[slab], EU);
    typename WPass<BF, H, NT>::T wr;
    wrequest<BF, H, NT>(g, false, slab, 0, tid, wr);
    stage_union<BF, H, NT>(g, b, slab, chunk, nk, umk, vt, tid);
    if (BF && umk != um) wpad<BF, H, NT>(pt, PP, um, umk, tid);
    wstore<BF, H, NT>(pt, PP, g, tid, wr);
    __syncthreads();
    for (int sub = 0; sub < nsub; ++sub) {
        const int row0 = slab * p.rows + sub * SR;
        if (row0 >= p.n_out) break;                                  // (workgroup-uniform: the mesh ends inside this slab)
        const bool more = sub + 1 < nsub && row0 + SR < p.n_out;
        if (more) wrequest<BF, H, NT>(g, false, slab, sub + 1, tid, wr);     // next pass's weights fly under this pass's contraction
        constexpr int NRT = SR / 16 / (NT / 256);
        f32x4_t acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < H; ++h) {
            for (int kg = 0; kg < umk / F::KG; ++kg) {
                const typename F::Frag bf = F::kfrag(vt + h * umk * VP, VP, kg, 16 * w4, l15, kq);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
                    acc[rt] = F::mma(F::rowfrag(pt + h * SR * PP, PP, 16 * (NRT * gsel + rt), kg, l15, kq), bf, acc[rt]);
            }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) zs[(16 * (NRT * gsel + rt) + 4 * kq + i) * ZP + 16 * w4 + l15] = acc[rt][i];
        __syncthreads();
        // the tile to memory as 4-column pieces (16 B of fp32, 8 B of bf16)
#pragma unroll
        for (int u = 0; u < SR * CW / 4 / NT; ++u) {
            const int e = tid + NT * u, row = e >> 4, q = e & 15, n = row0 + row;
            if (n < p.n_out) {
                const float4 v = *reinterpret_cast<const float4*>(zs + row * ZP + 4 * q);
                const long o = (long)b * g.z_bstride + (long)n * g.ld_z + chunk * CW + 4 * q;
                if (g.z16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.z) + o) = pack4_bf16(v);
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.z) + o) = v;
            }
        }
        if (more) wstore<BF, H, NT>(pt, PP, g, tid, wr);
        __syncthreads();
    }
}

// Eight wavefronts: wave = 4 gsel + w4 owns output columns [16 w4, 16 w4 + 16) and, of the two contractions of a pass,
//   two heads: head gsel of both (its own d(VW_h) accumulators, its own d(scale_h) sum),
//   one head:  slots [32 gsel, 32 gsel + 32) of d(VW) and rows [16 gsel, 16 gsel + 16) of the d(scale) product.
// (First version: four wavefronts, each with both heads - 156 registers, two workgroups = two wavefronts per SIMD, and each of a
// pass's two dependency chains - operands through LDS into 128 MFMAs, and load -> barrier -> park -> barrier - took ~3.3 us where
// the MFMAs are 1.9: 280 us at Darcy b=256, 176 of them with the contractions compiled out.  Twice the wavefronts per SIMD with
// half the accumulators each; a thread requests half the bytes.)
template <int H, bool BF>
__global__ __launch_bounds__(NTB, 4) void fold_bwd_kernel(FoldArgs g) {
    typedef Fl<BF> F;
    typedef typename F::T T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int um = g.um, umk = BF ? ((um + 31) & ~31) : um;
    const int PP = umk + F::PADE;
    constexpr int VP = CW + F::PADE;
    constexpr int NMT = (H == 2) ? EU / 16 : EU / 32, NRT = (H == 2) ? SR / 16 : SR / 32;
    T* vt = reinterpret_cast<T*>(smem_raw);            // [H][umk][VP]   union value rows (for d(scale))
    T* pt = vt + H * umk * VP;                          // [H][SR][PP]    P of this pass, read as the [k = row][i = slot] image
    T* qt = pt + H * SR * PP;                           // [H][SR][PP]    Q = P (m - mbar)
    T* dt = qt + H * SR * PP;                           // [SR][VP]       this pass's rows of dZ
    double* wred = reinterpret_cast<double*>(dt + SR * VP);      // [8]      (dt's end is 16-byte aligned: SR * VP * sizeof(T) is)
    int* keys_s = reinterpret_cast<int*>(wred + 8);              // [EU]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int gsel = wave >> 2, w4 = wave & 3;
    const int hh = (H == 2) ? gsel : 0, mt0 = (H == 2) ? 0 : NMT * gsel, rt0 = (H == 2) ? 0 : NRT * gsel;
    int b, slab, chunk;
    if (!fold_ids((int)blockIdx.x, g, b, slab, chunk)) return;
    const pit_slab_plan& p = g.p;
    const int nsub = p.rows / SR;
    const int nk = min(p.nkeys[slab], EU);
    const bool want_scale = g.dscale != nullptr;
    const int akey = p.keys[(long)slab * p.umax + (tid & (EU - 1))];
    typename WPass<BF, H, NTB>::T pr, qr;
    DReq<NTB> dr;
    wrequest<BF, H, NTB>(g, false, slab, 0, tid, pr);
    wrequest<BF, H, NTB>(g, true, slab, 0, tid, qr);
    dreq<NTB>(g, b, slab, chunk, 0, tid, dr);
    if (!(PIT_FOLD_DBG & 16)) stage_union<BF, H, NTB>(g, b, slab, chunk, nk, umk, vt, tid);
    if (tid < EU) keys_s[tid] = akey;
    if (BF && umk != um) { wpad<BF, H, NTB>(pt, PP, um, umk, tid); wpad<BF, H, NTB>(qt, PP, um, umk, tid); }
    wstore<BF, H, NTB>(pt, PP, g, tid, pr);
    wstore<BF, H, NTB>(qt, PP, g, tid, qr);
    dpark<BF, NTB>(dt, tid, dr);
    __syncthreads();
    f32x4_t accT[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) accT[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    double part = 0.0;
    const T* pth = pt + hh * SR * PP;
    const T* qth = qt + hh * SR * PP;
    const T* vth = vt + hh * umk * VP;
    for (int sub = 0; sub < nsub; ++sub) {
        const int row0 = slab * p.rows + sub * SR;
        if (row0 >= p.n_out) break;
        const bool more = sub + 1 < nsub && row0 + SR < p.n_out;
        if (more && !(PIT_FOLD_DBG & 8)) {
            wrequest<BF, H, NTB>(g, false, slab, sub + 1, tid, pr);
            wrequest<BF, H, NTB>(g, true, slab, sub + 1, tid, qr);
            dreq<NTB>(g, b, slab, chunk, sub + 1, tid, dr);
        }
        // d(VW_h)[slot][col] += sum_rows P_h[row][slot] dZ[row][col]
        if (!(PIT_FOLD_DBG & 2))
#pragma unroll
        for (int kg = 0; kg < SR / F::KG; ++kg) {
            const typename F::Frag bf = F::kfrag(dt, VP, kg, 16 * w4, l15, kq);
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
                if (16 * (mt0 + mt) < umk) accT[mt] = F::mma(F::kfrag(pth, PP, kg, 16 * (mt0 + mt), l15, kq), bf, accT[mt]);
        }
        // d(scale_h) -= sum (Q_h VW_h) . dZ
        if (want_scale && !(PIT_FOLD_DBG & 1)) {
            f32x4_t acc[NRT];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            for (int kg = 0; kg < umk / F::KG; ++kg) {
                const typename F::Frag bf = F::kfrag(vth, VP, kg, 16 * w4, l15, kq);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
                    acc[rt] = F::mma(F::rowfrag(qth, PP, 16 * (rt0 + rt), kg, l15, kq), bf, acc[rt]);
            }
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    part += (double)acc[rt][i] * (double)F::get(dt + (16 * (rt0 + rt) + 4 * kq + i) * VP + 16 * w4 + l15);
        }
        __syncthreads();
        if (more && !(PIT_FOLD_DBG & 8)) {
            wstore<BF, H, NTB>(pt, PP, g, tid, pr);
            wstore<BF, H, NTB>(qt, PP, g, tid, qr);
            dpark<BF, NTB>(dt, tid, dr);
        }
        __syncthreads();
    }
    // the slab's sums leave once, through LDS (the tiles are free now): ex[slot][col * H + h] is the head-interleaved row of d_vw, and
    // the workgroup then moves whole rows - plain 16-byte stores into the slab's own tile (fold_reduce_kernel adds the few tiles that
    // hold a key, in a fixed order: the same bits on every run; slots beyond the union are written too and never read), or fp32
    // atomic adds into d_vw[b][key(slot)]
    if ((PIT_FOLD_DBG & 4) && accT[0][0] != 123.456f) return;
    constexpr int EW = CW * H, EP = EW + 4;
    float* ex = reinterpret_cast<float*>(smem_raw);     // [EU][EP]
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) ex[(16 * (mt0 + mt) + 4 * kq + i) * EP + (16 * w4 + l15) * H + hh] = accT[mt][i];
    if (want_scale) {
        const double sum = wave_sum_d(part);
        if (lane == 0) wred[wave] = sum;
    }
    __syncthreads();
    if (g.tiles) {
#pragma unroll
        for (int u = 0; u < EU * EW / 4 / NTB; ++u) {
            const int e = tid + NTB * u, sl = e / (EW / 4), q = e % (EW / 4);
            *reinterpret_cast<float4*>(g.tiles + ((long)(b * g.p.n_slabs + slab) * EU + sl) * ((long)g.dim * H) + (long)chunk * EW + 4 * q) =
                *reinterpret_cast<const float4*>(ex + sl * EP + 4 * q);
        }
    } else if (g.d_vw) {
        // (one float per lane: a wavefront's atomic instruction adds to 256 contiguous bytes of one key row - as 16-byte pieces per
        // thread each instruction touched every fourth word of 1 KB: 639 us at Darcy b=256 against 280)
        float* dst = g.d_vw + (long)b * g.dvw_bstride + (long)chunk * EW;
#pragma unroll
        for (int u = 0; u < EU * EW / NTB; ++u) {
            const int e = tid + NTB * u, sl = e / EW, c = e % EW;
            if (sl < nk) atomicAdd(dst + (long)keys_s[sl] * g.ld_dvw + c, ex[sl * EP + c]);
        }
    }
    if (want_scale && tid < H) {
        double tot = 0.0;
        if (H == 2) { for (int w = 0; w < 4; ++w) tot += wred[4 * tid + w]; }
        else { for (int w = 0; w < 8; ++w) tot += wred[w]; }
        atomicAdd(g.dscale + (long)tid * PIT_DSCALE_SLOTS + ((int)blockIdx.x & (PIT_DSCALE_SLOTS - 1)), -tot);
    }
}

// d_vw[b][key][:] = sum of the tile rows that hold `key` (a key sits in the unions of ~4 tall slabs), in CSR order.  width / 4
// threads (<= 128) share a key with a float4 each, 128 / (width / 4) keys per workgroup (wider rows: a grid-stride loop).
__global__ __launch_bounds__(128) void fold_reduce_kernel(FoldArgs g, int n_head) {
    const long width = (long)g.dim * n_head;
    const int tpk = (int)(width / 4 < 128 ? width / 4 : 128), kpw = 128 / tpk;
    const int sub = (int)threadIdx.x / tpk, t = (int)threadIdx.x - sub * tpk;
    const long row = (long)blockIdx.x * kpw + sub;               // (sample, key) pair
    if (sub >= kpw || row >= (long)g.batch * g.p.n_in) return;
    const int b = (int)(row / g.p.n_in), key = (int)(row - (long)b * g.p.n_in);
    const int beg = g.rev_ptr[key], end = g.rev_ptr[key + 1];
    const float* tb = g.tiles + (long)b * g.p.n_slabs * EU * width;
    float* dst = g.d_vw + (long)b * g.dvw_bstride + (long)key * g.ld_dvw;
    for (long c = 4L * t; c < width; c += 4L * tpk) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = beg; e < end; ++e) {
            const float4 v = *reinterpret_cast<const float4*>(tb + (long)g.rev_ent[e] * width + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + c) = acc;
    }
}

size_t fold_smem(int H, bool bf, int um, bool bwd) {
    const int es = bf ? 2 : 4, pade = bf ? 8 : 4, umk = bf ? ((um + 31) & ~31) : um;
    const size_t vt = (size_t)H * umk * (CW + pade) * es, wt = (size_t)H * SR * (umk + pade) * es;
    if (!bwd) return vt + wt + (size_t)SR * (CW + 4) * 4;
    return std::max(vt + 2 * wt + (size_t)SR * (CW + pade) * es + 8 * 8 + EU * 4, (size_t)EU * (CW * H + 4) * 4);   // (the epilogue's exchange tile)
}

bool fold_plan_ok(const pit_slab_plan* p) {
    if (!p || !p->stats || !p->idx || !p->cnt || !p->m || !p->slot || !p->keys || !p->nkeys) return false;
    if (p->rows != 64 && p->rows != 128 && p->rows != 256) return false;
    return p->n_out > 0 && p->n_in > 0 && p->cap > 0 && p->cap <= 64 && p->umax == EU && p->n_slabs == (p->n_out + p->rows - 1) / p->rows;
}
int union_slots(int max_union) { return max_union <= 32 ? 32 : (max_union <= 48 ? 48 : 64); }

// ------------------------------------------------------------------------------------------------ thin tail
// y[m][o] = sum_n gelu(z[m][n] + b1[n]) w2[o][n] + b2[o], o < n2 <= 4.  tpr = n1 / 4 lanes share a row (16 / 32 / 64), each
// owning four adjacent columns; four row groups per wavefront in flight, at most 128 registers (four waves per SIMD: the kernels
// are an erf per element beside a stream - first version, 135-148 registers and 512 workgroups: 42 / 68 us for the 21 M elements
// of Vorticity's decoder at 1.0 / 1.3 TB/s).  FAST: the polynomial normal CDF (pit_common.h) - the bf16 math mode only.
struct TailArgs {
    const void* z; long ldz; int rows, n1, n2;
    const float *b1, *w2, *b2;
    float* y; long ldy;
    const float* d_y; long ld_dy;
    void* dz; long ld_dz;
    float* acc;                       // backward: TAIL_SLOTS x ((1 + TMAX) n1 + TMAX) partial sums (zero on entry)
    float *d_b1, *d_w2, *d_b2;        // finishing launch
};
constexpr int TMAX = 4, TAIL_SLOTS = 16;

__device__ __forceinline__ float seg_sum(float v, int tpr) {
    v += dpp_f<0xB1, 0xf>(v);
    v += dpp_f<0x4E, 0xf>(v);
    v += dpp_f<0x124, 0xf>(v);
    v += dpp_f<0x128, 0xf>(v);
    if (tpr >= 32) v += __shfl_xor(v, 16, 64);
    if (tpr >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <bool Z16>
__device__ __forceinline__ float4 tail_load(const void* z, long i, bool ok) {
    return Z16 ? ldh4_if(z, i, ok) : ldg4_if(reinterpret_cast<const float*>(z), i, ok);
}
template <bool FAST>
__device__ __forceinline__ void gelu_pair(float t, float& a, float& gp) {
    if (FAST) {
        float e;
        const float cdf = normal_cdf_fast(t, e);
        a = t * cdf;
        gp = fmaf(t * 0.39894228040143267794f, e, cdf);
    } else {
        const float cdf = 0.5f * (1.0f + erff(t * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * __expf(-0.5f * t * t);
        a = t * cdf;
        gp = cdf + t * pdf;
    }
}
template <bool FAST>
__device__ __forceinline__ float gelu_one(float t) {
    if (FAST) { float e; return t * normal_cdf_fast(t, e); }
    return gelu_erf(t);
}

template <int N2, bool FAST, bool Z16>
__global__ __launch_bounds__(256, 4) void thin_tail_fwd_kernel(TailArgs g) {
    constexpr int U = 4;
    const int tpr = g.n1 / 4, rpw = 64 / tpr;
    const int lane = threadIdx.x & 63, q = lane % tpr, sub = lane / tpr, k = 4 * q;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const float4 bq = *reinterpret_cast<const float4*>(g.b1 + k);
    float4 w[N2];
    float b2v[N2];
#pragma unroll
    for (int o = 0; o < N2; ++o) {
        w[o] = o < g.n2 ? *reinterpret_cast<const float4*>(g.w2 + (long)o * g.n1 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        b2v[o] = o < g.n2 ? g.b2[o] : 0.0f;
    }
    const long stride = nwaves * rpw;
    for (long mbase = wave0 * rpw; mbase < g.rows; mbase += U * stride) {
        float4 zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long m = mbase + u * stride + sub;
            zv[u] = tail_load<Z16>(g.z, m * g.ldz + k, m < g.rows);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (mbase + u * stride >= g.rows) break;             // wave-uniform
            const long m = mbase + u * stride + sub;
            const float a0 = gelu_one<FAST>(zv[u].x + bq.x), a1 = gelu_one<FAST>(zv[u].y + bq.y);
            const float a2 = gelu_one<FAST>(zv[u].z + bq.z), a3 = gelu_one<FAST>(zv[u].w + bq.w);
#pragma unroll
            for (int o = 0; o < N2; ++o) {
                float s = (a0 * w[o].x + a1 * w[o].y) + (a2 * w[o].z + a3 * w[o].w);
                s = seg_sum(s, tpr);
                if (q == 0 && m < g.rows && o < g.n2) g.y[m * g.ldy + o] = s + b2v[o];
            }
        }
    }
}

// dz[m][n] = (sum_o d_y[m][o] w2[o][n]) gelu'(z[m][n] + b1[n]);  d_b1 = column sums of dz, d_w2[o] = sum_m d_y[m][o] gelu(..)[m],
// d_b2[o] = sum_m d_y[m][o]: per-lane partial sums over the rows the lane visits, reduced through LDS, then ONE atomic per
// output element and workgroup into one of 16 slots of a scratch (adds to one address serialise at ~40 ns each: straight into
// the 512 gradient words, 512 workgroups were a 20 us tail), which thin_tail_finish_kernel drains into the gradients and clears.
template <int N2, bool FAST, bool Z16>
__global__ __launch_bounds__(256, N2 == 1 ? 4 : 3) void thin_tail_bwd_kernel(TailArgs g) {
    constexpr int U = 4;
    __shared__ float red[16 * 64 * (1 + N2)];
    const int tpr = g.n1 / 4, rpw = 64 / tpr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane % tpr, sub = lane / tpr, k = 4 * q;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const float4 bq = *reinterpret_cast<const float4*>(g.b1 + k);
    float4 w[N2];
#pragma unroll
    for (int o = 0; o < N2; ++o) w[o] = o < g.n2 ? *reinterpret_cast<const float4*>(g.w2 + (long)o * g.n1 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    float sb1[4] = {0.f, 0.f, 0.f, 0.f}, sw2[N2][4], sb2[N2];
#pragma unroll
    for (int o = 0; o < N2; ++o) { sb2[o] = 0.f; sw2[o][0] = sw2[o][1] = sw2[o][2] = sw2[o][3] = 0.f; }
    const long stride = nwaves * rpw;
    const __amdgpu_buffer_rsrc_t rdy = wide_rsrc(g.d_y);
    for (long mbase = wave0 * rpw; mbase < g.rows; mbase += U * stride) {
        float4 zv[U];
        float dy[U][N2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long m = mbase + u * stride + sub;
            const bool ok = m < g.rows;
            zv[u] = tail_load<Z16>(g.z, m * g.ldz + k, ok);
#pragma unroll
            for (int o = 0; o < N2; ++o) dy[u][o] = buf_load(rdy, (ok && o < g.n2) ? (unsigned)((m * g.ld_dy + o) * 4) : OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (mbase + u * stride >= g.rows) break;
            const long m = mbase + u * stride + sub;
            const float t[4] = {zv[u].x + bq.x, zv[u].y + bq.y, zv[u].z + bq.z, zv[u].w + bq.w};
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < N2; ++o) {
                s[0] += dy[u][o] * w[o].x; s[1] += dy[u][o] * w[o].y; s[2] += dy[u][o] * w[o].z; s[3] += dy[u][o] * w[o].w;
            }
            float dzv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a, gp;
                gelu_pair<FAST>(t[c], a, gp);          // (rows beyond the end: dy = 0, nothing accumulates)
                dzv[c] = s[c] * gp;
                sb1[c] += dzv[c];
#pragma unroll
                for (int o = 0; o < N2; ++o) sw2[o][c] += dy[u][o] * a;
            }
#pragma unroll
            for (int o = 0; o < N2; ++o) sb2[o] += dy[u][o];
            if (m < g.rows) {
                const float4 v = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                if (Z16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.dz) + m * g.ld_dz + k) = pack4_bf16(v);
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.dz) + m * g.ld_dz + k) = v;
            }
        }
    }
    // reduce over the workgroup's 4 * rpw row groups: quantity j (0 = b1, 1 + o = w2[o]) of column 4 q + c at
    // red[(grp * (1 + N2) + j) * n1 + 4 q + c]; thread t then sums a column of a quantity over the groups
    const int grp = wave * rpw + sub, ngrp = 4 * rpw;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        red[(grp * (1 + N2) + 0) * g.n1 + k + c] = sb1[c];
#pragma unroll
        for (int o = 0; o < N2; ++o) red[(grp * (1 + N2) + 1 + o) * g.n1 + k + c] = sw2[o][c];
    }
    __syncthreads();
    float* acc = g.acc + (long)((int)blockIdx.x & (TAIL_SLOTS - 1)) * ((1 + TMAX) * g.n1 + TMAX);
    for (int e = threadIdx.x; e < (1 + g.n2) * g.n1; e += 256) {
        const int j = e / g.n1, col = e - j * g.n1;
        float v = 0.0f;
        for (int gq = 0; gq < ngrp; ++gq) v += red[(gq * (1 + N2) + j) * g.n1 + col];
        atomicAdd(acc + j * g.n1 + col, v);
    }
    // d_b2: the lanes with q == 0 hold their rows' sums (the other lanes of a row saw the same d_y)
#pragma unroll
    for (int o = 0; o < N2; ++o) {
        const float v = wave_sum(q == 0 ? sb2[o] : 0.0f);
        if (lane == 0 && o < g.n2) atomicAdd(acc + (1 + TMAX) * g.n1 + o, v);
    }
}

// slots -> gradients (ADDED), scratch left zero for the next call
__global__ __launch_bounds__(256) void thin_tail_finish_kernel(TailArgs g) {
    const int per = (1 + TMAX) * g.n1 + TMAX;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int nw = (1 + g.n2) * g.n1;
    int src = -1;
    float* dst = nullptr;
    if (e < nw) {
        const int j = e / g.n1, col = e - j * g.n1;
        src = e;
        dst = j == 0 ? g.d_b1 + col : g.d_w2 + (long)(j - 1) * g.n1 + col;
    } else if (e < nw + g.n2) {
        src = (1 + TMAX) * g.n1 + (e - nw);
        dst = g.d_b2 + (e - nw);
    }
    if (src < 0) return;
    float v = 0.0f;
#pragma unroll
    for (int sl = 0; sl < TAIL_SLOTS; ++sl) { v += g.acc[(long)sl * per + src]; g.acc[(long)sl * per + src] = 0.0f; }
    *dst += v;
}

bool aligned16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pit_fold_supported(int n_head, int dim, int batch, int rows_per_sample, int n_in) {
    if ((n_head != 1 && n_head != 2) || dim < 64 || dim % 64 != 0 || batch <= 0 || rows_per_sample <= 0 || n_in <= 0) return 0;
    // 32-bit buffer offsets: every tensor a launch addresses stays below 2 GiB
    return (long)batch * rows_per_sample * dim * 4 < (1L << 31) - 65536 && (long)batch * n_in * n_head * dim * 4 < (1L << 31) - 65536;
}

namespace {
int fold_fill(FoldArgs& g, const pit_slab_plan* plan, const float* vw, long ld_vw, long vw_bstride, int batch, int n_head, int dim,
              const float* pw, int max_union) {
    if (!fold_plan_ok(plan) || !vw || !pw) return PIT_ERR_NULL;
    if (!pit_fold_supported(n_head, dim, batch, plan->n_out, plan->n_in) || max_union < 1 || max_union > EU) return PIT_ERR_UNSUPPORTED;
    if (ld_vw % 4 || vw_bstride % 4 || !aligned16p(vw) || !aligned16p(pw) || ld_vw < (long)n_head * dim) return PIT_ERR_SIZE;
    g = FoldArgs();
    g.p = *plan; g.um = union_slots(max_union); g.batch = batch; g.dim = dim; g.chunks = dim / CW;
    g.vw = vw; g.ld_vw = ld_vw; g.vw_bstride = vw_bstride; g.pw = pw;
    return 0;
}
unsigned fold_grid(const FoldArgs& g) { return (unsigned)(8 * ((g.p.n_slabs + 7) / 8) * g.batch * g.chunks); }
}  // namespace

extern "C" int pit_fold_att_fwd(const pit_slab_plan* plan, const float* vw, long ld_vw, long vw_bstride, int batch, int n_head, int dim,
                                const void* pw_, void* z, long ld_z, long z_bstride, int max_union, int math_mode, void* stream) {
    FoldArgs g;
    const float* pw = reinterpret_cast<const float*>(pw_);
    if (int rc = fold_fill(g, plan, vw, ld_vw, vw_bstride, batch, n_head, dim, pw, max_union)) return rc;
    if (!z) return PIT_ERR_NULL;
    const int mode = math_mode & 0xff, z16 = (math_mode & PIT_IO_OUT_BF16) ? 1 : 0;
    if ((mode != PIT_MATH_FP32 && mode != PIT_MATH_BF16) || (math_mode & ~(0xff | PIT_IO_OUT_BF16))) return PIT_ERR_UNSUPPORTED;
    if (ld_z % 4 || z_bstride % 4 || (reinterpret_cast<uintptr_t>(z) & (z16 ? 7 : 15)) || ld_z < dim) return PIT_ERR_SIZE;
    g.z = z; g.ld_z = ld_z; g.z_bstride = z_bstride; g.z16 = z16;
    const bool bf = mode == PIT_MATH_BF16;
    if (bf) g.pw16 = reinterpret_cast<const unsigned short*>(pw);            // (bf16 math mode: pw IS pit_fold_weights' bf16 tile tensor)
    const size_t sm = fold_smem(n_head, bf, g.um, false);
    const dim3 grid(fold_grid(g));
    hipStream_t s = (hipStream_t)stream;
    if (n_head == 1) { if (bf) hipLaunchKernelGGL((fold_fwd_kernel<1, true, 256>), grid, dim3(256), sm, s, g); else hipLaunchKernelGGL((fold_fwd_kernel<1, false, NTB>), grid, dim3(NTB), sm, s, g); }
    else { if (bf) hipLaunchKernelGGL((fold_fwd_kernel<2, true, 256>), grid, dim3(256), sm, s, g); else hipLaunchKernelGGL((fold_fwd_kernel<2, false, NTB>), grid, dim3(NTB), sm, s, g); }
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_fold_att_bwd(const pit_slab_plan* plan, const float* vw, long ld_vw, long vw_bstride, int batch, int n_head, int dim,
                                const void* pw_, const void* qw_, const void* dz, long ld_dz, long dz_bstride,
                                float* d_vw, long ld_dvw, long dvw_bstride, double* dscale,
                                float* tiles, const int* rev_ptr, const int* rev_ent,
                                int max_union, int math_mode, void* stream) {
    FoldArgs g;
    const float *pw = reinterpret_cast<const float*>(pw_), *qw = reinterpret_cast<const float*>(qw_);
    if (int rc = fold_fill(g, plan, vw, ld_vw, vw_bstride, batch, n_head, dim, pw, max_union)) return rc;
    if (!qw || !dz || (!d_vw && !dscale)) return PIT_ERR_NULL;
    const int mode = math_mode & 0xff, dz16 = (math_mode & PIT_IO_DOUT_BF16) ? 1 : 0;
    if ((mode != PIT_MATH_FP32 && mode != PIT_MATH_BF16) || (math_mode & ~(0xff | PIT_IO_DOUT_BF16))) return PIT_ERR_UNSUPPORTED;
    if (!aligned16p(qw) || ld_dz % 4 || dz_bstride % 4 || (reinterpret_cast<uintptr_t>(dz) & (dz16 ? 7 : 15)) || ld_dz < dim) return PIT_ERR_SIZE;
    if (d_vw && (ld_dvw < (long)n_head * dim)) return PIT_ERR_SIZE;
    g.qw = qw; g.dz = dz; g.ld_dz = ld_dz; g.dz_bstride = dz_bstride; g.dz16 = dz16;
    g.d_vw = d_vw; g.ld_dvw = ld_dvw; g.dvw_bstride = dvw_bstride; g.dscale = dscale;
    if (tiles && d_vw) {
        if (!rev_ptr || !rev_ent) return PIT_ERR_NULL;
        if (!aligned16p(tiles) || !aligned16p(d_vw) || ld_dvw % 4 || dvw_bstride % 4) return PIT_ERR_SIZE;
        g.tiles = tiles; g.rev_ptr = rev_ptr; g.rev_ent = rev_ent;
    }
    const bool bf = mode == PIT_MATH_BF16;
    if (bf) { g.pw16 = reinterpret_cast<const unsigned short*>(pw); g.qw16 = reinterpret_cast<const unsigned short*>(qw); }
    const size_t sm = fold_smem(n_head, bf, g.um, true);
    const dim3 grid(fold_grid(g));
    hipStream_t s = (hipStream_t)stream;
    if (n_head == 1) { if (bf) hipLaunchKernelGGL((fold_bwd_kernel<1, true>), grid, dim3(NTB), sm, s, g); else hipLaunchKernelGGL((fold_bwd_kernel<1, false>), grid, dim3(NTB), sm, s, g); }
    else { if (bf) hipLaunchKernelGGL((fold_bwd_kernel<2, true>), grid, dim3(NTB), sm, s, g); else hipLaunchKernelGGL((fold_bwd_kernel<2, false>), grid, dim3(NTB), sm, s, g); }
    PIT_CHECK_LAUNCH();
    if (g.tiles) {
        const long width = (long)dim * n_head;
        const int tpk = (int)(width / 4 < 128 ? width / 4 : 128), kpw = 128 / tpk;
        hipLaunchKernelGGL(fold_reduce_kernel, dim3((unsigned)(((long)batch * plan->n_in + kpw - 1) / kpw)), dim3(128), 0, s, g, n_head);
        PIT_CHECK_LAUNCH();
    }
    return 0;
}

namespace {
int tail_fill(TailArgs& g, const void* z, long ldz, int rows, int n1, int n2, const float* b1, const float* w2, int z_bf16) {
    if (!z || !b1 || !w2) return PIT_ERR_NULL;
    if (rows <= 0 || n2 < 1 || n2 > TMAX || (n1 != 64 && n1 != 128 && n1 != 256) || ldz < n1 || ldz % 4) return PIT_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(z) & (z_bf16 ? 7 : 15)) || !aligned16p(b1) || !aligned16p(w2)) return PIT_ERR_SIZE;
    if ((long)rows * ldz * 4 >= (1L << 31) - 65536) return PIT_ERR_UNSUPPORTED;
    g = TailArgs();
    g.z = z; g.ldz = ldz; g.rows = rows; g.n1 = n1; g.n2 = n2; g.b1 = b1; g.w2 = w2;
    return 0;
}
// N2 (1 or 4 at compile time) x the polynomial CDF (bf16 math mode) x bf16-stored z
#define PIT_TAIL_DISPATCH(KERNEL_, n2_, fast_, z16_, ...)                                                        \
    do {                                                                                                          \
        if ((n2_) == 1) {                                                                                         \
            if (fast_) { if (z16_) hipLaunchKernelGGL((KERNEL_<1, true, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL_<1, true, false>), __VA_ARGS__); }    \
            else { if (z16_) hipLaunchKernelGGL((KERNEL_<1, false, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL_<1, false, false>), __VA_ARGS__); }         \
        } else {                                                                                                  \
            if (fast_) { if (z16_) hipLaunchKernelGGL((KERNEL_<4, true, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL_<4, true, false>), __VA_ARGS__); }    \
            else { if (z16_) hipLaunchKernelGGL((KERNEL_<4, false, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL_<4, false, false>), __VA_ARGS__); }         \
        }                                                                                                         \
    } while (0)
}  // namespace

extern "C" int pit_thin_tail_scratch_floats(void) { return TAIL_SLOTS * ((1 + TMAX) * 256 + TMAX); }

extern "C" int pit_thin_tail_fwd(const void* z, long ldz, int rows, int n1, int n2, const float* b1, const float* w2, const float* b2,
                                 float* y, long ldy, int math_mode, void* stream) {
    const int mode = math_mode & 0xff, z16 = (math_mode & PIT_IO_X_BF16) ? 1 : 0;
    if ((mode != PIT_MATH_FP32 && mode != PIT_MATH_BF16) || (math_mode & ~(0xff | PIT_IO_X_BF16))) return PIT_ERR_UNSUPPORTED;
    TailArgs g;
    if (int rc = tail_fill(g, z, ldz, rows, n1, n2, b1, w2, z16)) return rc;
    if (!b2 || !y) return PIT_ERR_NULL;
    if (ldy < n2) return PIT_ERR_SIZE;
    g.b2 = b2; g.y = y; g.ldy = ldy;
    const int rpw = 64 / (n1 / 4);
    const long need = ((long)rows + 4L * rpw * 4 - 1) / (4L * rpw * 4);
    const dim3 grid((unsigned)std::max<long>(1, std::min<long>(need, 4096)));
    PIT_TAIL_DISPATCH(thin_tail_fwd_kernel, n2, mode == PIT_MATH_BF16, z16, grid, dim3(256), 0, (hipStream_t)stream, g);
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_thin_tail_bwd(const void* z, long ldz, int rows, int n1, int n2, const float* b1, const float* w2,
                                 const float* d_y, long ld_dy, void* dz, long ld_dz, float* d_b1, float* d_w2, float* d_b2,
                                 float* scratch, int math_mode, void* stream) {
    const int mode = math_mode & 0xff, z16 = (math_mode & PIT_IO_X_BF16) ? 1 : 0;
    if ((mode != PIT_MATH_FP32 && mode != PIT_MATH_BF16) || (math_mode & ~(0xff | PIT_IO_X_BF16))) return PIT_ERR_UNSUPPORTED;
    TailArgs g;
    if (int rc = tail_fill(g, z, ldz, rows, n1, n2, b1, w2, z16)) return rc;
    if (!d_y || !dz || !d_b1 || !d_w2 || !d_b2 || !scratch) return PIT_ERR_NULL;
    if (ld_dy < n2 || ld_dz < n1 || ld_dz % 4 || (reinterpret_cast<uintptr_t>(dz) & (z16 ? 7 : 15))) return PIT_ERR_SIZE;
    if ((long)rows * ld_dy * 4 >= (1L << 31) - 65536) return PIT_ERR_UNSUPPORTED;
    g.d_y = d_y; g.ld_dy = ld_dy; g.dz = dz; g.ld_dz = ld_dz; g.acc = scratch; g.d_b1 = d_b1; g.d_w2 = d_w2; g.d_b2 = d_b2;
    const int rpw = 64 / (n1 / 4);
    const long need = ((long)rows + 4L * rpw * 4 * 2 - 1) / (4L * rpw * 4 * 2);     // two batches of four row groups per wavefront
    const dim3 grid((unsigned)std::max<long>(1, std::min<long>(need, 2048)));
    hipStream_t s = (hipStream_t)stream;
    PIT_TAIL_DISPATCH(thin_tail_bwd_kernel, n2, mode == PIT_MATH_BF16, z16, grid, dim3(256), 0, s, g);
    PIT_CHECK_LAUNCH();
    hipLaunchKernelGGL(thin_tail_finish_kernel, dim3((unsigned)(((1 + n2) * n1 + n2 + 255) / 256)), dim3(256), 0, s, g);
    PIT_CHECK_LAUNCH();
    return 0;
}
