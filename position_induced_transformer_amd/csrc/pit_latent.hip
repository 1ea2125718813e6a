// Persistent latent-space kernels (round 4): the whole processor of pit.py:114-122 - n_blocks x [posatt.forward on the
// batch-free latent mesh -> kaiming_mlp -> gelu] - as ONE launch per direction, small (latency-bound) regime.
//
// pit_block.hip runs one launch per block (forward) and per block of the backward chain: 8 launches of ~10 us for
// Darcy b=8, each of which starts cold (kernel arguments, E rows and MLP weights fetched behind the launch) and drains
// before the next one starts.  Here a workgroup keeps its (sample, 16-row slab) for ALL blocks and the dependency between
// blocks - block i+1 contracts over every latent point of the SAME sample - is a per-sample hand-off inside the launch:
//
//   producer (a slab's output rows, 4 KB):  16-B stores, whole 128-B lines per wave instruction, `sc1` (write-through)
//       -> every storing wave: s_waitcnt vmcnt(0) -> workgroup barrier -> one lane stores the slab's flag word (`sc1`)
//   consumer (every slab workgroup of that sample):  wave 0 polls the sample's flag line (one sc1 load of `slabs` words,
//       s_sleep between polls, bounded by the 100 MHz clock) -> workgroup barrier -> EVERY load of handed-off bytes is an
//       `sc1` buffer load (never served by this CU's L1, which other CUs' stores do not refresh)
//
// This is the "sc1 payload + sc1 flag" form of MI355X_MICROARCH.md (Valid forms, table row 1): correct for ANY workgroup
// placement (tools/micro/handoff_probe.hip checks every word under uneven load with a sample's slabs on one XCD and
// spread over all eight: 0 stale words; 3.1 us per hop against 3.5-3.9 us for a kernel boundary doing the same work).
// What the launch wins over a boundary is everything that does NOT depend on the hand-off: the next block's E rows, its
// MLP weights and biases are requested BEFORE the wait, the slab's own rows never leave LDS, and there is no grid
// fill / drain between blocks.  All slab workgroups must be co-resident (they wait for each other): the host gate admits
// grids of at most one workgroup per compute unit; every wait is bounded (sync[0] != 0 afterwards = a wait timed out).
//
// Exact fp32 products (PIT_MATH_FP32), the arithmetic of pit_block.hip phase by phase (same MFMA order, same reductions):
// results are bit-identical to the per-block launches.
#include "pit_common.h"
#include "pit_gemm_rd.h"
#include "pit_block_dev.h"

namespace {

#ifdef PIT_STAMPS
// diagnostic build only (tools/block_bench.py): shader-clock stamps of waves 0 and 5 of one workgroup, [wave slot][layer][point]
__device__ unsigned long long pit_latent_stamps[2 * 16 * 16];
#define LSTAMP(layer_, i_) do { if (blockIdx.x == 5 && (threadIdx.x == 0 || threadIdx.x == 320))                          \
        pit_latent_stamps[((threadIdx.x ? 1 : 0) * 16 + (layer_)) * 16 + (i_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LSTAMP(layer_, i_) do { } while (0)
#endif

constexpr int AUX_SC1 = 16;                            // cache-policy bit of the raw-buffer intrinsics on gfx950: sc1
constexpr int SYNC_ERR = 0, SYNC_FAST = 1, SYNC_DONE = 16, SYNC_FLAGS = 128, SYNC_FLAG_STRIDE = 64, SYNC_MAX_SAMPLES = 64;
constexpr int SYNC_XCC = SYNC_FLAGS + SYNC_MAX_SAMPLES * SYNC_FLAG_STRIDE;        // per sample: 2 x 64 words (slab workgroups, helpers)
constexpr int SYNC_XCC_STRIDE = 128;
constexpr unsigned long long WAIT_LIMIT_TICKS = 200000000ull;      // 2 s of the 100 MHz clock per wait

__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX_SC1);
    return make_float4(__int_as_float(q.x), __int_as_float(q.y), __int_as_float(q.z), __int_as_float(q.w));
}
// hand-off payload store: write-through (`sc1`: visible to every XCD), or - `fast`: all consumers of the sample are known to
// run on the producer's XCD - a plain store, which stays in that XCD's L2 where their sc1 loads are served from
__device__ __forceinline__ void st4_handoff(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v, bool fast) {
    i32x4 q;
    q.x = __float_as_int(v.x); q.y = __float_as_int(v.y); q.z = __float_as_int(v.z); q.w = __float_as_int(v.w);
    if (fast) __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)byte_off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ void st2_handoff(float* p, float2 v, bool fast) {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p, 8u);
    i32x2 q; q.x = __float_as_int(v.x); q.y = __float_as_int(v.y);
    if (fast) __builtin_amdgcn_raw_buffer_store_b64(q, r, 0, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(q, r, 0, 0, AUX_SC1);
}
__device__ __forceinline__ void st1_handoff(float* p, float v, bool fast) {
    if (fast) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // global_store_dword sc1
}
__device__ __forceinline__ unsigned my_xcc() { return (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xfu) + 1u; }
// every workgroup that CONSUMES a sample's hand-offs publishes the XCD it runs on (slot = slab, or slabs + slab for a helper)
__device__ __forceinline__ void publish_xcc(unsigned* xcc_line, int slot) {
    if (threadIdx.x == 0) __hip_atomic_store(xcc_line + slot, my_xcc(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// producer, once, after its first wait: are ALL `count` consumers of the sample on this XCD?  (a missing word = not yet
// published = no: the producer then keeps the placement-independent form for the whole launch; the lines are cleared at
// the end of every launch)  Called by all threads; the answer goes through LDS.
__device__ __forceinline__ bool all_on_my_xcd(const unsigned* xcc_line, int count, int* s_flag) {
    if (threadIdx.x < 64) {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xcc_line, (unsigned)count * 4u);
        const unsigned mine = my_xcc();
        bool ok = true;
        for (int c0 = 0; c0 < count; c0 += 64) {
            const int c = c0 + (int)threadIdx.x;
            const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(xr, (c < count ? c : 0) * 4, 0, AUX_SC1);
            ok = ok && v == mine;
        }
        const bool all = __builtin_amdgcn_ballot_w64(!ok) == 0;
        if (threadIdx.x == 0) *s_flag = all ? 1 : 0;
    }
    (void)0;
    __syncthreads();
    return *s_flag != 0;
}

// consumer side: wave 0 polls the sample's flag line until every slab has posted `target`, then the workgroup barrier
__device__ __forceinline__ void wait_sample(unsigned* flags, int slabs, unsigned target, unsigned* err) {
    if (threadIdx.x < 64) {
        const __amdgpu_buffer_rsrc_t fr = make_rsrc(flags, (unsigned)slabs * 4u);
        const unsigned off = (threadIdx.x < (unsigned)slabs ? threadIdx.x : 0u) * 4u;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned f = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(fr, (int)off, 0, AUX_SC1);
            if (__builtin_amdgcn_ballot_w64(f < target) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) {
                if (threadIdx.x == 0) atomicOr(err, 1u);
                break;
            }
        }
    }
    __syncthreads();
}
// producer side, after the payload stores: drain, barrier, one lane posts
__device__ __forceinline__ void post_slab(unsigned* flags, int slab, unsigned value, bool fast) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fast) *reinterpret_cast<volatile unsigned*>(flags + slab) = value;                                  // plain: this XCD's L2
        else __hip_atomic_store(flags + slab, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                // global_store_dword sc1
    }
}
// after a workgroup's LAST wait of the launch: the sample's last arriver clears its flag line for the next launch
// (`waiters` workgroups wait on the sample's line: its slab workgroups, and in the backward its helper workgroups too)
__device__ __forceinline__ void retire_sample(unsigned* flags, unsigned* xcc_line, unsigned* done, int slabs, int waiters) {
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)waiters - 1u) {
            for (int s = 0; s < slabs; ++s) __hip_atomic_store(flags + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int s = 0; s < waiters; ++s) __hip_atomic_store(xcc_line + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------- forward
struct LatentLayer {
    const float *e, *inv;               // this layer: (H, L, L), (H, L)   (pit_block_weights)
    float* xcat;                        // (batch*L, (1+H)*64): columns [0,64) = block input, head columns written here
    const float *w1, *b1, *w2, *b2;     // the block's MLP ((1+H)*64 -> 64 -> 64)
    float *z1, *h, *z2;                 // saved for the backward: (batch*L, 64) each
};
struct LatentFwdArgs {
    int L, batch, n_layers, linear_map, no_fast;
    LatentLayer layer[MAX_LAYERS];
    float* out; long ld_out;            // the last block's output (batch*L, 64), rows ld_out apart
    unsigned* sync;
};

template <int H>
__global__ __launch_bounds__(512) void latent_fwd_kernel(LatentFwdArgs g) {
    constexpr int W = (1 + H) * BD;                     // concat width = K of the first contraction
    constexpr int XP = W + 4, HP = BD + 4;              // LDS pitches
    constexpr int KS = W / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* pk = smem;                                   // [H * PARK_FLOATS]: slot wave*H + head
    float* xs = smem + H * PARK_FLOATS;                 // [16][XP] concat tile; columns [0,64) = the slab's own input rows
    float* hs = xs + 16 * XP;                           // [16][HP] hidden tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int slabs = g.L / 16;
    int b, slab;
    if (!(g.linear_map ? slab_of_linear(blockIdx.x, g.batch, slabs, b, slab) : slab_of_xcd(blockIdx.x, g.batch, slabs, b, slab))) return;
    const int n0 = slab * 16;
    const long m0 = ((long)b * slabs + slab) * 16;      // first row of the slab in the (batch*L) row space
    const int klen = g.L / BW;                          // every wave: its eighth of the keys, all heads
    unsigned* err = g.sync + SYNC_ERR;
    unsigned* done = g.sync + SYNC_DONE + b;
    unsigned* flags = g.sync + SYNC_FLAGS + b * SYNC_FLAG_STRIDE;
    unsigned* xcc_line = g.sync + SYNC_XCC + b * SYNC_XCC_STRIDE;
    __shared__ int s_same;
    bool fast = false;                                  // decided after the first wait (the first hand-off is always write-through)
    publish_xcc(xcc_line, slab);
    const bool mlp_wave = wave < 4;                     // waves 0..3 own the four hidden / output tiles of the MLP
    const int c1 = (wave & 3) * 16 + l15;
    const int n = g.n_layers;

    for (int i = 0; i < n; ++i) {
        const LatentLayer& ly = g.layer[i];
        const bool handed = i > 0;                      // the block input was produced inside this launch
        LSTAMP(i, 0);
        // ---- (1) everything that does not depend on the hand-off: E fragments of the wave's first trip, MLP operands,
        // and for the first block the slab's own input rows
        const float* wp = ly.e + (long)(n0 + l15) * g.L + 4 * kq;
        float4 av[H][2];
#pragma unroll
        for (int hh = 0; hh < H; ++hh)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                av[hh][s] = *reinterpret_cast<const float4*>(wp + (long)hh * g.L * g.L + wave * klen + 16 * s);
        float4 bvw[KS], w2v[4];
        float bias = 0.0f, bias2 = 0.0f;
        if (mlp_wave) {
#pragma unroll
            for (int s = 0; s < KS; ++s) bvw[s] = *reinterpret_cast<const float4*>(ly.w1 + (long)c1 * W + 16 * s + 4 * kq);
#pragma unroll
            for (int s = 0; s < 4; ++s) w2v[s] = *reinterpret_cast<const float4*>(ly.w2 + (long)c1 * BD + 16 * s + 4 * kq);
            bias = ly.b1[c1];
            bias2 = ly.b2[c1];
        }
        float4 xown = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!handed && tid < 256) xown = *reinterpret_cast<const float4*>(ly.xcat + (m0 + (tid >> 4)) * W + 4 * (tid & 15));
        // ---- (2) the sample's rows of the previous block
        if (handed) {
            wait_sample(flags, slabs, (unsigned)i, err);
            if (i == 1 && !g.no_fast) {
                fast = all_on_my_xcd(xcc_line, slabs, &s_same);
                if (fast && tid == 0) atomicAdd(g.sync + SYNC_FAST, 1u);        // telemetry: producers that switched (cumulative)
            }
            if (i == n - 1) retire_sample(flags, xcc_line, done, slabs, slabs);
        }
        LSTAMP(i, 1);
        // ---- (3) O_h = E_h X for all heads against the same value rows (fetched once)
        f32x4_t acc[H][4];
#pragma unroll
        for (int hh = 0; hh < H; ++hh)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[hh][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(ly.xcat + (long)b * g.L * W, (unsigned)((long)g.L * W * 4));
        for (int j0 = wave * klen; j0 < (wave + 1) * klen; j0 += 32) {
            if (j0 != wave * klen) {
#pragma unroll
                for (int hh = 0; hh < H; ++hh)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        av[hh][s] = *reinterpret_cast<const float4*>(wp + (long)hh * g.L * g.L + j0 + 16 * s);
            }
            float4 bv[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    bv[s][m] = ld4_sc1(rx, (unsigned)(((j0 + 16 * s + 4 * kq + m) * W + 4 * l15) * 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) {
                        const float aw = (&av[hh][s].x)[m];
                        const float4 bb = bv[s][m];
                        acc[hh][0] = mfma_16x16x4(aw, bb.x, acc[hh][0]);
                        acc[hh][1] = mfma_16x16x4(aw, bb.y, acc[hh][1]);
                        acc[hh][2] = mfma_16x16x4(aw, bb.z, acc[hh][2]);
                        acc[hh][3] = mfma_16x16x4(aw, bb.w, acc[hh][3]);
                    }
        }
        LSTAMP(i, 2);
#pragma unroll
        for (int hh = 0; hh < H; ++hh) park(pk, wave * H + hh, lane, acc[hh]);
        if (!handed && tid < 256) *reinterpret_cast<float4*>(xs + (tid >> 4) * XP + 4 * (tid & 15)) = xown;
        LSTAMP(i, 3);
        __syncthreads();
        LSTAMP(i, 4);
        // reduce over the key splits, normalise -> concat tile (LDS) and concat buffer (memory: the backward reads it)
        for (int item = tid; item < H * 256; item += 512) {
            const int hh = item >> 8, ii = (item >> 6) & 3, ln = item & 63;
            const int r = 4 * (ln >> 4) + ii, col = 4 * (ln & 15);
            float4 s = parked_sum(pk, hh, BW, H, ii, ln);
            const float rinv = ly.inv[(long)hh * g.L + n0 + r];
            s.x *= rinv; s.y *= rinv; s.z *= rinv; s.w *= rinv;
            *reinterpret_cast<float4*>(xs + r * XP + BD + hh * BD + col) = s;
            *reinterpret_cast<float4*>(ly.xcat + (m0 + r) * W + BD + hh * BD + col) = s;
        }
        LSTAMP(i, 5);
        __syncthreads();
        LSTAMP(i, 6);
        // ---- the block's MLP on the 16 x W tile (the phases of mlp_fwd16_kernel, A operand from LDS)
        if (mlp_wave) {
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(xs + l15 * XP + 16 * s + 4 * kq);
                a0 = mfma_16x16x4(a.x, bvw[s].x, a0);
                a1 = mfma_16x16x4(a.y, bvw[s].y, a1);
                a0 = mfma_16x16x4(a.z, bvw[s].z, a0);
                a1 = mfma_16x16x4(a.w, bvw[s].w, a1);
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int r = 4 * kq + ii;
                const float z = a0[ii] + a1[ii] + bias;
                const float hv = gelu_erf(z);
                hs[r * HP + c1] = hv;
                ly.z1[(m0 + r) * BD + c1] = z;
                ly.h[(m0 + r) * BD + c1] = hv;
            }
        }
        LSTAMP(i, 7);
        __syncthreads();
        LSTAMP(i, 8);
        if (mlp_wave) {
            f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(hs + l15 * HP + 16 * s + 4 * kq);
                o0 = mfma_16x16x4(a.x, w2v[s].x, o0);
                o1 = mfma_16x16x4(a.y, w2v[s].y, o1);
                o0 = mfma_16x16x4(a.z, w2v[s].z, o0);
                o1 = mfma_16x16x4(a.w, w2v[s].w, o1);
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int r = 4 * kq + ii;
                float v = o0[ii] + o1[ii] + bias2;
                ly.z2[(m0 + r) * BD + c1] = v;
                v = gelu_erf(v);
                if (i + 1 < n) xs[r * XP + c1] = v;          // the next block's input rows of this slab stay in LDS
                else g.out[(m0 + r) * g.ld_out + c1] = v;
            }
        }
        LSTAMP(i, 9);
        if (i + 1 < n) {
            __syncthreads();
            LSTAMP(i, 10);
            // hand-off: the slab's 16 x 64 output -> columns [0, 64) of the next block's concat buffer
            if (tid < 256) {
                float* nx = g.layer[i + 1].xcat;
                const __amdgpu_buffer_rsrc_t rn = make_rsrc(nx + m0 * W, 16u * W * 4u);
                const float4 v = *reinterpret_cast<const float4*>(xs + (tid >> 4) * XP + 4 * (tid & 15));
                st4_handoff(rn, (unsigned)(((tid >> 4) * W + 4 * (tid & 15)) * 4), v, fast);
            }
            LSTAMP(i, 11);
            post_slab(flags, slab, (unsigned)(i + 1), fast);
            LSTAMP(i, 12);
        }
    }
}


// ---------------------------------------------------------------------------------------------- backward
// One launch for the whole backward chain of the processor:
//   chain workgroups (one per (sample, slab), ids [0, n_chain)):
//     top:        the LAST block's MLP backward (data path) from d_out -> its rows of d_xcat[n-1]            -> post
//     block i:    wait for the sample's d_xcat[i]; d(values) = residual + sum_h E_h^T (dO_h / rowsum)  (E symmetric), then
//                 block i-1's MLP backward on the slab (dZ2, dZ1 -> scratch for the weight-gradient reductions, dX -> its
//                 rows of d_xcat[i-1])                                                                       -> post
//     block 0:    d(values) is the gradient of the processor's input
//   helper workgroups (one per (sample, slab), ids [n_chain, 2 n_chain)): d(scale) of every block,
//     d c_h -= sum_{n,d} dO_h[n,d] sum_j Q_h[n,j] U[j,d]: the contraction Q_h U only needs FORWARD tensors, so a helper
//     runs it for block i while the chain is still on block i+1; only the final dot with dO waits for the hand-off.
// The weight-gradient reductions of the blocks' MLPs (dW = dZ^T X: sums over ALL rows) read the scratch after the launch.
struct LatentBwdLayer {
    const float *e, *inv, *qw;          // (H, L, L), (H, L), (H, L, L)
    const float* xcat;                  // the block's concat tensor of the forward (columns [0,64): the attention's values)
    float* dxc;                         // (rows, (1+H)*64): gradient of that concat tensor (produced and consumed in the launch)
    const float *w1, *w2, *z1, *z2;     // the block's MLP: weights and saved pre-activations
    float* scratch;                     // (rows*64) dZ1 | (rows*64) dZ2 of the block's MLP (pit_mlp_bwd_data's layout)
    double* dscale;                     // the block's d(scale) accumulators (n_head * PIT_DSCALE_SLOTS) or null
    const float* h;                     // the MLP's saved hidden activations (rows, 64)
    float *d_w1, *d_b1, *d_w2, *d_b2;   // its weight gradients, ACCUMULATED here by the helpers (d_w1 null: not in this launch)
};
struct LatentBwdArgs {
    int L, batch, n_layers, linear_map, no_fast;
    LatentBwdLayer layer[MAX_LAYERS];
    const float* d_out; long ld_dout;
    float* d_in; long ld_din;
    unsigned* sync;
    int n_chain;
    int dec_items;                      // > 0: a postponed job (the decoder MLP's weight gradients) the helpers perform first
    pit_detail::DwPair dec;
};

// One work item of a carried pair of weight-gradient reductions inside a PERSISTENT workgroup (dw_pair_body is written for
// launches whose upper waves END before a tile: here every wave stays, so waves 4..7 match the tile's barriers one for one)
__device__ __forceinline__ void persistent_rider_item(const pit_detail::DwPair& w, int id, float* smem) {
    const bool second = id >= w.n1;
    const pit_detail::GemmArgs& ga = second ? w.g2 : w.g1;
    const int lid = second ? id - w.n1 : id;
    if (second ? w.rr2 : w.rr1) {
        const int tx = second ? w.tx2 : w.tx1, tiles = second ? w.tiles2 : w.tiles1, slabs = second ? w.slabs2 : w.slabs1;
        const int slab = lid / tiles, tile = lid % tiles;
        const int kbeg = (int)((long)slab * w.nchunks / slabs) * pit_detail::RR_BK;
        const int kend = min(ga.K, (int)((long)(slab + 1) * w.nchunks / slabs) * pit_detail::RR_BK);
        if (threadIdx.x < 256) {
            gemm_rr_tile<1, 1, pit_detail::RR_BK, false>(ga, tile % tx, tile / tx, kbeg, kend, smem, smem + 2 * pit_detail::RR_BK * 64);
        } else {
            const int nb = 1 + (kend > kbeg ? (kend - kbeg + pit_detail::RR_BK - 1) / pit_detail::RR_BK : 0);
            for (int q = 0; q < nb; ++q) __syncthreads();
        }
    } else {
        const int gx = second ? w.gx2 : w.gx1, gy = second ? w.gy2 : w.gy1;
        gemm_rd_body<1, EPI_ATOMIC>(ga, lid % gx, (lid / gx) % gy, lid / (gx * gy));
    }
    __syncthreads();                                    // the next item reuses the LDS
}

// the weight-gradient reductions of block-MLP `ly` restricted to the rows of ONE sample, as 2 (tx + 1) tile items of half the
// sample's rows each: dW1 += dZ1^T X (tx = W / 64 tiles, the bias gradient as row sums of tile 0), dW2 += dZ2^T H (one tile).
// dZ1 / dZ2 were stored by the sample's chain workgroups in this launch: sc1 loads (SC1A).
template <int W>
__device__ __forceinline__ void sample_rider_item(const LatentBwdLayer& ly, long rows, int L, int b, int item, float* smem) {
    constexpr int TX = W / 64;
    const int tile = item % (TX + 1), khalf = item / (TX + 1);
    pit_detail::GemmArgs ga = pit_detail::GemmArgs();
    ga.a_rs = 1; ga.a_cs = BD; ga.b_cs = 1; ga.M = BD; ga.K = (int)rows; ga.atomic = 1; ga.epi = EPI_ATOMIC;
    ga.a_bytes = (unsigned)(rows * BD * 4);
    int bx;
    if (tile < TX) {                                    // dW1
        ga.A = ly.scratch; ga.B = ly.xcat; ga.b_rs = W; ga.N = W + 1; ga.ones_col = W;
        ga.b_bytes = (unsigned)(rows * W * 4); ga.C = ly.d_w1; ga.ldc = W; ga.C2 = ly.d_b1;
        bx = tile;
    } else {                                            // dW2
        ga.A = ly.scratch + rows * BD; ga.B = ly.h; ga.b_rs = BD; ga.N = BD + 1; ga.ones_col = BD;
        ga.b_bytes = (unsigned)(rows * BD * 4); ga.C = ly.d_w2; ga.ldc = BD; ga.C2 = ly.d_b2;
        bx = 0;
    }
    const int kbeg = b * L + khalf * (L / 2), kend = kbeg + L / 2;
    if (threadIdx.x < 256) {
        gemm_rr_tile<1, 1, pit_detail::RR_BK, false, true>(ga, bx, 0, kbeg, kend, smem, smem + 2 * pit_detail::RR_BK * 64);
    } else {
        const int nb = 1 + (L / 2 + pit_detail::RR_BK - 1) / pit_detail::RR_BK;
        for (int q = 0; q < nb; ++q) __syncthreads();
    }
    __syncthreads();
}

// operands of one MLP's backward data path on a slab, requested ahead of their use
struct MlpBackOperands { float w2v[4][4], z1v[4], w1v[2][4][4]; float2 z2v; };

template <int W>
__device__ __forceinline__ void mlp_back_prefetch(const LatentBwdLayer& ly, long m0, int wave, int l15, int kq, int orow, int ocol,
                                                  MlpBackOperands& op) {
    const int c1 = (wave & 3) * 16 + l15;
    op.z2v = *reinterpret_cast<const float2*>(ly.z2 + (m0 + orow) * BD + ocol);
    if (wave < 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) op.w2v[s][ee] = ly.w2[(long)(16 * s + 4 * kq + ee) * BD + c1];      // B(k,n) = w2[k][n]
#pragma unroll
        for (int i = 0; i < 4; ++i) op.z1v[i] = ly.z1[(m0 + 4 * kq + i) * BD + c1];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = (wave + t * BW) * 16 + l15;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ee = 0; ee < 4; ++ee)
                op.w1v[t][s][ee] = (col < W) ? ly.w1[(long)(16 * s + 4 * kq + ee) * W + col] : 0.0f;
    }
}

// `s` = this thread's two adjacent elements (row orow, columns ocol, ocol + 1) of dY of the MLP:
//   dZ2 = dY * gelu'(Z2) -> scratch + LDS;  dZ1 = (dZ2 W2) * gelu'(Z1) -> scratch + LDS;  dX = dZ1 W1 -> the LDS tile dxt [16][W + 4]
// (the phases of mlp_bwd16_kernel / block_bwd_chain; ends with a barrier: dxt is complete for every thread)
template <int W>
__device__ __forceinline__ void mlp_back(const LatentBwdLayer& ly, long rows, long m0, float2 s, const MlpBackOperands& op,
                                         float* ds2, float* ds1, float* dxt, int wave, int l15, int kq, int orow, int ocol, bool fast) {
    constexpr int P1 = BD + 4, WP = W + 4;
    const int c1 = (wave & 3) * 16 + l15;
    s.x *= gelu_erf_grad(op.z2v.x); s.y *= gelu_erf_grad(op.z2v.y);
    *reinterpret_cast<float2*>(ds2 + orow * P1 + ocol) = s;          // (dZ2 / dZ1 go to memory with the hand-off: hand_off_tile)
    __syncthreads();
    if (wave < 4) {                                     // dZ1 tile of this wave
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float4 a = *reinterpret_cast<const float4*>(ds2 + l15 * P1 + 16 * st + 4 * kq);
            a0 = mfma_16x16x4(a.x, op.w2v[st][0], a0);
            a1 = mfma_16x16x4(a.y, op.w2v[st][1], a1);
            a0 = mfma_16x16x4(a.z, op.w2v[st][2], a0);
            a1 = mfma_16x16x4(a.w, op.w2v[st][3], a1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            const float v = (a0[i] + a1[i]) * gelu_erf_grad(op.z1v[i]);
            ds1[r * P1 + c1] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {                       // dX tiles wave, wave + 8
        const int tile = wave + t * BW;
        if (tile * 16 >= W) break;
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float4 a = *reinterpret_cast<const float4*>(ds1 + l15 * P1 + 16 * st + 4 * kq);
            o0 = mfma_16x16x4(a.x, op.w1v[t][st][0], o0);
            o1 = mfma_16x16x4(a.y, op.w1v[t][st][1], o1);
            o0 = mfma_16x16x4(a.z, op.w1v[t][st][2], o0);
            o1 = mfma_16x16x4(a.w, op.w1v[t][st][3], o1);
        }
        const int col = tile * 16 + l15;
#pragma unroll
        for (int i = 0; i < 4; ++i) dxt[(4 * kq + i) * WP + col] = o0[i] + o1[i];
    }
    __syncthreads();
}

// the slab's dX tile (LDS) -> its 16 rows of d_xcat (contiguous: 16 * W floats), and the MLP's dZ1 / dZ2 tiles -> scratch (the
// helpers' weight-gradient tiles read them in this launch): 16 B per lane, whole lines per wave instruction; then post
template <int W>
__device__ __forceinline__ void hand_off_tile(const float* dxt, float* dxc_rows, const float* ds1, float* dz1_rows, const float* ds2,
                                              float* dz2_rows, unsigned* flags, int slab, unsigned value, bool fast) {
    constexpr int WP = W + 4, P1 = BD + 4, Q = 16 * W / 4;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dxc_rows, 16u * W * 4u);
    for (int q = threadIdx.x; q < Q; q += 512) {
        const int r = q / (W / 4), c4 = q % (W / 4);
        st4_handoff(rd, (unsigned)q * 16u, *reinterpret_cast<const float4*>(dxt + r * WP + 4 * c4), fast);
    }
    {   // 2 x 256 16-B pieces: threads [0, 256) dZ1, [256, 512) dZ2
        const int t = threadIdx.x & 255, r = t >> 4, c4 = t & 15;
        const bool second = threadIdx.x >= 256;
        const __amdgpu_buffer_rsrc_t rz = make_rsrc(second ? dz2_rows : dz1_rows, 16u * BD * 4u);
        st4_handoff(rz, (unsigned)t * 16u, *reinterpret_cast<const float4*>((second ? ds2 : ds1) + r * P1 + 4 * c4), fast);
    }
    post_slab(flags, slab, value, fast);
}

template <int H>
__device__ __forceinline__ void latent_bwd_chain(const LatentBwdArgs& g, float* smem, int b, int slab) {
    constexpr int W = (1 + H) * BD;
    constexpr int P1 = BD + 4, WP = W + 4;
    float* pk = smem;                                   // [PARK_FLOATS]
    float* ds2 = smem + PARK_FLOATS;                    // [16][P1]
    float* ds1 = ds2 + 16 * P1;                         // [16][P1]
    float* dxt = ds1 + 16 * P1;                         // [16][WP]: the slab's rows of the current d_xcat
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int slabs = g.L / 16, n = g.n_layers;
    const long rows = (long)g.batch * g.L;
    const int j0s = slab * 16;
    const long m0 = ((long)b * slabs + slab) * 16;
    const int klen = g.L / BW;
    unsigned* err = g.sync + SYNC_ERR;
    unsigned* done = g.sync + SYNC_DONE + b;
    unsigned* flags = g.sync + SYNC_FLAGS + b * SYNC_FLAG_STRIDE;
    unsigned* xcc_line = g.sync + SYNC_XCC + b * SYNC_XCC_STRIDE;
    __shared__ int s_same;
    bool fast = false;
    publish_xcc(xcc_line, slab);
    // every thread owns two adjacent elements of the slab's 16 x 64 d(values) / dY tile
    const int oi = (tid >> 6) & 3, oln = tid & 63, oh = tid >> 8;
    const int orow = 4 * (oln >> 4) + oi, ocol = 4 * (oln & 15) + 2 * oh;

    MlpBackOperands op;
    {   // ---- top of the chain: the last block's MLP from d_out
        const LatentBwdLayer& top = g.layer[n - 1];
        mlp_back_prefetch<W>(top, m0, wave, l15, kq, orow, ocol, op);
        const float2 dy = *reinterpret_cast<const float2*>(g.d_out + (m0 + orow) * g.ld_dout + ocol);
        mlp_back<W>(top, rows, m0, dy, op, ds2, ds1, dxt, wave, l15, kq, orow, ocol, false);
        hand_off_tile<W>(dxt, top.dxc + m0 * W, ds1, top.scratch + m0 * BD, ds2, top.scratch + (rows + m0) * BD, flags, slab, 1u, false);
    }
    for (int i = n - 1; i >= 0; --i) {
        const LatentBwdLayer& ly = g.layer[i];
        // ---- independent of the hand-off: E rows and 1/rowsum of the wave's first trip, the previous block's MLP operands,
        // the slab's own residual (columns [0, 64) of its rows of d_xcat[i]: still in LDS)
        const float* wp = ly.e + (long)(j0s + l15) * g.L + 4 * kq;
        float4 av[H], sv[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            av[hh] = *reinterpret_cast<const float4*>(wp + (long)hh * g.L * g.L + wave * klen);
            sv[hh] = *reinterpret_cast<const float4*>(ly.inv + (long)hh * g.L + wave * klen + 4 * kq);
        }
        if (i > 0) mlp_back_prefetch<W>(g.layer[i - 1], m0, wave, l15, kq, orow, ocol, op);
        const float2 res = *reinterpret_cast<const float2*>(dxt + orow * WP + ocol);
        wait_sample(flags, slabs, (unsigned)(n - i), err);
        if (i == n - 1 && !g.no_fast) {
            fast = all_on_my_xcd(xcc_line, 2 * slabs, &s_same);                                // (slab workgroups and helpers)
            if (fast && tid == 0) atomicAdd(g.sync + SYNC_FAST, 1u);
        }
        if (i == 0) retire_sample(flags, xcc_line, done, slabs, 2 * slabs);
        // ---- d(values)[j] = sum_h sum_n E_h[j][n] (inv_h[n] dO_h[n]): E symmetric, row j of E is column j
        f32x4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t rd = make_rsrc(ly.dxc + (long)b * g.L * W, (unsigned)((long)g.L * W * 4));
        for (int k0 = wave * klen; k0 < (wave + 1) * klen; k0 += 16) {
            if (k0 != wave * klen) {
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    av[hh] = *reinterpret_cast<const float4*>(wp + (long)hh * g.L * g.L + k0);
                    sv[hh] = *reinterpret_cast<const float4*>(ly.inv + (long)hh * g.L + k0 + 4 * kq);
                }
            }
            float4 bv[H][4];
#pragma unroll
            for (int hh = 0; hh < H; ++hh)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    bv[hh][m] = ld4_sc1(rd, (unsigned)(((k0 + 4 * kq + m) * W + BD + hh * BD + 4 * l15) * 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    const float aw = (&av[hh].x)[m] * (&sv[hh].x)[m];          // (row scale folded into the A operand)
                    const float4 bb = bv[hh][m];
                    acc[0] = mfma_16x16x4(aw, bb.x, acc[0]);
                    acc[1] = mfma_16x16x4(aw, bb.y, acc[1]);
                    acc[2] = mfma_16x16x4(aw, bb.z, acc[2]);
                    acc[3] = mfma_16x16x4(aw, bb.w, acc[3]);
                }
        }
        park(pk, wave, lane, acc);
        __syncthreads();
        float2 s = res;
#pragma unroll
        for (int w = 0; w < BW; ++w) {
            s.x += pk[((w * 4 + 2 * oh) * 4 + oi) * 64 + oln];
            s.y += pk[((w * 4 + 2 * oh + 1) * 4 + oi) * 64 + oln];
        }
        if (i == 0) {
            *reinterpret_cast<float2*>(g.d_in + (m0 + orow) * g.ld_din + ocol) = s;
            return;
        }
        const LatentBwdLayer& prev = g.layer[i - 1];
        mlp_back<W>(prev, rows, m0, s, op, ds2, ds1, dxt, wave, l15, kq, orow, ocol, fast);
        hand_off_tile<W>(dxt, prev.dxc + m0 * W, ds1, prev.scratch + m0 * BD, ds2, prev.scratch + (rows + m0) * BD, flags, slab,
                         (unsigned)(n - i + 1), fast);
    }
}

template <int H>
__device__ __forceinline__ void latent_bwd_helper(const LatentBwdArgs& g, float* smem, int b, int slab) {
    constexpr int W = (1 + H) * BD;
    float* pk = smem;                                   // [H * PARK_FLOATS]
    double* wred = reinterpret_cast<double*>(smem + H * PARK_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int slabs = g.L / 16, n = g.n_layers;
    const int n0 = slab * 16;
    const long m0 = ((long)b * slabs + slab) * 16;
    const int klen = g.L / BW;
    unsigned* err = g.sync + SYNC_ERR;
    unsigned* done = g.sync + SYNC_DONE + b;
    unsigned* flags = g.sync + SYNC_FLAGS + b * SYNC_FLAG_STRIDE;
    unsigned* xcc_line = g.sync + SYNC_XCC + b * SYNC_XCC_STRIDE;
    publish_xcc(xcc_line, slabs + slab);
    const bool own = tid < H * 256;
    const int hh = tid >> 8, ii = (tid >> 6) & 3, ln = tid & 63;
    const int r = 4 * (ln >> 4) + ii, col = 4 * (ln & 15);
    const long rows = (long)g.batch * g.L;
    // the postponed job (no dependency on this launch): dealt over the helpers while the chain runs its first phase
    for (int id = b * slabs + slab; id < g.dec_items; id += g.batch * slabs) persistent_rider_item(g.dec, id, smem);
    for (int i = n - 1; i >= 0; --i) {
        const LatentBwdLayer& ly = g.layer[i];
        const bool active = ly.dscale != nullptr;
        if (active) {
            // S_h = Q_h[slab rows] U: forward tensors only (plain loads), same contraction as the forward's
            f32x4_t acc[H][4];
#pragma unroll
            for (int h2 = 0; h2 < H; ++h2)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[h2][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const float* wp = ly.qw + (long)(n0 + l15) * g.L + 4 * kq;
            const float* vp = ly.xcat + (long)b * g.L * W + 4 * l15;
            for (int j0 = wave * klen; j0 < (wave + 1) * klen; j0 += 32) {
                float4 av[H][2], bv[2][4];
#pragma unroll
                for (int h2 = 0; h2 < H; ++h2)
#pragma unroll
                    for (int s = 0; s < 2; ++s) av[h2][s] = *reinterpret_cast<const float4*>(wp + (long)h2 * g.L * g.L + j0 + 16 * s);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int m = 0; m < 4; ++m) bv[s][m] = *reinterpret_cast<const float4*>(vp + (long)(j0 + 16 * s + 4 * kq + m) * W);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int h2 = 0; h2 < H; ++h2) {
                            const float aw = (&av[h2][s].x)[m];
                            const float4 bb = bv[s][m];
                            acc[h2][0] = mfma_16x16x4(aw, bb.x, acc[h2][0]);
                            acc[h2][1] = mfma_16x16x4(aw, bb.y, acc[h2][1]);
                            acc[h2][2] = mfma_16x16x4(aw, bb.z, acc[h2][2]);
                            acc[h2][3] = mfma_16x16x4(aw, bb.w, acc[h2][3]);
                        }
            }
#pragma unroll
            for (int h2 = 0; h2 < H; ++h2) park(pk, wave * H + h2, lane, acc[h2]);
        }
        wait_sample(flags, slabs, (unsigned)(n - i), err);           // (ends with the workgroup barrier: the parks are complete)
        if (i == 0) retire_sample(flags, xcc_line, done, slabs, 2 * slabs);
        if (active) {
            double part = 0.0;
            if (own) {
                const __amdgpu_buffer_rsrc_t rd = make_rsrc(ly.dxc + m0 * W, 16u * W * 4u);
                const float4 dov = ld4_sc1(rd, (unsigned)((r * W + BD + hh * BD + col) * 4));
                const float4 s = parked_sum(pk, hh, BW, H, ii, ln);
                part = (double)s.x * (double)dov.x + (double)s.y * (double)dov.y + (double)s.z * (double)dov.z + (double)s.w * (double)dov.w;
            }
            part = wave_sum_d(part);
            if (lane == 0) wred[wave] = part;
            __syncthreads();
            if (tid < H) {                              // waves [4h, 4h + 4) hold head h
                double tot = 0.0;
                for (int w = 4 * tid; w < 4 * tid + 4; ++w) tot += wred[w];
                atomicAdd(ly.dscale + (long)tid * PIT_DSCALE_SLOTS + ((int)(m0 >> 4) & (PIT_DSCALE_SLOTS - 1)), -tot);
            }
            __syncthreads();                            // the next block's parks reuse pk / wred
        }
        // this block's MLP: its dZ of THIS sample is complete (same hand-off as d_xcat[i]) - the sample's share of dW
        if (ly.d_w1 != nullptr && slab < 2 * (W / 64 + 1)) sample_rider_item<W>(ly, rows, g.L, b, slab, smem);
    }
}

template <int H>
__global__ __launch_bounds__(512) void latent_bwd_kernel(LatentBwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int slabs = g.L / 16;
    int id = blockIdx.x;
    const bool helper = id >= g.n_chain;
    if (helper) id -= g.n_chain;
    int b, slab;
    if (!(g.linear_map ? slab_of_linear(id, g.batch, slabs, b, slab) : slab_of_xcd(id, g.batch, slabs, b, slab))) return;
    if (helper) latent_bwd_helper<H>(g, smem, b, slab);
    else latent_bwd_chain<H>(g, smem, b, slab);
}

constexpr size_t latent_bwd_smem(int H) {
    return std::max(std::max(((size_t)PARK_FLOATS + 2 * 16 * (BD + 4) + 16 * ((1 + H) * BD + 4)) * sizeof(float),
                             (size_t)H * PARK_FLOATS * sizeof(float) + BW * sizeof(double)),
                    (size_t)4 * pit_detail::RR_BK * 64 * sizeof(float));          // (the riders' tiles: 64 KiB)
}

constexpr size_t latent_fwd_smem(int H) { return ((size_t)H * PARK_FLOATS + 16 * ((1 + H) * BD + 4) + 16 * (BD + 4)) * sizeof(float); }

int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

}  // namespace

// workgroups of the forward / backward launch the current device keeps resident at once (0: no device)
static void latent_capacity(int n_head, long& fwd_slots, long& bwd_slots) {
    fwd_slots = bwd_slots = 0;
    const int cus = device_cus();
    if (cus <= 0) return;
    int of = 0, ob = 0;
    if (n_head == 1) {
        (void)hipFuncSetAttribute((const void*)latent_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)latent_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&of, latent_fwd_kernel<1>, 64 * BW, latent_fwd_smem(1)) != hipSuccess) return;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&ob, latent_bwd_kernel<1>, 64 * BW, latent_bwd_smem(1)) != hipSuccess) return;
    } else {
        (void)hipFuncSetAttribute((const void*)latent_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)latent_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&of, latent_fwd_kernel<2>, 64 * BW, latent_fwd_smem(2)) != hipSuccess) return;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&ob, latent_bwd_kernel<2>, 64 * BW, latent_bwd_smem(2)) != hipSuccess) return;
    }
    // (the occupancy query can be one workgroup high near an SGPR edge - MI355X_MICROARCH.md, Residency: at most 2 are counted on)
    fwd_slots = (long)cus * std::min(of, 2);
    bwd_slots = (long)cus * std::min(ob, 2);
}

// 1 when the persistent latent kernels cover this shape on the current device (include/pit_hip.h): every workgroup of the
// forward launch (one per slab) and of the backward launch (chain + helper per slab) must be resident at once
extern "C" int pit_latent_supported(int n_pts, int n_head, int dim, int batch, int n_layers) {
    if (!pit_block_supported(n_pts, n_head, dim, batch)) return 0;
    if (n_layers < 1 || n_layers > MAX_LAYERS) return 0;
    const int slabs = n_pts / 16;
    if (slabs > SYNC_FLAG_STRIDE || batch > SYNC_MAX_SAMPLES) return 0;
    long fs, bs;
    latent_capacity(n_head, fs, bs);
    const long grid = slab_grid(batch, slabs);
    return grid <= fs && 2 * grid <= bs;
}

#ifdef PIT_STAMPS
extern "C" int pit_latent_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pit_latent_stamps), sizeof(unsigned long long) * 2 * 16 * 16);
}
#endif

extern "C" int pit_latent_fwd(const float* e, const float* inv, int n_pts, int n_head, int dim, int batch, int n_layers,
                              float* const* xcat, const float* const* w1, const float* const* b1, const float* const* w2,
                              const float* const* b2, float* z1, float* h, float* z2, float* out, long ld_out,
                              unsigned* sync, int flags, int math_mode, void* stream) {
    if (!e || !inv || !xcat || !w1 || !b1 || !w2 || !b2 || !z1 || !h || !z2 || !out || !sync) return PIT_ERR_NULL;
    if (math_mode != PIT_MATH_FP32 || !pit_latent_supported(n_pts, n_head, dim, batch, n_layers)) return PIT_ERR_UNSUPPORTED;
    if (ld_out < dim || !aligned16(e)) return PIT_ERR_SIZE;
    LatentFwdArgs g;
    g.L = n_pts; g.batch = batch; g.n_layers = n_layers; g.linear_map = (flags & PIT_LATENT_LINEAR_MAP) ? 1 : 0;
    g.no_fast = (flags & PIT_LATENT_NO_FAST) ? 1 : 0;
    const long rows = (long)batch * n_pts;
    for (int l = 0; l < n_layers; ++l) {
        if (!xcat[l] || !w1[l] || !b1[l] || !w2[l] || !b2[l]) return PIT_ERR_NULL;
        if (!aligned16(xcat[l]) || !aligned16(w1[l]) || !aligned16(w2[l])) return PIT_ERR_SIZE;
        LatentLayer& ly = g.layer[l];
        ly.e = e + (long)l * n_head * n_pts * n_pts; ly.inv = inv + (long)l * n_head * n_pts;
        ly.xcat = xcat[l]; ly.w1 = w1[l]; ly.b1 = b1[l]; ly.w2 = w2[l]; ly.b2 = b2[l];
        ly.z1 = z1 + l * rows * dim; ly.h = h + l * rows * dim; ly.z2 = z2 + l * rows * dim;
    }
    g.out = out; g.ld_out = ld_out; g.sync = sync;
    const dim3 grid((unsigned)slab_grid(batch, n_pts / 16)), block(64 * BW);
    const size_t sm = latent_fwd_smem(n_head);
#define PIT_LATENT_FWD(H_)                                                                                                 \
    do {                                                                                                                   \
        static bool once = ((void)hipFuncSetAttribute((const void*)latent_fwd_kernel<H_>,                                  \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 131072), true);                 \
        (void)once;                                                                                                        \
        hipLaunchKernelGGL((latent_fwd_kernel<H_>), grid, block, sm, (hipStream_t)stream, g);                              \
    } while (0)
    if (n_head == 1) PIT_LATENT_FWD(1); else PIT_LATENT_FWD(2);
#undef PIT_LATENT_FWD
    PIT_CHECK_LAUNCH();
    return 0;
}

extern "C" int pit_latent_bwd(const float* e, const float* inv, const float* qw, int n_pts, int n_head, int dim, int batch, int n_layers,
                              const float* const* xcat, float* const* d_xcat, const float* const* w1, const float* const* w2,
                              const float* z1, const float* z2, float* const* scratch, double* const* dscale,
                              const float* h, float* const* d_w1, float* const* d_b1, float* const* d_w2, float* const* d_b2,
                              const pit_mlp_params_job* rider,
                              const float* d_out, long ld_dout, float* d_in, long ld_din,
                              unsigned* sync, int flags, int math_mode, void* stream) {
    if (!e || !inv || !qw || !xcat || !d_xcat || !w1 || !w2 || !z1 || !z2 || !scratch || !dscale || !d_out || !d_in || !sync) return PIT_ERR_NULL;
    if (math_mode != PIT_MATH_FP32 || !pit_latent_supported(n_pts, n_head, dim, batch, n_layers)) return PIT_ERR_UNSUPPORTED;
    if (ld_dout < dim || ld_din < dim || (ld_dout & 1) || (ld_din & 1) || !aligned16(e) || !aligned16(qw) ||
        (reinterpret_cast<uintptr_t>(d_out) & 7) || (reinterpret_cast<uintptr_t>(d_in) & 7)) return PIT_ERR_SIZE;
    LatentBwdArgs g;
    g.L = n_pts; g.batch = batch; g.n_layers = n_layers; g.linear_map = (flags & PIT_LATENT_LINEAR_MAP) ? 1 : 0;
    g.no_fast = (flags & PIT_LATENT_NO_FAST) ? 1 : 0;
    const long rows = (long)batch * n_pts;
    for (int l = 0; l < n_layers; ++l) {
        if (!xcat[l] || !d_xcat[l] || !w1[l] || !w2[l] || !scratch[l]) return PIT_ERR_NULL;
        if (!aligned16(xcat[l]) || !aligned16(d_xcat[l]) || !aligned16(scratch[l])) return PIT_ERR_SIZE;
        LatentBwdLayer& ly = g.layer[l];
        ly.e = e + (long)l * n_head * n_pts * n_pts; ly.inv = inv + (long)l * n_head * n_pts; ly.qw = qw + (long)l * n_head * n_pts * n_pts;
        ly.xcat = xcat[l]; ly.dxc = d_xcat[l]; ly.w1 = w1[l]; ly.w2 = w2[l];
        ly.z1 = z1 + l * rows * dim; ly.z2 = z2 + l * rows * dim; ly.scratch = scratch[l]; ly.dscale = dscale[l];
        const bool dw = h && d_w1 && d_w1[l];
        if (dw && (!d_b1 || !d_w2 || !d_b2 || !d_b1[l] || !d_w2[l] || !d_b2[l])) return PIT_ERR_NULL;
        ly.h = dw ? h + l * rows * dim : nullptr;
        ly.d_w1 = dw ? d_w1[l] : nullptr; ly.d_b1 = dw ? d_b1[l] : nullptr;
        ly.d_w2 = dw ? d_w2[l] : nullptr; ly.d_b2 = dw ? d_b2[l] : nullptr;
    }
    g.dec_items = 0;
    bool rider_carried = false;
    if (rider && pit_detail::plan_dw_pair(*rider, BW, &g.dec, 64, true)) { g.dec_items = g.dec.n1 + g.dec.n2; rider_carried = true; }
    g.d_out = d_out; g.ld_dout = ld_dout; g.d_in = d_in; g.ld_din = ld_din; g.sync = sync;
    g.n_chain = slab_grid(batch, n_pts / 16);
    const dim3 grid((unsigned)(2 * g.n_chain)), block(64 * BW);
    const size_t sm = latent_bwd_smem(n_head);
#define PIT_LATENT_BWD(H_)                                                                                                 \
    do {                                                                                                                   \
        static bool once = ((void)hipFuncSetAttribute((const void*)latent_bwd_kernel<H_>,                                  \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 131072), true);                 \
        (void)once;                                                                                                        \
        hipLaunchKernelGGL((latent_bwd_kernel<H_>), grid, block, sm, (hipStream_t)stream, g);                              \
    } while (0)
    if (n_head == 1) PIT_LATENT_BWD(1); else PIT_LATENT_BWD(2);
#undef PIT_LATENT_BWD
    PIT_CHECK_LAUNCH();
    if (rider && !rider_carried)                        // too large to ride: the launches pit_mlp_bwd_params would have made
        return pit_mlp_bwd_params(rider->x, rider->ldx, rider->rows, rider->n0, rider->n1, rider->n2, rider->h, rider->out_gelu,
                                  rider->d_y, rider->ld_dy, rider->d_w1, rider->d_b1, rider->d_w2, rider->d_b2, rider->accumulate,
                                  rider->scratch, rider->math_mode, stream);
    return 0;
}
