// Hand-off probe for the persistent latent kernels (round 4): the per-sample producer -> consumer protocol of
// csrc/pit_latent.hip in isolation, at the Darcy b=8 geometry (8 samples x 16 slab workgroups of 512 threads, each slab
// 16 rows x 64 floats = 4 KB, every workgroup then reads its sample's 64 KB).
//   protocol (MI355X_MICROARCH.md, "Valid forms", table row 3): payload stores `sc1` 16 B per lane (whole 128-B lines per
//   wave instruction) -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier -> one lane: agent-scope atomic add on
//   the sample's counter; consumer: one lane polls the counter with an sc1 load (+ s_sleep) -> workgroup barrier -> every
//   load of the handed-off bytes is an sc1 buffer load.
// Checks EVERY word of every phase (stale data = failure), with random per-workgroup delays (uneven load) and an
// L1-warming plain re-read of the PREVIOUS phase's buffer, and prices a hop against a kernel boundary:
//   persistent: P phases in one launch;  launches: the same phase body as P graph-captured launches.
// build: hipcc -O3 --offload-arch=gfx950 -o handoff_probe handoff_probe.hip ; run: ./handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int SLABS = 16, ROWS = 16, D = 64, WG = 512;
constexpr int SAMPLE_FLOATS = SLABS * ROWS * D;          // 16384 floats = 64 KB

__device__ __forceinline__ unsigned word_of(int phase, int sample, int idx) { return (unsigned)(phase * 1000003 + sample * 65537 + idx) * 2654435761u; }

struct Args { unsigned* buf[2]; unsigned* counters; unsigned* errors; int batch, phases, delay_mask, persistent, phase0, mode, linear; };
// mode 0: sc1 payload stores + agent atomic counter (placement-independent); 1: PLAIN payload stores + atomic counter;
// 2: plain payload stores + one plain flag word per producer, polled as one 64-B line (modes 1, 2: same-XCD only)
__device__ __forceinline__ void map_wg(const Args& a, int& sample, int& slab) {
    const int id = blockIdx.x;
    if (a.linear) { sample = id / SLABS; slab = id % SLABS; return; }       // a sample's slabs spread over all 8 XCDs
    const int x = id & 7, k = id >> 3; sample = x + 8 * (k / SLABS); slab = k % SLABS;
}

__device__ __forceinline__ void phase_body(const Args& a, int phase, int sample, int slab, bool wait) {
    const int tid = threadIdx.x;
    unsigned* dst = a.buf[phase & 1] + (long)sample * SAMPLE_FLOATS;
    // uneven load: a pseudo-random delay per (workgroup, phase)
    if (a.delay_mask) {
        const unsigned h = (unsigned)(blockIdx.x * 7919 + phase * 104729) * 2654435761u;
        const int n = (h >> 20) & a.delay_mask;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
    }
    // produce: this slab's 4 KB, 16 B per lane, sc1 (threads 0..255)
    if (tid < 256) {
        const int idx = slab * ROWS * D + tid * 4;
        i32x4 v;
        v.x = (int)word_of(phase, sample, idx); v.y = (int)word_of(phase, sample, idx + 1);
        v.z = (int)word_of(phase, sample, idx + 2); v.w = (int)word_of(phase, sample, idx + 3);
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(dst, 0, SAMPLE_FLOATS * 4, 0x00020000);
        if (a.mode == 0 || a.mode == 3 || a.mode == 5) __builtin_amdgcn_raw_buffer_store_b128(v, r, idx * 4, 0, 16);
        else __builtin_amdgcn_raw_buffer_store_b128(v, r, idx * 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.persistent) {
        if (a.mode >= 2) {
            unsigned* flags = a.counters + 128 + sample * 16;                 // one 64-B line per sample
            if (tid == 0) {
                if (a.mode >= 4) __hip_atomic_store(flags + slab, (unsigned)(phase + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 store
                else flags[slab] = (unsigned)(phase + 1);                     // plain store (L2 of this XCD)
            }
            if (wait) {
                if (tid < 64) {
                    __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 64, 0x00020000);
                    long spins = 0;
                    for (;;) {
                        const unsigned f = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(fr, (tid & 15) * 4, 0, 16);
                        if (__builtin_amdgcn_ballot_w64(f < (unsigned)(phase + 1)) == 0) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 20000000L) { if (tid == 0) atomicAdd(a.errors + 1, 1u); break; }
                    }
                }
                __syncthreads();
            }
        } else {
        if (tid == 0) __hip_atomic_fetch_add(a.counters + sample, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wait) {
            if (tid == 0) {
                const unsigned target = (unsigned)(SLABS * (phase - a.phase0 + 1));
                long spins = 0;
                while (__hip_atomic_load(a.counters + sample, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000L) { atomicAdd(a.errors + 1, 1u); break; }
                }
            }
            __syncthreads();
        }
        }
    }
}

__device__ __forceinline__ void consume(const Args& a, int phase, int sample) {
    // every workgroup reads its sample's whole 64 KB with sc1 loads: 8 x 16 B per thread
    const int tid = threadIdx.x;
    const unsigned* src = a.buf[phase & 1] + (long)sample * SAMPLE_FLOATS;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, SAMPLE_FLOATS * 4, 0x00020000);
    i32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, (u * WG + tid) * 16, 0, 16);
    unsigned bad = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = (u * WG + tid) * 4;
        bad += (unsigned)v[u].x != word_of(phase, sample, idx);
        bad += (unsigned)v[u].y != word_of(phase, sample, idx + 1);
        bad += (unsigned)v[u].z != word_of(phase, sample, idx + 2);
        bad += (unsigned)v[u].w != word_of(phase, sample, idx + 3);
    }
    if (bad) atomicAdd(a.errors, bad);
    // warm this CU's L1 with PLAIN loads of the buffer the next phase's producers are about to overwrite... that would be
    // a hazard only for plain consumers; here it makes sure stale L1 lines exist for the sc1 loads to (not) hit
    const unsigned* nxt = a.buf[(phase + 1) & 1] + (long)sample * SAMPLE_FLOATS;
    unsigned s = nxt[tid] + nxt[tid + 4096];
    if (s == 0x12345u) atomicAdd(a.errors + 2, 1u);
}

__global__ __launch_bounds__(WG) void persistent_kernel(Args a) {
    int sample, slab; map_wg(a, sample, slab);
    if (sample >= a.batch) return;
    for (int p = 0; p < a.phases; ++p) {
        phase_body(a, a.phase0 + p, sample, slab, true);
        consume(a, a.phase0 + p, sample);
    }
    // last arrival of the sample resets its counter (nobody reads it after its own last wait)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.counters + 64 + sample, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == SLABS - 1) {
            __hip_atomic_store(a.counters + sample, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.counters + 64 + sample, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(WG) void produce_kernel(Args a, int phase) {
    int sample, slab; map_wg(a, sample, slab);
    if (sample >= a.batch) return;
    if (phase > a.phase0) consume(a, phase - 1, sample);          // plain kernel boundary orders it: sc1 not needed, kept equal
    phase_body(a, phase, sample, slab, false);
}

int main(int argc, char** argv) {
    const int batch = 8, phases = argc > 1 ? atoi(argv[1]) : 6;
    Args a;
    CK(hipMalloc(&a.buf[0], batch * SAMPLE_FLOATS * 4)); CK(hipMalloc(&a.buf[1], batch * SAMPLE_FLOATS * 4));
    CK(hipMalloc(&a.counters, 4096)); CK(hipMemset(a.counters, 0, 4096));
    CK(hipMalloc(&a.errors, 64)); CK(hipMemset(a.errors, 0, 64));
    a.batch = batch; a.phases = phases; a.phase0 = 0;
    const int grid = 8 * ((batch + 7) / 8) * SLABS;
    int launch_no = 0;                                  // phase numbers stay monotone over the whole run (flag-line modes)
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[6] = {"sc1 payload + atomic counter", "plain payload + atomic counter", "plain payload + plain flag line", "sc1 payload + plain flag line",
                            "plain payload + sc1 flag line", "sc1 payload + sc1 flag line"};
    for (int cfg = 0; cfg < 8; ++cfg)
    for (int delay_mask : {0, 15, 255}) {
        a.mode = cfg >= 6 ? (cfg == 6 ? 0 : 5) : cfg; a.linear = cfg >= 6;
        CK(hipMemset(a.errors, 0, 64));
        if (delay_mask == 0) printf("---- mode %d (%s), %s map\n", a.mode, names[a.mode], a.linear ? "LINEAR (a sample over all XCDs)" : "XCD-local");
        a.delay_mask = delay_mask;
        // ---- persistent
        a.persistent = 1;
        const int reps = delay_mask ? 200 : 2000;
        for (int i = 0; i < 20; ++i) { a.phase0 = (launch_no++) * phases; hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(WG), 0, s, a); }
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) { a.phase0 = (launch_no++) * phases; hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(WG), 0, s, a); }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err[4]; CK(hipMemcpy(err, a.errors, 16, hipMemcpyDeviceToHost));
        printf("delay_mask %3d  persistent: %7.2f us per launch of %d phases = %5.2f us per phase | stale words %u, spin timeouts %u\n",
               delay_mask, ms * 1e3 / reps, phases, ms * 1e3 / reps / phases, err[0], err[1]);
        // ---- the same as one launch per phase, graph-captured
        a.persistent = 0; a.phase0 = 0;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int p = 0; p <= phases; ++p) {
            if (p < phases) hipLaunchKernelGGL(produce_kernel, dim3(grid), dim3(WG), 0, s, a, p);
            else { Args b = a; hipLaunchKernelGGL(produce_kernel, dim3(grid), dim3(WG), 0, s, b, p); }    // (last: consume only + a write nobody reads)
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(err, a.errors, 16, hipMemcpyDeviceToHost));
        printf("delay_mask %3d  launches  : %7.2f us per graph of %d launches = %5.2f us per launch | stale words %u\n",
               delay_mask, ms * 1e3 / reps, phases + 1, ms * 1e3 / reps / (phases + 1), err[0]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
