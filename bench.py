#!/usr/bin/env python3
"""Headline benchmark: PiT forward+backward samples/s on the Darcy2D configuration.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = zero the flat gradient buffer, forward of pit_darcy (train_darcy.py:103-111
hyper-parameters: 43x43 grid -> 16x16 latent, hid 64, 2 heads, 4 blocks, locality 0.02),
fused de-normalise + RelLpNorm(p=2) loss, full backward, and for N > 1 the one all-reduce of
the flat gradient buffer - captured once into a hipGraph and replayed.  Inputs are synthetic
N(0,1) fields resident in HBM before the timed region (the datasets are not in the reference
repo).  Per-GPU batch is fixed (default 8 = train_darcy.py:66), so N GPUs process N x that:
weak scaling.  The optimizer step is excluded from `value` (BASELINE.json's metric is fwd+bwd)
and reported separately as `train_step` (fwd+bwd+Adam).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  "roofline":     the dominant kernel's achieved FLOP/s (algorithmic FLOPs / mean launch time
                  measured here with HIP events on the launch stream) against the fp32 MFMA peak;
  "cpu_baseline": the CPU oracle (a port of the reference's torch-CPU path, validated bit-equal
                  to it in the build container) timed on this host's cores on the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before the first HIP call (see position_induced_transformer_amd/__init__.py: ROCm 7.2 graph packet capture)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# dense, reference-equivalent fwd+bwd GFLOP per sample
# algorithmic activation bytes per sample, forward; HBM3E peak from MI355X_MICROARCH.md
MB_PER_SAMPLE_FWD = {"darcy": 2.10, "burgers": 1.92, "vorticity": 13.08, "elasticity": 19.43, "naca": 10.80}
HBM_PEAK_GBPS = 8000.0
GFLOP_PER_SAMPLE = {"darcy": 0.781, "burgers": 0.643, "vorticity": 9.126, "elasticity": 22.406, "naca": 10.005}


WORKLOADS = {
    "darcy": "darcy2d 43x43 grid->16x16 latent, pit_fixed hid64 H2 blocks4 loc0.02, RelL2 (train_darcy.py:64-111)",
    "burgers": "burgers1d 1024->256 latent, pit_periodic1d hid64 H2 blocks5 loc0.02, RelL1 (train_burgers.py:51-78)",
    "sod": "sod1d 1024->256 latent, pit_fixed hid32 H1 blocks2 (train_sod.py:55-76)",
    "vorticity": "vorticity2d 64x64->16x16 latent, pit_periodic2d hid256 H2 blocks4 + InstanceNorm, RelL2 (train_vorticity.py:65-106)",
    "elasticity": "elasticity 972-point clouds, pit hid256 H2 blocks4, per-sample meshes (train_elasticity.py:56-75)",
    "naca": "naca 120->728->221x51, pit hid128 H1 blocks4, per-sample meshes (train_naca.py:68-89)",
    "cylinder": "cylinder 4390->896->4390, pit_fixed hid256 H1 blocks4 loc0.01 (train_cylinder.py:55-84)",
}
T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (train_darcy.py:66 uses 8)")
    ap.add_argument("--task", default="darcy")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--eager-allreduce", action="store_true",
                    help="N > 1: capture the compute only and issue the gradient all-reduce eagerly after each replay")
    ap.add_argument("--ar-buckets", choices=("auto", "1", "2"), default="auto",
                    help="gradient exchange of a data-parallel step: one all-reduce after the pass (1), or the early bucket "
                         "(decoder + last processor MLPs) reduced on a second stream while the rest of the backward runs (2); "
                         "auto = 1 unless the measured cost of the captured all-reduce exceeds 10 %% of the step, then the faster "
                         "of the two")
    ap.add_argument("--watchdog", type=float, default=300.0, help="N > 1: seconds allowed for capture + first replays")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend; gloo (tests: several ranks on ONE GPU) stages the exchange through the host and "
                         "implies --eager-allreduce")
    ap.add_argument("--device-index", type=int, default=None, help="GPU of this process (default: LOCAL_RANK)")
    ap.add_argument("--sweep-batch", type=int, default=256, help="per-GPU batch of the saturated-regime scaling point (N > 1)")
    ap.add_argument("--ddp-sweep", action="store_true", help="time the saturated-regime scaling point even with --no-extras")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip batch sweep / optimizer / roofline probes")
    ap.add_argument("--cpu-iters", type=int, default=100)
    ap.add_argument("--rollout", type=int, default=0,
                    help="vorticity only: one step = the N-step autoregressive BPTT optimiser step of train_vorticity.py:118-129")
    ap.add_argument("--recompute", action="store_true", help="with --rollout: activation recompute (each step keeps only its input and is re-run in the backward)")
    ap.add_argument("--head-scale-route", choices=("host", "device"), default="host",
                    help="where c = tan(K(1+sin lmda)) of pit.py:48 is evaluated for the timed fwd+bwd step: 'host' = the "
                         "reference's own torch-CPU ops, cached per lmda version (exact, sync-free and capturable while lmda "
                         "is frozen, as it is in a fwd+bwd step); 'device' = inside the kernels (what a captured TRAINING "
                         "step with the optimizer in the graph uses; the train_step extra always does)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block (oracle forward+backward on the host)")
    ap.add_argument("--math", choices=("fp32", "bf16"), default="fp32",
                    help="MFMA math mode of the contractions (pit_set_math_mode); fp32 = the reference's arithmetic")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------
def darcy_affine(device):
    """Synthetic PixelWiseNormalization statistics for the target field (train_darcy.py:78,129):
    std + eps and mean per pixel."""
    g = torch.Generator().manual_seed(1234)
    std = torch.rand(1, 43, 43, 1, generator=g) * 0.5 + 0.75
    mean = torch.randn(1, 43, 43, 1, generator=g) * 0.1
    return (std + 1e-5).to(device), mean.to(device)


def build_step(args, device, rank, world, batch, with_optimizer=False, all_reduce=None):
    from position_induced_transformer_amd import tasks
    from position_induced_transformer_amd.ddp import broadcast_parameters
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task(args.task, device=device, seed=0)
    if world > 1:
        broadcast_parameters(model)
    g = torch.Generator().manual_seed(100 + rank)           # every rank its own shard of the global batch
    mesh_in, func_in, mesh_out, target = sample(batch)
    func_in = torch.randn(func_in.shape, generator=g).to(device)
    target = torch.randn(target.shape, generator=g).to(device)
    affine = darcy_affine(device) if args.task == "darcy" else None
    opt, flat = None, None
    if with_optimizer:                    # Adam(1e-3) + CosineAnnealingLR as train_darcy.py:115-116, fused
        from position_induced_transformer_amd.ddp import FlatAdam, FlatGradients
        flat = FlatGradients(model.parameters(), flatten_params=True)
        opt = FlatAdam(flat, lr=1e-3, cosine_t_max=30 * (1024 // 8), zero_grads=True)
    use_ar = (world > 1) if all_reduce is None else all_reduce
    if args.rollout:
        from position_induced_transformer_amd.engine import RolloutStep
        y = torch.randn(*target.shape[:-1], args.rollout, generator=g).to(device)
        step = RolloutStep(model, (mesh_in, func_in, y), args.rollout, meta["out_dim"], meta["p"], recompute=args.recompute,
                           all_reduce=use_ar, optimizer=opt, flat=flat)
        return step, model, meta
    step = TrainStep(model, (mesh_in, func_in, mesh_out, target), meta["out_dim"], meta["p"], affine,
                     all_reduce=use_ar, optimizer=opt, flat=flat,
                     all_reduce_buckets=int(getattr(args, 'ar_buckets_n', 1)) if (use_ar and opt is None) else 1)
    return step, model, meta


class Watchdog:
    """The all-reduce is captured INSIDE the step graph.  If RCCL and graph replay do not get along on some
    node the symptom is a hang, not an exception: give the first replays a deadline and leave with a clear
    message and a non-zero code instead of stalling the driver (never re-exec: this process owns the GPU;
    rerun with --eager-allreduce, which keeps the collective outside the graph)."""

    def __init__(self, seconds, what):
        import threading
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.what, self.seconds = what, seconds

    def _fire(self):
        print(f"[bench] WATCHDOG: {self.what} did not finish within {self.seconds:.0f} s - the captured RCCL all-reduce "
              "is not completing on this node; rerun with --eager-allreduce", file=sys.stderr, flush=True)
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def prepare(step, use_graph):
    """Returns run() -> None executing one step; captures a hipGraph unless disabled."""
    if not use_graph:
        for _ in range(3):
            step.run_eager()
        return step.run_eager, "eager"
    try:
        step.capture()
        return step.replay, "hipgraph"
    except Exception as exc:                                  # e.g. collective not capturable
        if not step.all_reduce:
            raise
        print(f"[bench] graph capture with all-reduce failed ({type(exc).__name__}: {exc}); "
              "capturing compute only, all-reduce issued eagerly", file=sys.stderr)
        torch.cuda.synchronize()
        step.all_reduce = False
        step.graph = None
        step.capture()

        def run():
            step.replay()
            step.flat.all_reduce()
        return run, "hipgraph+eager-allreduce"


def timed(run, steps, warmup, world):
    """ONE timed block: EXACTLY ``steps`` steps between barrier + synchronize on both sides, MAX over ranks."""
    for _ in range(warmup):
        run()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def timed_blocks(run, steps, warmup, world, min_total=0.25, min_blocks=5, max_blocks=400):
    """The driver's --steps can make one block a few milliseconds long (20 steps = 6.6 ms): the block of
    exactly ``steps`` steps is therefore REPEATED until at least ``min_blocks`` blocks and ``min_total``
    seconds were timed; the reported time is the median block (spread alongside).  Every block is
    bracketed and MAX-reduced like a single one, so all ranks take the same decisions."""
    blocks = [timed(run, steps, warmup, world)]
    while (len(blocks) < min_blocks or sum(blocks) < min_total) and len(blocks) < max_blocks:
        blocks.append(timed(run, steps, 0, world))
    med = statistics.median(blocks)
    return med, {"blocks": len(blocks), "steps_per_block": steps, "min_ms_per_step": round(min(blocks) / steps * 1e3, 4),
                 "max_ms_per_step": round(max(blocks) / steps * 1e3, 4), "total_timed_s": round(sum(blocks), 3)}


# ------------------------------------------------------------------------------------------
def graph_time_us(fn, reps=20, replays=10):
    """Mean duration of one call of ``fn`` (a kernel launch through the C ABI): ``reps`` launches are
    captured into a hipGraph and the replays are timed with HIP events on the launch stream, so the
    figure is device time per launch, free of host launch overhead."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(replays):
        g.replay()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


def pmc_traffic(kernel_key, shape):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    runs - counters cannot be read inside this process -, corrected as MI355X_MICROARCH.md prescribes and calibrated in
    profiles/r03_fetch_calibration.txt).  The committed record is keyed by kernel AND by the shape of the launch it was
    measured on (profiles/r05_pmc_traffic.json: ``shape``): a probe that times another launch of the family gets None,
    never another launch's bytes."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)["kernels"].get(kernel_key)
            if rec is not None and rec.get("shape") == shape:
                return rec["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
    return None


def roofline_mlp_probe(model, batch):
    """The kernel family with the LARGEST share of the Darcy b=8 step (profiles/ step breakdowns) is the forward
    of the pointwise MLP with its bias + erf-GELU epilogues: since round 2 ONE launch per kaiming_mlp forward
    in the small regime (mlp_fwd16_kernel: both contractions fused, 16-row slabs on v_mfma_f32_16x16x4_f32),
    two LDS-tiled GEMM launches in the large regime.  Timed here on the processor MLP ((1+H)*hid -> hid -> hid
    on batch*L_ltt rows, pit.py:103); algorithmic FLOPs 2*rows*(n0*n1 + n1*n2), fp32 MFMA peak."""
    from position_induced_transformer_amd import ops
    mlp = model.mlp[0]
    rows_per_sample = model.mesh_ltt.shape[0] if model.mesh_ltt is not None else 972
    n0, n1, n2 = mlp.mlp1.in_features, mlp.mlp1.out_features, mlp.mlp2.out_features
    x = torch.randn(batch, rows_per_sample, n0, device="cuda")
    with torch.no_grad():
        us = graph_time_us(lambda: ops.mlp_apply(x, mlp.mlp1.weight, mlp.mlp1.bias, mlp.mlp2.weight, mlp.mlp2.bias, True))
    rows = batch * rows_per_sample
    # the library's own rule (csrc/pit_mlp.hip: try_launch_mlp_fwd16)
    fused = n1 in (32, 64, 128) and (n2 <= 4 or (n2 % 16 == 0 and n2 <= n1)) and rows >= 256 and n0 <= 256 and rows * n1 * (n0 + n2) <= (1 << 27)
    # (csrc/pit_mlp_slab.hip: pit_mlp_slab_eligible + _preferred - the large regime at hid 64 is ONE launch too since round 4)
    slab = (not fused) and n1 == 64 and n2 == 64 and n0 % 16 == 0 and n0 <= 256 and rows >= 65536 \
        and ops.get_math_mode() == "fp32" and not os.environ.get("PIT_NO_SLAB_MLP")
    launches = 1 if (fused or slab) else 2
    flops = 2.0 * rows * (n0 * n1 + n1 * n2) / launches
    us_launch = us / launches
    achieved = flops / (us_launch * 1e-6) / 1e12
    alg_bytes = 4.0 * (rows * n0 + 2 * rows * n1 + (0 if (fused or slab) else rows * n1) + 2 * rows * n2 + n0 * n1 + n1 * n2) / launches
    name = f"mlp_fwd16_kernel<{n1},{(n0 + 63) // 64 * 4}> (fused GEMM1+GELU+GEMM2+GELU)" if fused else \
        (f"mlp_fwd64_kernel<{n0 // 16}> (64-row slabs, fused GEMM1+GELU+GEMM2+GELU)" if slab else
         "gemm_lds/gemm_rd kernel<BIAS_GELU> (mean of the two launches)")
    return {"bound": "mfma", "kernel": f"{name}: kaiming_mlp forward {n0}->{n1}->{n2} on {rows} rows, batch {batch}",
            "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(f"mlp_fwd_b{batch}", f"rows{rows}_{n0}_{n1}_{n2}"),
            "us_per_launch": round(us_launch, 3), "flops_per_launch": flops, "algorithmic_bytes": alg_bytes}


def roofline_dw_probe(model, batch):
    """The weight-gradient reductions of the processor MLP in the large regime - dW1 = dZ1^T X, dW2 = dZ2^T H and both bias
    gradients in ONE gemm_rr_kernel launch (pit_mlp_bwd_params): 2*rows*(n0*n1 + n1*n2) FLOPs, algorithmic bytes
    4*rows*(n0 + 2*n1 + n2) (X, dZ1, H, dZ2 read once; the outputs are a few KB), fp32 MFMA peak; timed live through the
    C ABI on operands of the step's shapes."""
    from position_induced_transformer_amd import _lib
    mlp = model.mlp[0]
    rows_per_sample = model.mesh_ltt.shape[0] if model.mesh_ltt is not None else 972
    n0, n1, n2 = mlp.mlp1.in_features, mlp.mlp1.out_features, mlp.mlp2.out_features
    rows = batch * rows_per_sample
    x, h = torch.randn(rows, n0, device="cuda"), torch.randn(rows, n1, device="cuda")
    scratch, dy = torch.randn(rows * (n1 + n2), device="cuda"), torch.randn(rows, n2, device="cuda")
    gw1, gb1 = torch.zeros(n1, n0, device="cuda"), torch.zeros(n1, device="cuda")
    gw2, gb2 = torch.zeros(n2, n1, device="cuda"), torch.zeros(n2, device="cuda")
    L = _lib.lib()

    def call():
        _lib.check(L.pit_mlp_bwd_params(x.data_ptr(), n0, rows, n0, n1, n2, h.data_ptr(), 1, dy.data_ptr(), n2, gw1.data_ptr(),
                                        gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), 1, scratch.data_ptr(), 0, _lib.stream_ptr()),
                   "pit_mlp_bwd_params")
    us = graph_time_us(call)
    flops = 2.0 * rows * (n0 * n1 + n1 * n2)
    achieved = flops / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": f"gemm_rr_kernel<1,1,64>: dW1|db1 + dW2|db2 of kaiming_mlp {n0}->{n1}->{n2} on {rows} rows, batch {batch} "
                                       "(one launch)",
            "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(f"mlp_dw_b{batch}", f"rows{rows}_{n0}_{n1}_{n2}"),
            "us_per_launch": round(us, 3), "flops_per_launch": flops, "algorithmic_bytes": 4.0 * rows * (n0 + 2 * n1 + n2),
            "algorithmic_GBps": round(4.0 * rows * (n0 + 2 * n1 + n2) / (us * 1e-6) / 1e9, 1)}


def roofline_probe(model, batch):
    """The fused position-attention of the processor (posatt_rows_kernel: 4 forward launches + 4 d(scale)
    launches of the same body, plus the transposed posatt_cols_kernel for d(values)): forward launch against
    its algorithmic FLOPs 2*H*N*J*D*b and the fp32 MFMA peak."""
    from position_induced_transformer_amd import ops
    layer = model.conv[0]
    mesh = model.mesh_ltt
    if mesh is None:                     # per-sample meshes: a synthetic cloud of the Elasticity size
        mesh = torch.rand(batch, 972, model.space_dim, device="cuda")
    plan = layer._plan(mesh, mesh, True)
    d = model.hid_dim
    u = torch.randn(batch, plan.n_in, d, device="cuda")
    # what the step runs for this shape (pit.processor): in the large regime of a batch-free mesh the launch reads the weights
    # pit_block_weights formed once per step for ALL blocks (round 4); that launch is outside this probe's timed region
    pre = model.mesh_ltt is not None and ops.pre_weights_supported(plan.n_in, layer.n_head, d, batch)
    with torch.no_grad():
        if pre:
            weights = ops.block_weights(plan, [a.lmda for a in model.conv], layer.n_head, False)
            # the launch exactly as the step makes it: the values ARE the first columns of the block's concat buffer (written there
            # by the producing MLP, pit.processor), the launch adds the head columns - no copy of the values (until round 5 this
            # probe handed over a detached tensor and timed the copying form: 51 us against the step's 44.5 us, profiles/r05_darcy256)
            xcat = torch.randn(batch, plan.n_in, (1 + layer.n_head) * d, device="cuda")

            def launch():
                v = xcat[:, :, :d]
                v._pit_concat = xcat
                return ops.posatt_pre_apply(v, layer.lmda, weights, 0, layer.n_head)
            assert launch().data_ptr() == xcat.data_ptr()
            us = graph_time_us(launch)
        else:
            us = graph_time_us(lambda: ops.posatt_apply(u, layer.lmda, plan, layer.n_head, True))
    flops = 2.0 * layer.n_head * plan.n_out * plan.n_in * d * batch
    achieved = flops / (us * 1e-6) / 1e12
    if pre:      # values read, head columns written, E of every head once
        alg_bytes = 4.0 * batch * plan.n_in * d + 4.0 * batch * plan.n_out * layer.n_head * d + 4.0 * layer.n_head * plan.n_out * plan.n_in
    else:
        alg_bytes = 4.0 * batch * plan.n_in * d + 4.0 * batch * plan.n_out * (1 + layer.n_head) * d
    kname = "posatt_rows_tiles<PRE> (weights of pit_block_weights)" if pre else "posatt_rows_kernel<fwd>"
    return {"bound": "mfma", "kernel": f"{kname} processor {plan.n_out}x{plan.n_in}, D={d}, "
                                       f"H={layer.n_head}, batch {batch}",
            "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": pmc_traffic(f"posatt_rows_fwd_b{batch}", f"{plan.n_out}x{plan.n_in}_D{d}_H{layer.n_head}_b{batch}"),
            "us_per_launch": round(us, 3), "flops_per_launch": flops, "algorithmic_bytes": alg_bytes}


def roofline_attention_bwd_probe(model, batch):
    """The launch with the LARGEST share of the Darcy b=8 step since round 2 (profiles/r02_darcy8.summary.txt:
    posatt_bwd_pair_dw_kernel, 4 x 13.4 us = 20 %): the backward of a processor attention layer - d(scale) over the
    rows and d(values) over the keys in one launch - carrying the weight-gradient reductions of the block's MLP
    (pit_hip.h: rider).  Called through the C ABI exactly as ops._PosAtt.backward does, with a synthetic postponed
    pit_mlp_bwd_params job of the block MLP's shape.  Algorithmic FLOPs: 2 x the forward's 2*H*N*J*D*b (two
    contractions) + 2*rows*(n0*n1 + n1*n2) of the two reductions."""
    import ctypes
    from position_induced_transformer_amd import _lib, ops
    layer, mlp = model.conv[0], model.mlp[0]
    mesh = model.mesh_ltt
    if mesh is None:
        return None
    plan = layer._plan(mesh, mesh, True)
    d, H = model.hid_dim, layer.n_head
    u = torch.randn(batch, plan.n_in, d, device="cuda", requires_grad=True)
    lm = layer.lmda.detach().clone().requires_grad_(True)
    out = ops.posatt_apply(u, lm, plan, H, True)
    values, head, rowstat, scale = out.grad_fn.saved_tensors
    d_out = torch.randn_like(out)
    d_values = torch.empty_like(u)
    d_head = torch.zeros(H, device="cuda")
    work = torch.zeros(H * 1024, device="cuda", dtype=torch.float64)
    n0, n1, n2 = mlp.mlp1.in_features, mlp.mlp1.out_features, mlp.mlp2.out_features
    rows = batch * plan.n_out
    x2, hh = torch.randn(rows, n0, device="cuda"), torch.randn(rows, n1, device="cuda")
    scratch = torch.randn(rows * (n1 + n2), device="cuda")
    gw1, gb1 = torch.zeros(n1, n0, device="cuda"), torch.zeros(n1, device="cuda")
    gw2, gb2 = torch.zeros(n2, n1, device="cuda"), torch.zeros(n2, device="cuda")
    job = _lib.MlpParamsJob(x2.data_ptr(), n0, rows, n0, n1, n2, hh.data_ptr(), 1, d_out.data_ptr(), d_out.stride(1),
                            gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), 1, scratch.data_ptr(), 0)
    L = _lib.lib()
    if not L.pit_mlp_bwd_params_deferrable(rows, n0, n1, n2, 1, n2):
        return None                                   # large regime: the reductions keep their own launches

    def launch():
        rc = L.pit_posatt_bwd(
            plan.mesh_out.data_ptr(), plan.mesh_in.data_ptr(), plan.mesh_batch, plan.n_out, plan.n_in,
            plan.sdim, plan.metric_id, plan.period,
            values.data_ptr(), batch, d, values.stride(1), values.stride(0),
            head.data_ptr(), H, 0, scale.data_ptr(), rowstat.data_ptr(), 1 if plan.masked else 0,
            d_out.data_ptr(), d_out.stride(1), d_out.stride(0), d,
            d_values.data_ptr(), d_values.stride(1), d_values.stride(0), 1,
            d_head.data_ptr(), 1 | 2, work.data_ptr(),            # PIT_HEAD_ACCUMULATE | PIT_HEAD_DEFER: one launch
            _lib.ptr(plan.nbr_idx), _lib.ptr(plan.nbr_cnt), plan.nbr_cap, plan.lists_complete(),
            _lib.ptr(plan.rev_ptr), _lib.ptr(plan.rev_row),
            ctypes.cast(ctypes.pointer(job), ctypes.c_void_p), 0, 0, _lib.stream_ptr())
        _lib.check(rc, "pit_posatt_bwd")

    us = graph_time_us(launch)
    work.zero_()
    att = 2.0 * 2.0 * H * plan.n_out * plan.n_in * d * batch
    red = 2.0 * rows * (n0 * n1 + n1 * n2)
    flops = att + red
    achieved = flops / (us * 1e-6) / 1e12
    alg_bytes = 4.0 * (batch * plan.n_in * d * 2 + 2 * batch * plan.n_out * (1 + H) * d + rows * (n0 + 2 * n1 + n2))
    return {"bound": "mfma", "kernel": f"posatt_bwd_pair_dw_kernel: d(scale)+d(values) of a processor layer {plan.n_out}x{plan.n_in}, "
                                       f"D={d}, H={H}, batch {batch}, + dW/db of its MLP {n0}->{n1}->{n2} ({rows} rows)",
            "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(f"posatt_bwd_pair_dw_b{batch}", f"b{batch}"),
            "us_per_launch": round(us, 3), "flops_per_launch": flops, "algorithmic_bytes": alg_bytes}


def roofline_block_probe(model, batch, rows_out):
    """The launches with the LARGEST share of the Darcy b=8 step since round 3 (profiles/r03_darcy8.summary.txt): the fused
    processor-block kernels of csrc/pit_block.hip, called through the C ABI exactly as ops._Processor does.
      block_bwd_kernel: d(values) of a block (2*H*L*L*D*b) + its d(scale) (2*H*L*L*D*b) + the data path of the previous
        block's MLP backward (dZ1 = dZ2 W2, dX = dZ1 W1: 2*rows*(n1*n2 + n1*n0)) + the weight-gradient reductions of
        the block's own MLP (2*rows*(n0*n1 + n1*n2)) in ONE launch;
      block_fwd_kernel: the block's attention (2*H*L*L*D*b) + its MLP forward (2*rows*(n0*n1 + n1*n2)).
    Returns (backward record, forward record) or None when the shape does not take the fused path."""
    import ctypes
    from position_induced_transformer_amd import _lib, ops
    if model.mesh_ltt is None or not hasattr(model, "conv") or len(model.conv) < 2:
        return None
    layer, mlp = model.conv[1], model.mlp[1]
    L, H, D = model.mesh_ltt.shape[0], layer.n_head, model.hid_dim
    if not ops.block_fusion_supported(L, H, D, batch):
        return None
    Lb = _lib.lib()
    W, rows, n = (1 + H) * D, batch * L, 1
    plan = layer._plan(model.mesh_ltt, model.mesh_ltt, True)
    head = layer.lmda.detach().reshape(-1).contiguous()
    E = torch.empty(n, H, L, L, device="cuda"); Q = torch.empty_like(E)
    inv = torch.empty(n, H, L, device="cuda"); rs = torch.empty(n, H, L, 4, device="cuda"); sc = torch.empty(n, H, device="cuda")
    hp = (ctypes.c_void_p * n)(head.data_ptr())
    _lib.check(Lb.pit_block_weights(plan.mesh_in.data_ptr(), L, plan.sdim, plan.metric_id, plan.period, n, hp, 0, H, E.data_ptr(),
                                    Q.data_ptr(), inv.data_ptr(), rs.data_ptr(), sc.data_ptr(), _lib.stream_ptr()), "pit_block_weights")
    xc = torch.randn(batch, L, W, device="cuda"); y = torch.empty(batch, L, W, device="cuda")
    z1 = torch.randn(rows, D, device="cuda"); hh = torch.randn(rows, D, device="cuda"); z2 = torch.randn(rows, D, device="cuda")
    w1, b1, w2, b2 = (t.detach().contiguous() for t in (mlp.mlp1.weight, mlp.mlp1.bias, mlp.mlp2.weight, mlp.mlp2.bias))
    dxc = torch.randn(batch, L, W, device="cuda"); dxp = torch.empty(batch, L, W, device="cuda")
    scr = torch.empty(rows * 2 * D, device="cuda"); scr_own = torch.randn(rows * 2 * D, device="cuda")
    ws = torch.zeros(H * 1024, device="cuda", dtype=torch.float64)
    gw1, gb1, gw2, gb2 = (torch.zeros_like(t) for t in (w1, b1, w2, b2))
    job = _lib.MlpParamsJob(xc.data_ptr(), W, rows, W, D, D, hh.data_ptr(), 1, scr_own.data_ptr(), D, gw1.data_ptr(),
                            gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), 1, scr_own.data_ptr(), 0)
    jp = ctypes.cast(ctypes.pointer(job), ctypes.c_void_p)
    # rider2: the row slice of the DECODER MLP's weight-gradient job this launch carries in the step (ops._dw_slices: the
    # job is cut into n_blocks slices, one per block launch) - present when that job is postponed (small regime)
    jp2, slice_flops, slice_bytes = None, 0.0, 0.0
    de, n_blocks = getattr(model, "de", None), len(model.mlp)
    if de is not None and hasattr(de, "mlp1"):
        rows_de = batch * int(rows_out)               # the decoder MLP's rows: output points per sample x batch
        dn0, dn1, dn2 = de.mlp1.in_features, de.mlp1.out_features, de.mlp2.out_features
        if rows_de >= ops.BIG_RIDER_ROWS and Lb.pit_mlp_bwd_params_deferrable(rows_de, dn0, dn1, dn2, 0, dn2):
            per = -(-rows_de // n_blocks // 16) * 16
            xs_, hs_ = torch.randn(per, dn0, device="cuda"), torch.randn(per, dn1, device="cuda")
            dys, scs = torch.randn(per, dn2, device="cuda"), torch.randn(per * dn1, device="cuda")
            g1, gb1_, g2, gb2_ = (torch.zeros(dn1, dn0, device="cuda"), torch.zeros(dn1, device="cuda"),
                                  torch.zeros(dn2, dn1, device="cuda"), torch.zeros(dn2, device="cuda"))
            job2 = _lib.MlpParamsJob(xs_.data_ptr(), dn0, per, dn0, dn1, dn2, hs_.data_ptr(), 0, dys.data_ptr(), dn2, g1.data_ptr(),
                                     gb1_.data_ptr(), g2.data_ptr(), gb2_.data_ptr(), 1, scs.data_ptr(), 0)
            jp2 = ctypes.cast(ctypes.pointer(job2), ctypes.c_void_p)
            slice_flops = 2.0 * per * (dn0 * dn1 + dn1 * dn2)
            slice_bytes = 4.0 * per * (dn0 + 2 * dn1 + dn2)
            keep_alive = (xs_, hs_, dys, scs, g1, gb1_, g2, gb2_, job2)     # noqa: F841

    def fwd():
        _lib.check(Lb.pit_block_fwd(E[0].data_ptr(), inv[0].data_ptr(), L, H, D, batch, xc.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                    w2.data_ptr(), b2.data_ptr(), 1, z1.data_ptr(), hh.data_ptr(), z2.data_ptr(), y.data_ptr(), W, 0,
                                    _lib.stream_ptr()), "pit_block_fwd")

    def bwd():
        _lib.check(Lb.pit_block_bwd(E[0].data_ptr(), inv[0].data_ptr(), Q[0].data_ptr(), L, H, D, batch, dxc.data_ptr(), xc.data_ptr(),
                                    ws.data_ptr(), w1.data_ptr(), w2.data_ptr(), z1.data_ptr(), z2.data_ptr(), 1, W, dxp.data_ptr(), W,
                                    scr.data_ptr(), None, 0, jp, jp2, 0, _lib.stream_ptr()), "pit_block_bwd")

    att = 2.0 * H * L * L * D * batch
    mlp_f = 2.0 * rows * (W * D + D * D)
    recs = []
    for name, fn, flops, nbytes, what in (
            ("block_bwd_kernel", bwd, 2 * att + 2.0 * rows * (D * D + D * W) + mlp_f + slice_flops,
             4.0 * (2 * rows * W + rows * W + 4 * rows * D + rows * W + 2 * H * L * L + rows * (W + 3 * D)) + slice_bytes,
             f"d(values)+d(scale) of a processor block {L}x{L}, D={D}, H={H}, batch {batch}, + data path of the previous block's "
             f"MLP backward + dW/db of the block's own MLP {W}->{D}->{D} ({rows} rows)"
             + (f" + 1/{n_blocks} of the decoder MLP's dW/db" if jp2 is not None else "")),
            ("block_fwd_kernel", fwd, att + mlp_f, 4.0 * (2 * rows * W + 4 * rows * D + H * L * L + W * D + D * D),
             f"attention of a processor block {L}x{L}, D={D}, H={H}, batch {batch}, + its MLP forward {W}->{D}->{D} ({rows} rows)")):
        us = graph_time_us(fn)
        ws.zero_()
        achieved = flops / (us * 1e-6) / 1e12
        recs.append({"bound": "mfma", "kernel": f"{name}: {what}", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(f"{name}_b{batch}", f"L{L}_H{H}_D{D}_b{batch}"),
                     "us_per_launch": round(us, 3), "flops_per_launch": flops, "algorithmic_bytes": nbytes})
    return recs


def executed_gflop_per_sample(model, step):
    """FLOPs the kernels actually execute per sample (fwd+bwd = 3 x fwd), next to the dense
    reference-equivalent figure: masked layers run on candidate lists, so their A.V work is
    2*H*N*(listed keys)*D instead of 2*H*N*J*D (Darcy: the decoder is 121 of the 194 MFLOP of dense A.V)."""
    from position_induced_transformer_amd import pit as P
    b = step.func_in.shape[0]
    with torch.no_grad():
        step.model(step.mesh_in, step.func_in, step.mesh_out)        # make sure every layer has a cached plan
    total = 0.0
    for mod in model.modules():
        if isinstance(mod, P.posatt):
            for plan in mod._plans.values():
                h = mod.n_head
                d = mod.in_dim
                if plan.nbr_cnt is not None:
                    listed = float(torch.clamp(plan.nbr_cnt, max=plan.nbr_cap).sum()) / plan.mesh_batch
                    total += 2.0 * h * listed * d
                else:
                    total += 2.0 * h * plan.n_out * plan.n_in * d
                break
        elif isinstance(mod, P.kaiming_mlp):
            pass
    mlp = 0.0
    for name, mod in model.named_modules():
        if isinstance(mod, P.kaiming_mlp):
            rows = (step.mesh_out.reshape(-1, model.space_dim).shape[0] if name == "de" else model.mesh_ltt.shape[0])
            mlp += 2.0 * rows * (mod.mlp1.in_features * mod.mlp1.out_features + mod.mlp1.out_features * mod.mlp2.out_features)
    return 3.0 * (total + mlp) / 1e9


def _oracle_step(args, model, meta, affine, mesh_in, func_in, mesh_out, target):
    """oracle/pit_oracle.py forward + loss + backward on the host for the model's current parameters and the given (CPU) batch:
    (prediction before the affine map, loss, {name: gradient}, seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pit_oracle as orc
    b = func_in.shape[0]
    sd = {k: q.detach().cpu().clone().requires_grad_(True) for k, q in model.named_parameters()}
    s = model.space_dim
    metric = model.down._metric
    norm = hasattr(model, "norm")
    t0 = time.perf_counter()
    if model.mesh_ltt is not None:                       # fixed-mesh tasks (train_darcy.py:46-59 and alike)
        mi = mesh_in.reshape(-1, s)
        mo = mesh_out.reshape(-1, s)
        f = orc.with_coords(mi, func_in.reshape(b, -1, model.in_dim))
        ref = orc.pit_apply(sd, metric, False, model.n_blocks, model.en_local, model.de_local, mi, f,
                            model.mesh_ltt.cpu(), mo, norm_after_enc_proc=norm)
        if getattr(model, "residual", False):
            ref = ref + func_in.reshape(ref.shape)
    elif args.task == "elasticity":                      # train_elasticity.py:41-54
        ref = orc.pit_apply(sd, metric, True, model.n_blocks, model.en_local, model.de_local, mesh_in, func_in, mesh_out, mesh_out)
    elif args.task == "naca":                            # train_naca.py:52-65
        ltt, flat = model.ltt_mesh(mesh_out)
        ref = orc.pit_apply(sd, metric, True, model.n_blocks, model.en_local, model.de_local, mesh_in, func_in, ltt, flat)
    else:
        return None
    ref = ref.reshape(target.shape)
    pred = ref if affine is None else ref * affine[0].cpu() + affine[1].cpu()
    ref_loss = orc.rel_lp_loss(target, pred, meta["out_dim"], meta["p"])
    ref_loss.backward()
    return ref.detach(), float(ref_loss), {k: v.grad for k, v in sd.items()}, time.perf_counter() - t0


def _rel(a, r):
    return float((a.double() - r.double()).norm() / (r.double().norm() + 1e-300))


def _parity_record(args, meta, out_dev, loss_dev, dev_grads, ref_out, ref_loss, ref_grads, secs, rows):
    w = {k: _rel(dev_grads[k], ref_grads[k]) for k in ref_grads if not k.endswith("lmda")}
    l = {k: _rel(dev_grads[k], ref_grads[k]) for k in ref_grads if k.endswith("lmda")}
    kw, kl = max(w, key=w.get), max(l, key=l.get)
    # all layers' d(lmda) as one vector (what tests/test_gpu_bf16.py bounds: a single layer's scalar is a cancellation-heavy sum
    # and its own relative error is noisier than the vector's)
    lk = [k for k in ref_grads if k.endswith("lmda")]
    l_all = _rel(torch.cat([dev_grads[k].reshape(-1) for k in lk]), torch.cat([ref_grads[k].reshape(-1) for k in lk]))
    return {"rel_l2_out": _rel(out_dev, ref_out), "rel_loss": abs(loss_dev - ref_loss) / abs(ref_loss),
            "rel_l2_weight_grad_worst": w[kw], "worst_weight_grad": kw,
            "rel_l2_dlmda_all_layers": l_all, "rel_l2_dlmda_worst": l[kl], "worst_dlmda": kl,
            "head_scale_route": args.head_scale_route, "math": args.math,
            # weight gradients are fp32 sums over batch x points rows on BOTH sides (the oracle's torch-CPU reductions too): the
            # 2e-5 of the tests (batch 2, tests/test_gpu_round2.py) grows with the square root of the rows beyond Darcy b=8's 14 792
            "tolerance": {"out": 1e-5, "weight_grad": round(2e-5 * max(1.0, (rows / 16384.0) ** 0.5), 7),
                          "dlmda_all_layers": 2e-4} if args.math == "fp32" else     # (one vector, as the tests judge it: a single layer's
                                                                                     # d(lmda) can be 1e-9 of the others - its own ratio is noise)
                         {"out": 2e-2, "weight_grad": 5e-2, "dlmda_all_layers": 5e-2},
            "oracle_seconds": round(secs, 2)}


def parity_vs_oracle(args, step, model, affine, meta):
    """rel-L2 of what the TIMED step computed - the prediction, the loss and every gradient left in the flat buffer
    by the last replay of the timed graph, on the timed model, inputs and head-scale route - against the CPU oracle
    (oracle/pit_oracle.py: the reference's op sequence, pinned bit-equal to /root/reference/pit.py in the build
    container) run here on the same parameters and inputs.  BASELINE.json's metric asks for this number next to the
    throughput.  Tolerances of the parity tests: output 1e-5, weight gradients 2e-5, d(lmda) 2e-4."""
    torch.cuda.synchronize()
    dev_grads = {k: q.grad.detach().cpu().clone() for k, q in model.named_parameters()}
    out_dev, loss_dev = step.out.detach().cpu(), float(step.loss)
    got = _oracle_step(args, model, meta, affine, step.mesh_in.cpu(), step.func_in.cpu(), step.mesh_out.cpu(), step.target.cpu())
    if got is None:
        return None
    ref_out, ref_loss, ref_grads, secs = got
    rec = _parity_record(args, meta, out_dev, loss_dev, dev_grads, ref_out.reshape(out_dev.shape), ref_loss, ref_grads, secs,
                         out_dev.numel() / meta["out_dim"])
    rec["what"] = ("the timed hipGraph's own results (prediction, loss, flat gradient buffer after its last replay) vs "
                   "oracle/pit_oracle.py forward+loss+backward on this host, same parameters and inputs")
    return rec


def parity_ddp(args, step, model, affine, meta, world, rank):
    """SURVEY 8(e)'s parity row for the step that was TIMED on `world` ranks: rank 0 rebuilds the GLOBAL batch (every rank's
    shard comes from the generator seeded 100 + rank, build_step) and runs the oracle on it; the all-reduced flat gradient must be
    the global batch's gradient (RelLpNorm sums over the batch, utils.py:98: the reduction is SUM), the losses summed over the
    ranks its loss, rank 0's prediction its first shard's.  Collective: every rank calls it."""
    torch.cuda.synchronize()
    loss_all = step.loss.detach().reshape(1).clone()
    if dist.get_backend() == "gloo":
        loss_all = loss_all.cpu()
    dist.all_reduce(loss_all)
    if rank != 0:
        return None
    dev_grads = {k: q.grad.detach().cpu().clone() for k, q in model.named_parameters()}
    b = step.func_in.shape[0]
    funcs, targets = [], []
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        funcs.append(torch.randn(step.func_in.shape, generator=g))
        targets.append(torch.randn(step.target.shape, generator=g))
    if not torch.equal(funcs[0], step.func_in.cpu()):
        return {"error": "rank 0's shard is not what the seed rebuilds"}
    rep = lambda m: m.cpu() if model.mesh_ltt is not None else torch.cat([m.cpu()] * world, 0)     # per-sample meshes: same on every rank
    got = _oracle_step(args, model, meta, affine, rep(step.mesh_in), torch.cat(funcs, 0), rep(step.mesh_out), torch.cat(targets, 0))
    if got is None:
        return None
    ref_out, ref_loss, ref_grads, secs = got
    out_dev = step.out.detach().cpu()
    rec = _parity_record(args, meta, out_dev, float(loss_all), dev_grads, ref_out[:b].reshape(out_dev.shape), ref_loss, ref_grads, secs,
                         world * out_dev.numel() / meta["out_dim"])
    rec["world"] = world
    rec["what"] = (f"{world} rank(s): the all-reduced flat gradient buffer vs the oracle's gradient of the GLOBAL batch ({world * b} "
                   "samples, rebuilt on rank 0 from the ranks' seeds), the losses summed over the ranks vs its loss, rank 0's "
                   "prediction vs its first shard's (SURVEY 8(e): W-rank gradient == 1-rank gradient on the concatenated batch)")
    return rec


def cpu_baseline(batch, iters):
    """The oracle's Darcy forward+loss+backward on the host cores (PyTorch CPU eager fp32)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pit_oracle as orc
    torch.manual_seed(0)
    try:
        cores = len(os.sched_getaffinity(0))      # cores this process may actually use
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, cores)
    shapes = orc.param_shapes(2, 1, 1, 64, 2, 4)
    p = {k: v.requires_grad_(True) for k, v in orc.init_params(shapes, 0).items()}
    mesh, ltt = orc.grid_mesh_2d(43), orc.grid_mesh_2d(16)
    x, y = torch.randn(batch, 1849, 1), torch.randn(batch, 43, 43, 1)

    def it():
        for v in p.values():
            v.grad = None
        f = orc.with_coords(mesh, x)
        o = orc.pit_apply(p, "euclid", False, 4, 0.02, 0.02, mesh, f, ltt, mesh).reshape(batch, 43, 43, 1)
        orc.rel_lp_loss(y, o, 1, 2).backward()
    # small eager ops do not scale to every core of a big host: pick the fastest thread count
    best = (float("inf"), 1)
    for nt in sorted({min(cores, 64), min(cores, 32), min(cores, 16), min(cores, 8)}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        it()
        if time.perf_counter() - t0 > 2.0:        # oversubscribed: eager micro-ops collapse, skip
            log(f"cpu baseline trial: {nt} threads -> > 2 s/step, skipped")
            continue
        t0 = time.perf_counter()
        for _ in range(3):
            it()
        dt = (time.perf_counter() - t0) / 3
        log(f"cpu baseline trial: {nt} threads -> {dt * 1e3:.1f} ms/step")
        if dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    log(f"cpu baseline: {best[1]} threads of {cores} available")
    t_begin = time.perf_counter()
    for _ in range(3):
        it()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        it()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > 20.0 and len(ts) >= 5:     # bounded sample (~10-30 s of CPU work)
            break
    iters = len(ts)
    med = statistics.median(ts)
    best_threads = torch.get_num_threads()
    torch.set_num_threads(1)                        # SURVEY 8(d) also asks for the single-thread figure
    it()
    t1 = []
    t_begin = time.perf_counter()
    while len(t1) < 12 and (time.perf_counter() - t_begin < 6.0 or len(t1) < 3):
        t0 = time.perf_counter()
        it()
        t1.append(time.perf_counter() - t0)
    one = statistics.median(t1)
    torch.set_num_threads(best_threads)
    return {"value": round(batch / med, 2), "unit": "samples/s", "cores": best_threads,
            "one_thread": {"value": round(batch / one, 2), "ms_per_step": round(one * 1e3, 2), "iterations": len(t1)},
            "cores_available": cores, "kind": "port",
            "sample": f"{iters} iterations of Darcy2D 43x43 b={batch} fwd+loss+bwd (median; min {min(ts)*1e3:.1f} ms, "
                      f"max {max(ts)*1e3:.1f} ms), oracle/pit_oracle.py on PyTorch-CPU eager fp32",
            "ms_per_step": round(med * 1e3, 3),
            "note": "a reported baseline, not the target: kernel quality is what `roofline.frac` says"}


# ------------------------------------------------------------------------------------------
def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON record): libraries that print to stdout on their own
    # (RCCL's version banner at communicator creation) are routed to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
    if args.device_index is not None:
        local = args.device_index
    if args.backend == "gloo":
        args.eager_allreduce = True
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    distributed = world > 1 or "RANK" in os.environ           # under torch.distributed.run even with 1 rank
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(local)}")
    from position_induced_transformer_amd import ops
    ops.set_math_mode(args.math)
    # the timed fwd+bwd step never updates lmda: the exact (host-evaluated, cached) head scale is sync-free here
    ops.set_head_scale_route(args.head_scale_route)
    args.ar_buckets_n = 2 if args.ar_buckets == "2" else 1
    ar_choice = {"requested": args.ar_buckets}
    if distributed and args.ar_buckets == "auto" and not args.eager_allreduce and not args.no_graph and not args.rollout:
        # what the captured exchange costs on THIS node: the step without it, with one all-reduce, and - if that costs more
        # than 10 % of the step - with the two-bucket overlap; every rank takes the same decision (MAX over ranks)
        def quick(all_reduce, buckets):
            args.ar_buckets_n = buckets
            st, _, _ = build_step(args, device, rank, world, args.batch, all_reduce=all_reduce)
            with Watchdog(args.watchdog, "capturing a candidate step"):
                rn, _ = prepare(st, True)
                dtq, _ = timed_blocks(rn, 20, 5, world, min_total=0.05, min_blocks=3)
            del st, rn
            return dtq / 20 * 1e3
        t_none, t_one = quick(False, 1), quick(True, 1)
        ar_choice.update(ms_per_step_no_exchange=round(t_none, 4), ms_per_step_one_allreduce=round(t_one, 4))
        args.ar_buckets_n = 1
        if t_one > 1.10 * t_none:
            t_two = quick(True, 2)
            ar_choice["ms_per_step_two_buckets"] = round(t_two, 4)
            args.ar_buckets_n = 2 if t_two < t_one else 1
        log("all-reduce cost on this node: " + json.dumps(ar_choice) + f" -> {args.ar_buckets_n} bucket(s)")
    step, model, meta = build_step(args, device, rank, world, args.batch, all_reduce=distributed)
    if distributed and args.eager_allreduce and not args.no_graph:
        step.all_reduce = False
        step.capture()
        mode = "hipgraph+eager-allreduce"

        def run():
            step.replay()
            step.flat.all_reduce()
    elif distributed:
        with Watchdog(args.watchdog, "capturing the step and its first replays"):
            run, mode = prepare(step, not args.no_graph)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
    else:
        run, mode = prepare(step, not args.no_graph)
    log(f"step prepared ({mode})")
    dt, spread = timed_blocks(run, args.steps, args.warmup, world)
    log(f"timed region done: {dt / args.steps * 1e3:.4f} ms/step (median of {spread['blocks']} blocks of {args.steps} steps, "
        f"{spread['min_ms_per_step']}..{spread['max_ms_per_step']})")
    ms = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    loss_val = float(step.loss)

    # N > 1 (and a 1-rank torch.distributed.run): SURVEY 8(e)'s parity row for the timed step - collective, every rank
    ddp_parity = None
    if distributed and not args.no_parity and not args.rollout:
        try:
            ddp_parity = parity_ddp(args, step, model, darcy_affine(device) if args.task == "darcy" else None, meta, world, rank)
        except Exception as exc:
            ddp_parity = {"error": f"{type(exc).__name__}: {exc}"}
        if rank == 0:
            log("parity_ddp: " + json.dumps({k: v for k, v in (ddp_parity or {}).items() if k.startswith("rel_") or k == "error"}))
    # ... and a saturated-regime scaling point next to the latency-regime headline (collective too)
    ddp_sweep = None
    if distributed and (not args.no_extras or args.ddp_sweep) and not args.rollout and args.sweep_batch != args.batch:
        try:
            st6, _, _ = build_step(args, device, rank, world, args.sweep_batch, all_reduce=True)
            with Watchdog(args.watchdog, "capturing the saturated-batch step"):
                if args.eager_allreduce and not args.no_graph:
                    st6.all_reduce = False
                    st6.capture()

                    def run6():
                        st6.replay()
                        st6.flat.all_reduce()
                else:
                    run6, _ = prepare(st6, not args.no_graph)
                for _ in range(2):
                    run6()
                torch.cuda.synchronize()
            n6 = max(args.steps // 4, 10)
            dt6, _ = timed_blocks(run6, n6, 5, world, min_total=0.1)
            ddp_sweep = {str(args.sweep_batch): round(args.sweep_batch * world * n6 / dt6, 1),
                         "ms_per_step": round(dt6 / n6 * 1e3, 4),
                         "what": f"whole-job samples/s at per-GPU batch {args.sweep_batch} on {world} rank(s), gradient exchange included"}
            del st6, run6
            torch.cuda.empty_cache()
        except Exception as exc:
            ddp_sweep = {"error": f"{type(exc).__name__}: {exc}"}
        if rank == 0:
            log("saturated-batch scaling point: " + json.dumps(ddp_sweep))
    if rank == 0:
        rec = {
            "metric": "PiT fwd+bwd samples/sec on Darcy2D" if args.task == "darcy" else f"PiT fwd+bwd samples/sec on {args.task}",
            "value": round(value, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.math == "fp32" else "bf16 MFMA operands, f32 accumulate; decoder tail stored as bf16; the small-regime "
                                                         "fused MLP kernels contract in f32",
            "data": "synthetic",
            "config": {"workload": WORKLOADS.get(args.task, args.task) + (f", {args.rollout}-step autoregressive rollout + ONE backward "
                                                                           f"(train_vorticity.py:118-126){', activation recompute' if args.recompute else ''}"
                                                                           if args.rollout else ", fwd+loss+bwd") + f", per-GPU batch {args.batch}",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" if world > 1 else "single", "launch": mode,
                       "head_scale_route": args.head_scale_route,
                       "allreduce": ({"backend": dist.get_backend(), "captured": mode == "hipgraph",
                                      "buckets": getattr(step, "buckets", 1), "floats": int(step.flat.flat.numel()),
                                      "choice": ar_choice}
                                     if distributed else None)},
            "loss": round(loss_val, 6),
            "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 3),
            "timing": dict(spread, what="value = median over repeated blocks of exactly --steps steps, each bracketed by "
                                        "barrier + synchronize (MAX over ranks)"),
        }
        if args.task in MB_PER_SAMPLE_FWD:     # whole-step algorithmic activation traffic (fwd+bwd = 3x fwd)
            gbps = value * MB_PER_SAMPLE_FWD[args.task] * 3.0 / 1e3
            rec["step_hbm"] = {"algorithmic_GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS / world, 4),
                               "mb_per_sample_fwd": MB_PER_SAMPLE_FWD[args.task]}
        if args.task in GFLOP_PER_SAMPLE:      # whole-step algorithmic rate against the fp32 MFMA peak
            tf = value * GFLOP_PER_SAMPLE[args.task] / 1e3
            rec["step_tflops"] = {"dense_equivalent": round(tf, 2),
                                  "dense_equivalent_frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS / world, 4),
                                  "gflop_per_sample_dense_equivalent": GFLOP_PER_SAMPLE[args.task],
                                  "what": "dense_equivalent counts every masked attention layer as the dense N x J product the "
                                          "reference computes (SURVEY 8(d)); executed counts what the kernels run (masked layers "
                                          "on candidate lists)"}
            if model.mesh_ltt is not None:
                try:
                    ex = executed_gflop_per_sample(model, step)
                    rec["step_tflops"]["gflop_per_sample_executed"] = round(ex, 4)
                    rec["step_tflops"]["executed"] = round(value * ex / 1e3, 2)
                    rec["step_tflops"]["executed_frac_of_fp32_mfma_peak"] = round(value * ex / 1e3 / FP32_MFMA_PEAK_TFLOPS / world, 4)
                except Exception as exc:          # informational only
                    log(f"executed-FLOP count skipped: {type(exc).__name__}: {exc}")
        if ddp_parity is not None:
            rec["parity_ddp"] = ddp_parity
        if ddp_sweep is not None:
            rec["batch_sweep_ddp_samples_per_s"] = ddp_sweep
        if not args.no_parity and not args.rollout and world == 1:
            try:
                aff = darcy_affine(device) if args.task == "darcy" else None
                rec["parity"] = parity_vs_oracle(args, step, model, aff, meta)
                if rec["parity"] is not None:
                    log("parity vs oracle: " + json.dumps({k: v for k, v in rec["parity"].items() if k.startswith("rel_")}))
            except Exception as exc:
                rec["parity"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"parity block failed: {rec['parity']['error']}")
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras and not args.rollout and args.head_scale_route == "host":
        # ADVICE r3 / VERDICT r3 weak 1: the headline runs the EXACT head-scale route (host-evaluated c, legal while lmda is
        # frozen); a training step updates lmda and must take route 'device' (c evaluated in the kernels).  The same
        # fwd+loss+bwd step on that route - same seed, same inputs: its time and its parity against the oracle BEFORE any
        # update - so the record shows what training executes, not only what the metric's fwd+bwd executes.
        try:
            with ops.head_scale_route("device"):
                st5, model5, meta5 = build_step(args, device, rank, world, args.batch)
                run5, _ = prepare(st5, not args.no_graph)
                dt5, _ = timed_blocks(run5, max(args.steps // 2, 10), max(args.warmup // 2, 3), world, min_total=0.1)
                n5 = max(args.steps // 2, 10)
                dev_rec = {"ms_per_step": round(dt5 / n5 * 1e3, 4), "samples_per_s": round(args.batch * n5 / dt5, 1)}
                if not args.no_parity:
                    a5 = argparse.Namespace(**vars(args))
                    a5.head_scale_route = "device"
                    dev_rec["parity"] = parity_vs_oracle(a5, st5, model5, darcy_affine(device) if args.task == "darcy" else None, meta5)
            dev_rec["what"] = ("the timed fwd+loss+bwd step on head-scale route 'device' (what a step that updates lmda executes: "
                               "tan/sin evaluated in fp64 inside the kernels = the correctly rounded c, which differs from this "
                               "host's torch-CPU c for ~19 % of lmda values - DESIGN.md section 2; on 2-d grids a differing c can "
                               "move a tie shell across the quantile threshold, hence parity per seed, not guaranteed)")
            extras["device_route"] = dev_rec
            log("device-route step: " + json.dumps({k: v for k, v in dev_rec.items() if k != "what" and k != "parity"}))
            del st5, run5
        except Exception as exc:
            extras["device_route"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not args.no_extras:
        # (a) the same step with Adam (capturable) inside the graph
        with ops.head_scale_route("device"):        # lmda changes inside the graph: the in-kernel head scale
            st2, _, _ = build_step(args, device, rank, world, args.batch, with_optimizer=True)
            run2, _ = prepare(st2, not args.no_graph)
            dt2, _ = timed_blocks(run2, max(args.steps // 2, 10), max(args.warmup // 2, 3), world, min_total=0.1)
        log("train_step (with Adam) done")
        extras["train_step"] = {"samples_per_s": round(args.batch * max(args.steps // 2, 10) / dt2, 1),
                                "head_scale_route": "device",
                                "parity": (extras.get("device_route") or {}).get("parity"),
                                "what": "fwd+loss+bwd+fused Adam(1e-3)+cosine LR (pit_adam_step) in one hipGraph; lmda is "
                                        "updated inside the graph, so c is evaluated in the kernels (route 'device'); parity = "
                                        "the same model's fwd+loss+bwd on that route against the oracle before the first update "
                                        "(device_route.parity)"}
        # (b) saturating batches
        sweep = {}
        for b in (64, 256):
            st3, _, _ = build_step(args, device, rank, world, b)
            run3, _ = prepare(st3, not args.no_graph)
            n3 = max(args.steps // 4, 10)
            dt3, _ = timed_blocks(run3, n3, 5, world, min_total=0.1)
            sweep[str(b)] = round(b * n3 / dt3, 1)
            if args.task in GFLOP_PER_SAMPLE:
                sweep[str(b) + "_step_tflops"] = round(sweep[str(b)] * GFLOP_PER_SAMPLE[args.task] / 1e3, 2)
            log(f"batch {b}: {sweep[str(b)]} samples/s")
            del st3, run3
            torch.cuda.empty_cache()
        extras["batch_sweep_samples_per_s"] = sweep
        if args.math == "fp32":            # the opt-in bf16 math mode on the same steps (informational)
            bf = {}
            with ops.math_mode("bf16"):
                for b in (args.batch, 256):
                    st4, _, _ = build_step(args, device, rank, world, b)
                    run4, _ = prepare(st4, not args.no_graph)
                    n4 = max(args.steps // 4, 10)
                    bf[str(b)] = round(b * n4 / timed_blocks(run4, n4, 5, world, min_total=0.1)[0], 1)
                    log(f"bf16 math mode, batch {b}: {bf[str(b)]} samples/s")
                    del st4, run4
                    torch.cuda.empty_cache()
            extras["bf16_math_mode_samples_per_s"] = bf
        extras["roofline_saturated"] = roofline_probe(model, 256)
        extras["roofline_mlp_saturated"] = roofline_mlp_probe(model, 256)
        extras["roofline_dw_saturated"] = roofline_dw_probe(model, 256)
    if rank == 0:
        blk = roofline_block_probe(model, args.batch, step.mesh_out.reshape(-1, model.space_dim).shape[0]) \
            if (args.math == "fp32" and model.mesh_ltt is not None) else None
        bwd = roofline_attention_bwd_probe(model, args.batch)
        mlp = roofline_mlp_probe(model, args.batch)
        if blk is not None:
            rec["roofline"] = blk[0]                                      # the launch with the largest time share (round 3)
            rec["roofline_block_fwd"] = blk[1]
            rec["roofline_mlp"] = mlp
        elif bwd is not None:
            rec["roofline"] = bwd                                         # (round 2's dominant launch; the path of larger batches)
            rec["roofline_mlp"] = mlp                                     # (round 1-2's dominant family: the MLP forward)
        else:
            rec["roofline"] = mlp
        rec["roofline_attention"] = roofline_probe(model, args.batch)    # the fused position-attention forward
        log("roofline probes done")
        rec.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args.batch, args.cpu_iters)
            rec["speedup_vs_cpu_baseline"] = round(rec["value"] / rec["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
