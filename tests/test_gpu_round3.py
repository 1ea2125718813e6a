"""GPU: round 3 - the exact head-scale route made capturable for a frozen lmda (cached host evaluation),
the RCCL path exercised through a 1-rank torch.distributed.run child, subclass overrides of dist2att /
convolution, concurrency of the backward bookkeeping."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_io as gio
import model_cases as mc
import pit_oracle as orc
from test_gpu_models import build_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params_whose_device_c_differs(cs, lo=5000, hi=5400):
    from position_induced_transformer_amd import ops
    for seed in range(lo, hi):
        params = gio.synth_params(cs["shapes"], seed)
        differs = [k for k, v in params.items() if k.endswith("lmda") and not np.array_equal(
            orc.head_scale(torch.from_numpy(v)).numpy(), ops.head_scale(torch.from_numpy(v).cuda()).cpu().numpy())]
        if differs:
            return params, differs
    pytest.skip("no seed in range with a differing c on this host")


def _oracle_step(cs, params):
    cfg = cs["cfg"]
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in params.items()}
    mi = cs["mesh_in"].reshape(-1, 2)
    ref = orc.pit_apply(p, "euclid", False, cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"], mi,
                        orc.with_coords(mi, cs["func_in"].reshape(2, -1, 1)), cs["mesh_ltt"].reshape(-1, 2), mi)
    ref_loss = orc.rel_lp_loss(cs["target"], ref.reshape(2, 43, 43, 1), 1, 2)
    ref_loss.backward()
    return p, ref.detach().numpy().reshape(-1), float(ref_loss)


# --------------------------------------------------------------------------- exact route, cached
def test_host_route_graph_replay_matches_oracle_on_a_seed_whose_device_c_differs():
    """VERDICT r2 next-1 'done' criterion.  A Darcy parameter set for which the in-kernel c differs from this
    host's ATen c in at least one layer: the forward+loss+backward step captured under route 'host' (c evaluated
    by the reference's torch-CPU ops once, cached while lmda is frozen) replays to the oracle's prediction
    <= 1e-5, loss, weight gradients <= 2e-5 and d(lmda) <= 2e-4; replays take no device->host copy."""
    from position_induced_transformer_amd import ops
    from position_induced_transformer_amd.engine import TrainStep
    cs = mc.build_case("F9_model_darcy")
    params, differs = _params_whose_device_c_differs(cs)
    model = build_model(cs, params)
    step = TrainStep(model, (cs["mesh_in"].cuda(), cs["func_in"].cuda(), cs["mesh_out"].cuda(), cs["target"].cuda()), 1, 2)
    with ops.head_scale_route("host"):
        step.capture()
        evals = ops.HOST_SCALE_EVALUATIONS[0]
        for _ in range(3):
            step.replay()
        torch.cuda.synchronize()
        assert ops.HOST_SCALE_EVALUATIONS[0] == evals
    p, ref, ref_loss = _oracle_step(cs, params)
    assert gio.rel_l2(ref, step.out.cpu().numpy().reshape(-1)) <= 1e-5, differs
    assert abs(float(step.loss) - ref_loss) <= 1e-5 * abs(ref_loss)
    for k, q in model.named_parameters():
        tol = 2e-4 if k.endswith("lmda") else 2e-5
        assert gio.rel_l2(p[k].grad.numpy().reshape(-1), q.grad.cpu().numpy().reshape(-1)) <= tol, (k, differs)


def test_host_route_is_sync_free_while_lmda_is_frozen_and_re_evaluates_when_it_changes():
    from position_induced_transformer_amd import ops, tasks, utils
    model, sample, meta = tasks.make_task("darcy", seed=2)
    mesh_in, func_in, mesh_out, target = sample(2)
    loss_fn = utils.RelLpNorm(1, 2)
    n_layers = sum(1 for k, _ in model.named_parameters() if k.endswith("lmda"))
    with ops.head_scale_route("host"):
        e0 = ops.HOST_SCALE_EVALUATIONS[0]
        out0 = model(mesh_in, func_in, mesh_out)
        loss_fn(target, out0).backward()                 # (the first backward checks the cached plans' lists once: a sync)
        assert ops.HOST_SCALE_EVALUATIONS[0] == e0 + n_layers
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")          # any synchronising call raises from here on
        try:
            out1 = model(mesh_in, func_in, mesh_out)
            loss_fn(target, out1).backward()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert ops.HOST_SCALE_EVALUATIONS[0] == e0 + n_layers
        assert torch.equal(out0, out1)
        with torch.no_grad():                            # a torch in-place update bumps the version counter
            model.up.lmda.add_(0.25)
        out2 = model(mesh_in, func_in, mesh_out)
        assert ops.HOST_SCALE_EVALUATIONS[0] == e0 + n_layers + 1
        assert not torch.equal(out1, out2)
        ops.parameters_changed()                         # what raw-pointer writers (ddp.FlatAdam) announce
        model(mesh_in, func_in, mesh_out)
        assert ops.HOST_SCALE_EVALUATIONS[0] == e0 + 2 * n_layers + 1


def test_host_route_capture_is_refused_for_an_unevaluated_lmda_and_for_an_optimizer_in_the_graph():
    from position_induced_transformer_amd import ops, tasks
    from position_induced_transformer_amd.ddp import FlatAdam, FlatGradients
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, _ = tasks.make_task("darcy", seed=1)
    mesh_in, func_in, mesh_out, target = sample(2)
    model(mesh_in, func_in, mesh_out)                       # warm the plan caches (device route: no c cached)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.head_scale_route("host"):
        with pytest.raises(RuntimeError, match="cannot be captured"):
            with torch.cuda.graph(g):
                model(mesh_in, func_in, mesh_out)
    torch.cuda.synchronize()
    # (a) the fused optimizer (raw-pointer update) inside the graph
    flat = FlatGradients(model.parameters(), flatten_params=True)
    step = TrainStep(model, (mesh_in, func_in, mesh_out, target), 1, 2, optimizer=FlatAdam(flat, lr=1e-3), flat=flat)
    with ops.head_scale_route("host"):
        with pytest.raises(RuntimeError, match="FROZEN lmda"):
            step.capture()
    torch.cuda.synchronize()
    assert step.graph is None
    # (b) a torch optimizer (capturable) inside the graph: caught by lmda's version counter after the capture
    model2, _, _ = tasks.make_task("darcy", seed=1)
    opt = torch.optim.Adam(model2.parameters(), lr=1e-3, capturable=True)
    step2 = TrainStep(model2, (mesh_in, func_in, mesh_out, target), 1, 2, optimizer=opt)
    with ops.head_scale_route("host"):
        with pytest.raises(RuntimeError, match="FROZEN lmda"):
            step2.capture()
    torch.cuda.synchronize()
    assert step2.graph is None
    # the same steps capture fine on the device route
    step.capture()
    step.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(step.loss)


# --------------------------------------------------------------------------- RCCL path, one rank
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("extra", [[], ["--ar-buckets", "2"], ["--task", "elasticity", "--batch", "4", "--ar-buckets", "1"],
                                   ["--task", "naca", "--batch", "6", "--math", "bf16", "--ar-buckets", "1"]],
                         ids=["one-allreduce", "two-buckets", "elasticity-per-sample-plans", "naca-bf16-per-sample-plans"])
def test_bench_under_torch_distributed_run_one_rank_captures_the_rccl_allreduce(extra):
    """VERDICT r2 next-4 / r4 next-6c.  `python -m torch.distributed.run --nproc-per-node 1 bench.py ...` exactly as the driver
    launches N > 1 (a FRESH child: the launcher runs before anything touches the GPU): process group on backend
    'nccl' (= RCCL), the flat-gradient all-reduce captured INSIDE the step's hipGraph and replayed - rc 0, one JSON
    line, launch mode 'hipgraph' (not the eager-all-reduce fallback), parity block within tolerance."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--no-extras", "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    said = "\n--- child stdout ---\n" + res.stdout[-3000:] + "\n--- child stderr ---\n" + res.stderr[-6000:]
    assert res.returncode == 0, said
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, said
    rec = json.loads(lines[0])
    assert rec["config"]["launch"] == "hipgraph", said
    assert rec["config"]["allreduce"]["backend"] == "nccl" and rec["config"]["allreduce"]["captured"] is True, rec["config"]
    assert rec["config"]["allreduce"]["buckets"] == (2 if extra == ["--ar-buckets", "2"] else 1)
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    # BASELINE configs 4 / 5 (VERDICT r4 missing-6): per-sample selection plans rebuilt INSIDE the graph together with a
    # captured RCCL all-reduce of the 5.1 MB / 0.93 MB flat buffer; bf16 mode is held to the bf16 tolerances of tests/test_gpu_bf16.py
    # (tolerances: the parity block's own - 1e-5 / 2e-5 at Darcy b=8; the weight-gradient bound grows with the square root of the
    # rows summed beyond 16 384, bench.parity_vs_oracle)
    tol = rec["parity"]["tolerance"]
    assert tol["out"] == (2e-2 if "bf16" in extra else 1e-5) and tol["weight_grad"] <= (5e-2 if "bf16" in extra else 1e-4)
    assert rec["parity"]["rel_l2_out"] <= tol["out"], rec["parity"]
    assert rec["parity"]["rel_l2_weight_grad_worst"] <= tol["weight_grad"], rec["parity"]


# --------------------------------------------------------------------------- subclass overrides (pit.py:42-43)
def test_overridden_convolution_and_dist2att_are_what_forward_runs():
    from position_induced_transformer_amd import pit as P
    torch.manual_seed(5)
    mesh = torch.rand(64, 2, device="cuda")
    x = torch.randn(3, 64, 8, device="cuda", requires_grad=True)

    class doubled(P.posatt_fixed):
        def convolution(self, A, U):
            return 2.0 * super().convolution(A, U)

    class all_keys(P.posatt_cross_fixed):
        def dist2att(self, mesh_out, mesh_in, scale, locality):
            return super().dist2att(mesh_out, mesh_in, scale, 1.0)       # ignores the layer's locality

    base, mine = P.posatt_fixed(2, 8, 1.0).cuda(), doubled(2, 8, 1.0).cuda()
    mine.load_state_dict(base.state_dict())
    ref, got = base(mesh, x), mine(mesh, x)
    assert torch.equal(got[..., :8], ref[..., :8])
    assert gio.rel_l2((2.0 * ref[..., 8:]).detach().cpu().numpy(), got[..., 8:].detach().cpu().numpy()) <= 1e-6
    got.square().sum().backward()                                         # gradients flow through the composed path
    assert mine.lmda.grad is not None and torch.isfinite(mine.lmda.grad).all() and x.grad is not None
    masked, unmasked, override = P.posatt_cross_fixed(2, 8, 0.1).cuda(), P.posatt_cross_fixed(2, 8, 1.0).cuda(), all_keys(2, 8, 0.1).cuda()
    unmasked.load_state_dict(masked.state_dict())
    override.load_state_dict(masked.state_dict())
    out_mesh = torch.rand(16, 2, device="cuda")
    a, b, c = masked(out_mesh, mesh, x), unmasked(out_mesh, mesh, x), override(out_mesh, mesh, x)
    assert gio.rel_l2(b.detach().cpu().numpy(), c.detach().cpu().numpy()) <= 1e-6
    assert gio.rel_l2(b.detach().cpu().numpy(), a.detach().cpu().numpy()) > 1e-3


# --------------------------------------------------------------------------- backward bookkeeping under threads
def test_two_threads_running_backward_concurrently_give_the_single_thread_gradients():
    """the deferred d(lmda) finishes and the postponed weight-gradient reductions are keyed on
    autograd's graph-task id in process-global tables.  Two Python threads, each with its own model, flat gradient
    buffer and stream, run forward+backward at the same time: every gradient must equal the one the same model gives
    alone (the bookkeeping either isolates the passes or falls back to un-merged launches - never mixes them)."""
    import threading
    from position_induced_transformer_amd import tasks, utils
    from position_induced_transformer_amd.ddp import FlatGradients
    loss_fn = utils.RelLpNorm(1, 2)
    jobs = []
    for seed in (11, 12):
        model, sample, _ = tasks.make_task("darcy", seed=seed)
        mesh_in, func_in, mesh_out, target = sample(4)
        flat = FlatGradients(model.parameters())
        loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
        torch.cuda.synchronize()
        jobs.append(dict(model=model, batch=(mesh_in, func_in, mesh_out, target), flat=flat, want=flat.flat.clone(),
                         stream=torch.cuda.Stream(), errors=[]))
    barrier = threading.Barrier(2)

    def work(job):
        try:
            mesh_in, func_in, mesh_out, target = job["batch"]
            with torch.cuda.stream(job["stream"]):
                for _ in range(20):
                    job["flat"].zero_()
                    barrier.wait()
                    loss_fn(target, job["model"](mesh_in, func_in, mesh_out)).backward()
                    job["stream"].synchronize()
                    err = gio.rel_l2(job["want"].cpu().numpy(), job["flat"].flat.cpu().numpy())
                    if not err <= 2e-5:
                        job["errors"].append(err)
        except Exception as exc:                                        # a clean refusal is acceptable, silence is not
            job["errors"].append(repr(exc))
            barrier.abort()

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for j in jobs:
        assert not j["errors"], j["errors"][:3]


# --------------------------------------------------------------------------- fused processor blocks (pit_block.hip)
def _model_and_batch(task, seed, batch):
    from position_induced_transformer_amd import tasks
    model, sample, meta = tasks.make_task(task, seed=seed)
    return model, sample(batch), meta


@pytest.mark.parametrize("metric,sdim", [("euclid", 2), ("periodic1d", 1), ("periodic2d", 2)])
def test_block_weights_are_the_oracle_attention_matrix(metric, sdim):
    """pit_block_weights: E * inv == the reference's softmax weights (pit.py:133-139 / 190-200 / 248-258 with locality
    1.0), E is symmetric bit for bit, Q = P (m - mbar), rowstat as pit_posatt_fwd would save it."""
    import ctypes
    from position_induced_transformer_amd import _lib, ops
    L, H, n = 256, 2, 3
    if metric == "euclid":
        mesh = orc.grid_mesh_2d(16)
    elif metric == "periodic1d":
        mesh = orc.line_mesh_1d(256)
    else:
        mesh = orc.grid_mesh_2d(16, False)
    lm = [torch.from_numpy(gio.synth((H, 1, 1), 900 + i, 0.0, 1.0)) for i in range(n)]
    plan = ops.MeshPlan(metric, mesh.cuda(), mesh.cuda(), 1.0, True)
    heads = [t.reshape(-1).cuda() for t in lm]
    E = torch.empty(n, H, L, L, device="cuda"); Q = torch.empty_like(E)
    inv = torch.empty(n, H, L, device="cuda"); rs = torch.empty(n, H, L, 4, device="cuda"); sc = torch.empty(n, H, device="cuda")
    hp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in heads])
    rc = _lib.lib().pit_block_weights(plan.mesh_in.data_ptr(), L, sdim, plan.metric_id, plan.period, n, hp, 0, H, E.data_ptr(),
                                      Q.data_ptr(), inv.data_ptr(), rs.data_ptr(), sc.data_ptr(), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(E, E.transpose(-1, -2))
    m = orc.sqdist(metric, mesh, mesh)
    for i in range(n):
        c = sc[i].cpu().reshape(H, 1, 1)                                    # the c the kernel used (device route)
        att = orc.attention_weights(m, c, 1.0, False)                       # (H, L, L)
        got = (E[i] * inv[i].unsqueeze(-1)).cpu()
        assert gio.rel_l2(att.numpy(), got.numpy()) <= 1e-6
        mbar = (att * m).sum(-1, keepdim=True)
        assert gio.rel_l2((att * (m - mbar)).numpy(), Q[i].cpu().numpy()) <= 1e-5
        assert gio.rel_l2(mbar.reshape(H, L).numpy(), rs[i, :, :, 3].cpu().numpy()) <= 1e-6
        assert torch.equal(rs[i, :, :, 2], inv[i]) and float(rs[i, :, :, 1].abs().max()) == 0.0


@pytest.mark.parametrize("task,batch", [("darcy", 8), ("darcy", 3), ("burgers", 4)])
@pytest.mark.parametrize("inplace", [False, True])
def test_fused_processor_equals_the_block_by_block_path(task, batch, inplace):
    """ops.processor_apply (weights in one launch, one launch per block forward / backward) against the same model
    run block by block through pit_posatt_* and pit_mlp_*: prediction <= 1e-6, every gradient <= 1e-5 (lmda 1e-4),
    with the gradients returned to autograd and accumulated in place into a flat buffer."""
    from position_induced_transformer_amd import ops, utils
    from position_induced_transformer_amd.ddp import FlatGradients
    model, (mesh_in, func_in, mesh_out, target), meta = _model_and_batch(task, 21, batch)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])
    if inplace:
        flat = FlatGradients(model.parameters())
    res = {}
    calls = {"n": 0}
    orig = ops.processor_apply

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    ops.processor_apply = counting
    try:
        for fused in (True, False):
            ops.BLOCK_FUSION = fused
            if inplace:
                flat.zero_()
            else:
                model.zero_grad(set_to_none=True)
            out = model(mesh_in, func_in, mesh_out)
            loss_fn(target, out).backward()
            torch.cuda.synchronize()
            res[fused] = (out.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()})
    finally:
        ops.BLOCK_FUSION = True
        ops.processor_apply = orig
    assert calls["n"] == 1, "the fused path did not run"
    assert gio.rel_l2(res[False][0], res[True][0]) <= 1e-6
    # d(lmda) is a cancellation-heavy global sum: a layer whose gradient happens to be ~1e-10 carries the
    # rounding noise of the others' scale - its error is measured against the largest d(lmda) of the model
    lm_scale = max(float(np.linalg.norm(v)) for k, v in res[False][1].items() if k.endswith("lmda"))
    for k in res[False][1]:
        if k.endswith("lmda"):
            assert float(np.linalg.norm(res[False][1][k] - res[True][1][k])) <= 1e-4 * lm_scale, k
        else:
            assert gio.rel_l2(res[False][1][k], res[True][1][k]) <= 1e-5, k


def test_fused_processor_is_skipped_when_a_block_is_hooked_or_overridden():
    from position_induced_transformer_amd import ops, pit as P, tasks
    model, (mesh_in, func_in, mesh_out, _), _ = _model_and_batch("darcy", 22, 4)
    calls = {"n": 0}
    orig = ops.processor_apply

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    ops.processor_apply = counting
    try:
        seen = []
        ref = model(mesh_in, func_in, mesh_out)
        assert calls["n"] == 1
        h = model.mlp[1].register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
        out = model(mesh_in, func_in, mesh_out)
        h.remove()
        assert calls["n"] == 1 and seen == [(4, 256, 64)]                  # the user's hook fired: blocks ran one by one
        assert gio.rel_l2(ref.detach().cpu().numpy(), out.detach().cpu().numpy()) <= 1e-6

        class doubled(P.posatt_fixed):
            def convolution(self, A, U):
                return super().convolution(A, U)
        mine = doubled(2, 64, 1.0).cuda()
        mine.load_state_dict(model.conv[2].state_dict())
        model.conv[2] = mine
        out = model(mesh_in, func_in, mesh_out)
        assert calls["n"] == 1
        assert gio.rel_l2(ref.detach().cpu().numpy(), out.detach().cpu().numpy()) <= 1e-5
        with ops.math_mode("bf16"):
            model2, _, _ = tasks.make_task("darcy", seed=22)
            model2(mesh_in, func_in, mesh_out)
        assert calls["n"] == 2                  # (round 4: bf16 mode runs the fused blocks too - they contract in fp32 in every mode)
    finally:
        ops.processor_apply = orig


# --------------------------------------------------------------------------- two-bucket exchange: nothing lands after its reduce
@pytest.mark.parametrize("task,batch", [("darcy", 8), ("vorticity", 2), ("elasticity", 4)])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
def test_two_bucket_step_reduces_only_finished_gradients(task, batch, graph):
    """A one-process stand-in for the sum over two identical ranks: the collective DOUBLES the bucket it is given, in
    stream order, exactly where the step issues its all-reduce.  A weight-gradient contribution enqueued AFTER the early
    bucket's reduce (a rider in a later launch) would be added undoubled - the two-bucket step must give the same
    2 x gradient as the one-bucket step, eagerly and replayed."""
    from position_induced_transformer_amd.engine import TrainStep
    model, b4, meta = _model_and_batch(task, 31, batch)
    got = {}
    for buckets in (1, 2):
        step = TrainStep(model, b4, meta["out_dim"], meta["p"], all_reduce=True, all_reduce_buckets=buckets)
        assert step.buckets == buckets
        flat = step.flat

        def doubling(average=False, group=None, part="all", flat=flat):
            if part != "tail":
                flat.attach()
            buf = flat.flat if part == "all" else (flat.flat[flat.tail_start:] if part == "tail" else flat.flat[:flat.tail_start])
            buf.mul_(2.0)
        flat.all_reduce = doubling
        if graph:
            step.capture()
            for _ in range(3):
                step.replay()
        else:
            for _ in range(3):
                step.run_eager()
        torch.cuda.synchronize()
        got[buckets] = {k: q.grad.detach().clone() for k, q in model.named_parameters()}
        if step._early_hook is not None:
            step._early_hook.remove()
        for q in model.parameters():
            q.grad = None
    plain = TrainStep(model, b4, meta["out_dim"], meta["p"])
    plain.run_eager()
    torch.cuda.synchronize()
    for k, q in model.named_parameters():
        tol = 2e-4 if k.endswith("lmda") else 2e-5
        assert float(q.grad.abs().max()) > 0, k
        assert gio.rel_l2((2.0 * q.grad).cpu().numpy(), got[1][k].cpu().numpy()) <= tol, ("one bucket", k)
        assert gio.rel_l2((2.0 * q.grad).cpu().numpy(), got[2][k].cpu().numpy()) <= tol, ("two buckets", k)


def test_data_parallel_capture_tolerates_another_thread_polling_events():
    """The intermittent abort of the captured RCCL step (1 run in 8): ProcessGroupNCCL's watchdog thread calls
    hipEventQuery on the warm-up collectives' events while the step is being captured, which HIP refuses under the
    default GLOBAL capture error mode (tools/micro/capture_query_probe.py shows the three modes side by side).  A
    data-parallel TrainStep captures in thread-local mode: a second thread polling an event for the whole capture must
    not fail and must not invalidate the capture."""
    import threading
    from position_induced_transformer_amd.engine import TrainStep
    model, b4, meta = _model_and_batch("darcy", 41, 4)
    ev, other = torch.cuda.Event(), torch.cuda.Stream()
    with torch.cuda.stream(other):
        torch.zeros(8, device="cuda").add_(1)
        ev.record(other)
    torch.cuda.synchronize()

    def capture_while_polling(all_reduce):
        stop, errors, polls = threading.Event(), [], [0]

        def poll():
            while not stop.is_set():
                try:
                    ev.query()
                    polls[0] += 1
                except Exception as exc:
                    errors.append(repr(exc))
                    return
        step = TrainStep(model, b4, meta["out_dim"], meta["p"], all_reduce=all_reduce)
        step._step()                                    # (allocations and caches before the poller starts)
        torch.cuda.synchronize()
        orig = step._step

        def slow_step():                                # hold the capture window open long enough to be polled
            orig()
            n = polls[0]
            import time
            t0 = time.time()
            while polls[0] < n + 50 and not errors and time.time() - t0 < 5.0:
                time.sleep(0.001)
        t = threading.Thread(target=poll)
        t.start()
        failed = None
        try:
            step._step = slow_step
            step.capture(warmup=1)
        except Exception as exc:                        # (global mode: the refused query may also invalidate the capture)
            failed = exc
        finally:
            stop.set()
            t.join()
        torch.cuda.synchronize()
        if all_reduce:
            assert failed is None, failed
            step.replay()
            torch.cuda.synchronize()
            assert torch.isfinite(step.loss)
        return errors

    assert capture_while_polling(True) == []


# --------------------------------------------------------------------------- large-regime weight gradients (gemm_rr_kernel)
@pytest.mark.parametrize("rows,n0,n1,n2", [(16384, 192, 64, 64), (5120, 768, 256, 256), (14560, 256, 128, 128),
                                           (9999, 192, 64, 64), (7000, 320, 96, 160), (65536, 192, 64, 64)])
@pytest.mark.parametrize("math", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("out_gelu", [1, 0])
def test_weight_gradient_pair_launch_matches_fp64(rows, n0, n1, n2, math, out_gelu):
    """pit_mlp_bwd_params in the large regime (both reductions of an MLP in ONE gemm_rr_kernel launch, K slabs dealt so that
    every CU holds the same number of workgroups): dW1 = dZ1^T X, db1 = sum dZ1, dW2 = dZ2^T H, db2 = sum dZ2 against fp64
    on the same operands - ragged row counts (partial last chunk, uneven slabs), tiles that are not multiples of 64, both
    accumulate modes (9999 rows stay below the large regime: the register-direct pair kernel, same contract).  fp32 mode:
    exact products, 2e-6; bf16 mode: operands rounded (RNE), 8e-3 (the large-regime kernel sums the bias gradients from the
    stored values, the small-regime one from the rounded operands: both within the mode's tolerance)."""
    from position_induced_transformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(rows + n0 + 7 * math)
    x = torch.randn(rows, n0, device="cuda", generator=g)
    h = torch.randn(rows, n1, device="cuda", generator=g)
    scratch = torch.randn(rows * (n1 + n2), device="cuda", generator=g)
    dy = torch.randn(rows, n2, device="cuda", generator=g)
    dz1 = scratch[:rows * n1].view(rows, n1).double()
    dz2 = (scratch[rows * n1:].view(rows, n2) if out_gelu else dy).double()
    want = [dz1.t() @ x.double(), dz1.sum(0), dz2.t() @ h.double(), dz2.sum(0)]
    for accumulate in (0, 1):
        base = [torch.randn(n1, n0, device="cuda", generator=g), torch.randn(n1, device="cuda", generator=g),
                torch.randn(n2, n1, device="cuda", generator=g), torch.randn(n2, device="cuda", generator=g)]
        got = [b.clone() for b in base]
        rc = L.pit_mlp_bwd_params(x.data_ptr(), n0, rows, n0, n1, n2, h.data_ptr(), out_gelu, dy.data_ptr(), n2, got[0].data_ptr(),
                                  got[1].data_ptr(), got[2].data_ptr(), got[3].data_ptr(), accumulate, scratch.data_ptr(), math,
                                  torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "pit_mlp_bwd_params")
        torch.cuda.synchronize()
        for k, (a, b, w) in enumerate(zip(got, base, want)):
            ref = w + b.double() if accumulate else w
            err = float((a.double() - ref).norm() / ref.norm())
            tol = 8e-3 if math == 1 else 2e-6
            assert err <= tol, (accumulate, ("dW1", "db1", "dW2", "db2")[k], err)


# --------------------------------------------------------------------------- large-regime MLP data path (64-row tiles, fused gelu prologue)
@pytest.mark.parametrize("rows,n0,n1,n2", [(16384, 192, 64, 64), (14560, 256, 128, 128), (5120, 768, 256, 256), (9000, 320, 96, 96)])
def test_large_mlp_forward_and_backward_data_path_match_fp64(rows, n0, n1, n2):
    """pit_mlp_fwd / pit_mlp_bwd_data in the LDS-tiled regime, trailing gelu on: the forward (Z1, H, Z2, Y) and the backward
    data path - dZ2 = dY * gelu'(Z2) formed inside the dZ1 GEMM's staging for K <= 128 (gemm_lds_kernel<..., AGZ>) or by its
    own pass, dZ1, dX - against fp64 on the same operands; 64-row tiles for K <= 256, 128-/32-row tiles otherwise."""
    from position_induced_transformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(rows + n0)
    x = torch.randn(rows, n0, device="cuda", generator=g)
    w1, b1 = torch.randn(n1, n0, device="cuda", generator=g) * 0.05, torch.randn(n1, device="cuda", generator=g)
    w2, b2 = torch.randn(n2, n1, device="cuda", generator=g) * 0.05, torch.randn(n2, device="cuda", generator=g)
    z1, h = torch.empty(rows, n1, device="cuda"), torch.empty(rows, n1, device="cuda")
    z2, y = torch.empty(rows, n2, device="cuda"), torch.empty(rows, n2, device="cuda")
    dy, dx = torch.randn(rows, n2, device="cuda", generator=g), torch.empty(rows, n0, device="cuda")
    scratch = torch.empty(rows * (n1 + n2), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.pit_mlp_fwd(x.data_ptr(), n0, rows, n0, n1, n2, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 1,
                             z1.data_ptr(), h.data_ptr(), z2.data_ptr(), y.data_ptr(), n2, 0, st), "pit_mlp_fwd")
    _lib.check(L.pit_mlp_bwd_data(rows, n0, n1, n2, w1.data_ptr(), w2.data_ptr(), z1.data_ptr(), z2.data_ptr(), 1, dy.data_ptr(), n2,
                                  dx.data_ptr(), n0, scratch.data_ptr(), 0, st), "pit_mlp_bwd_data")
    torch.cuda.synchronize()
    z1r = x.double() @ w1.double().t() + b1.double()
    hr = torch.nn.functional.gelu(z1r)
    z2r = hr @ w2.double().t() + b2.double()
    gp = lambda z: 0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * np.pi) ** 0.5
    dz2 = dy.double() * gp(z2r)
    dz1 = (dz2 @ w2.double()) * gp(z1r)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    assert rel(z1, z1r) <= 1e-6 and rel(h, hr) <= 1e-6 and rel(z2, z2r) <= 1e-6 and rel(y, torch.nn.functional.gelu(z2r)) <= 1e-6
    assert rel(scratch[rows * n1:].view(rows, n2), dz2) <= 1e-6
    assert rel(scratch[:rows * n1].view(rows, n1), dz1) <= 2e-6
    assert rel(dx, dz1 @ w1.double()) <= 2e-6
