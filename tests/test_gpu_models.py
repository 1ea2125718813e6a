"""GPU: model-level parity through the drop-in nn.Module API, and the
hipGraph-captured training step."""
import numpy as np
import pytest
import torch

import golden_io as gio
import model_cases as mc
import pit_oracle as orc

pytestmark = pytest.mark.gpu
TOL_OUT, TOL_GRAD, TOL_HEAD = 1e-5, 2e-5, 2e-4


def build_model(cs, params):
    from position_induced_transformer_amd import tasks
    cfg = cs["cfg"]
    args = (cfg["space_dim"], cfg["in_dim"], cfg["out_dim"], cfg["hid_dim"], cfg["n_head"], cfg["n_blocks"])
    if cs["kind"] == "cloud":
        model = tasks.pit_elasticity(*args, None, cfg["en_loc"], cfg["de_loc"])
    else:
        cls = {"euclid": tasks.pit_darcy, "periodic1d": tasks.pit_burgers, "periodic2d": tasks.pit_darcy}[cs["metric"]]
        if cs["metric"] == "periodic2d":
            from position_induced_transformer_amd import pit as P

            class _P2d(tasks._FixedMeshForward, P.pit_periodic2d):
                pass
            cls = _P2d
        model = cls(*args, cs["mesh_ltt"].cuda(), cfg["en_loc"], cfg["de_loc"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return model.cuda()


@pytest.mark.parametrize("name", mc.CASES)
def test_model_forward_loss_gradients(name):
    from position_induced_transformer_amd import ops, utils
    fx = gio.load(name)
    cs = mc.build_case(name)
    params = gio.synth_params(cs["shapes"], int(fx["param_seed"]))
    model = build_model(cs, params)
    assert list(model.state_dict().keys()) == [str(s) for s in fx["param_names"]]
    for k in params:
        if k.endswith("lmda"):
            c = ops.head_scale(model.state_dict()[k]).cpu().numpy()
            assert np.array_equal(c, fx["c/" + k]), f"{k}: device lmda->c differs from the reference's"
    out = model(cs["mesh_in"].cuda(), cs["func_in"].cuda(), cs["mesh_out"].cuda())
    loss = utils.RelLpNorm(cs["cfg"]["out_dim"], cs["p_norm"])(cs["target"].cuda(), out)
    loss.backward()
    e, g, _, _ = gio.expect(fx, "out", out.detach().cpu().numpy())
    assert gio.rel_l2(e, g) <= TOL_OUT
    assert abs(float(loss.detach()) - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    for k, p in model.named_parameters():
        e, g, _, _ = gio.expect(fx, "grad/" + k, p.grad.cpu().numpy())
        tol = TOL_HEAD if k.endswith("lmda") else TOL_GRAD
        assert gio.rel_l2(e, g) <= tol, (k, gio.rel_l2(e, g))


def test_graph_captured_step_matches_eager():
    from position_induced_transformer_amd import tasks
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task("darcy", seed=1)
    batch = sample(4)
    eager = TrainStep(model, batch, meta["out_dim"], meta["p"])
    eager.run_eager()
    ref_loss = float(eager.loss)
    ref_grad = eager.flat.flat.clone()
    graphed = TrainStep(model, batch, meta["out_dim"], meta["p"])
    graphed.capture()
    for _ in range(3):
        graphed.replay()
    torch.cuda.synchronize()
    assert abs(float(graphed.loss) - ref_loss) <= 1e-6 * abs(ref_loss)
    assert gio.rel_l2(ref_grad.cpu().numpy(), graphed.flat.flat.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("task", ["burgers", "sod", "vorticity", "naca", "cylinder"])
def test_task_wrappers_run(task):
    """Every task configuration of the reference steps forward+backward with finite results
    and the right output shape (small batch)."""
    from position_induced_transformer_amd import tasks, utils
    model, sample, meta = tasks.make_task(task, seed=2)
    mesh_in, func_in, mesh_out, target = sample(2)
    out = model(mesh_in, func_in, mesh_out)
    assert out.shape == target.shape
    loss = utils.RelLpNorm(meta["out_dim"], meta["p"])(target, out)
    loss.backward()
    assert torch.isfinite(loss)
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_inplace_gradient_accumulation_matches_returned_gradients():
    """ops.FUSED_GRAD_ACCUMULATION: kernels adding straight into existing .grad buffers (flat
    gradient buffer) must give the same gradients as the autograd-returned ones, and must
    ACCUMULATE (two backward passes = twice the gradient)."""
    from position_induced_transformer_amd import ops, tasks, utils
    from position_induced_transformer_amd.ddp import FlatGradients
    model, sample, meta = tasks.make_task("darcy", seed=5)
    mesh_in, func_in, mesh_out, target = sample(3)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])

    loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()          # .grad is None -> returned path
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    for p in model.parameters():
        p.grad = None
    flat = FlatGradients(model.parameters())
    assert ops.FUSED_GRAD_ACCUMULATION
    loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    assert gio.rel_l2(ref.cpu().numpy(), flat.dense().cpu().numpy()) <= 1e-5
    loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    assert gio.rel_l2(2.0 * ref.cpu().numpy(), flat.dense().cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("zero_grads", [False, True])
def test_fused_adam_matches_torch_adam_with_cosine_schedule(zero_grads):
    """ddp.FlatAdam (pit_adam_step) against torch.optim.Adam + CosineAnnealingLR on the CPU, ten
    steps with fresh gradients each step (train_darcy.py:115-116,131-134); with zero_grads the
    update also clears the gradients it consumed.  Large enough for several workgroups (the step
    counter is advanced by the last one to finish)."""
    from position_induced_transformer_amd.ddp import FlatAdam, FlatGradients
    shapes = [(17, 5), (33,), (4, 1, 1), (129, 64), (700, 300)]
    cpu = [torch.nn.Parameter(torch.from_numpy(gio.synth(s, 60 + i))) for i, s in enumerate(shapes)]
    gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in cpu]
    opt = torch.optim.Adam(cpu, lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=25)
    flat = FlatGradients(gpu, flatten_params=True)
    fused = FlatAdam(flat, lr=1e-3, cosine_t_max=25, zero_grads=zero_grads)
    for step in range(10):
        for i, (pc, pg) in enumerate(zip(cpu, gpu)):
            g = torch.from_numpy(gio.synth(tuple(pc.shape), 100 * step + i)) * (1.0 + step)
            pc.grad = g.clone()
            pg.grad.copy_(g.cuda())
        opt.step()
        sched.step()
        fused.step()
        if zero_grads:
            assert float(flat.flat.abs().max()) == 0.0
    torch.cuda.synchronize()
    assert int(fused.step_count) == 10
    for pc, pg in zip(cpu, gpu):
        assert gio.rel_l2(pc.detach().numpy(), pg.detach().cpu().numpy()) <= 2e-6
    assert abs(float(fused.scalars[0]) - sched.get_last_lr()[0]) > 0          # scalars hold the rate of step 10 ...
    lr10 = 0.5 * 1e-3 * (1 + np.cos(np.pi * 9 / 25))
    assert abs(float(fused.scalars[0]) - lr10) <= 1e-9


def test_vorticity_rollout_matches_oracle():
    """3-step autoregressive rollout with BPTT (train_vorticity.py:118-126) of a reduced
    pit_vorticity (InstanceNorm after encoder and processor) against the CPU oracle."""
    from position_induced_transformer_amd import tasks, utils
    s_in, s_ltt, mem, hid, steps, b = 16, 8, 4, 32, 3, 2
    mesh = tasks.grid_mesh_2d(s_in, False, "cuda")
    ltt = tasks.grid_mesh_2d(s_ltt, False, "cuda")
    model = tasks.pit_vorticity(2, mem, 1, hid, 2, 2, ltt, 0.05, 0.05).cuda()
    shapes = orc.param_shapes(2, mem, 1, hid, 2, 2)
    params = gio.synth_params(shapes, 3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    x = torch.from_numpy(gio.synth((b, s_in, s_in, mem), 71))
    y = torch.from_numpy(gio.synth((b, s_in, s_in, steps), 72))
    loss = tasks.rollout_loss(model, mesh, x.cuda(), y.cuda(), steps, utils.RelLpNorm(1, 2))
    loss.backward()

    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in params.items()}
    mi = orc.grid_mesh_2d(s_in, False)
    lt = orc.grid_mesh_2d(s_ltt, False)
    xx, ref = x, 0.0
    for t in range(steps):
        f = orc.with_coords(mi, xx.reshape(b, -1, mem))
        out = orc.pit_apply(p, "periodic2d", False, 2, 0.05, 0.05, mi, f, lt, mi, norm_after_enc_proc=True)
        out = out.reshape(b, s_in, s_in, 1)
        ref = ref + orc.rel_lp_loss(out, y[..., t:t + 1], 1, 2)          # script order: loss(out, y)
        xx = torch.cat((xx[..., 1:], out), dim=-1)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-5 * abs(float(ref.detach()))
    for k, q in model.named_parameters():
        tol = 5e-4 if k.endswith("lmda") else 1e-4
        assert gio.rel_l2(p[k].grad.numpy(), q.grad.cpu().numpy()) <= tol, k


def _oracle_params(model):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}


def test_naca_wrapper_matches_oracle():
    """pit_naca forward (train_naca.py:47-65): latent mesh = strided sub-grid of the body-fitted
    output grid, per-sample meshes; reduced grid, against the oracle with the same slicing."""
    from position_induced_transformer_amd import tasks, utils
    torch.manual_seed(11)
    model = tasks.pit_naca(2, 2, 4, 32, 1, 2, None, 4, 4, 0.05, 0.05).cuda()
    model.x_res, model.y_res = 12, 5                                  # 45 x 19 grid -> 12 x 5 latent
    b = 2
    th = torch.linspace(0, 6.2, 40)
    outline = torch.stack((torch.cos(th), 0.3 * torch.sin(th)), -1).unsqueeze(0).repeat(b, 1, 1)
    outline = outline + 0.01 * torch.from_numpy(gio.synth((b, 40, 2), 81))
    gx, gy = torch.meshgrid(torch.linspace(-2, 2, 45), torch.linspace(-1, 1, 19), indexing="ij")
    grid = torch.stack((gx, gy), -1).unsqueeze(0).repeat(b, 1, 1, 1) + 0.01 * torch.from_numpy(gio.synth((b, 45, 19, 2), 82))
    y = torch.from_numpy(gio.synth((b, 45, 19, 4), 83))
    out = model(outline.cuda(), outline.cuda(), grid.cuda())
    loss = utils.RelLpNorm(4, 2)(y.cuda(), out)
    loss.backward()
    p = _oracle_params(model)
    ltt = grid[:, ::4, ::4][:, :12, :5].reshape(b, -1, 2)
    flat = grid.reshape(b, -1, 2)
    ref = orc.pit_apply(p, "euclid", True, 2, 0.05, 0.05, outline, outline, ltt, flat).reshape(b, 45, 19, 4)
    rl = orc.rel_lp_loss(y, ref, 4, 2)
    rl.backward()
    assert gio.rel_l2(ref.detach().numpy(), out.detach().cpu().numpy()) <= TOL_OUT
    for k, q in model.named_parameters():
        tol = TOL_HEAD if k.endswith("lmda") else TOL_GRAD
        assert gio.rel_l2(p[k].grad.numpy(), q.grad.cpu().numpy()) <= tol, k


def test_cylinder_wrapper_residual_matches_oracle():
    """pit_cylinder forward (train_cylinder.py:40-52): unstructured fixed mesh, one head, the
    input added back onto the prediction; loss called as myloss(out, y) (train_cylinder.py:101)."""
    from position_induced_transformer_amd import tasks, utils
    torch.manual_seed(12)
    mesh = torch.from_numpy(gio.synth((300, 2), 84, 0.0, 2.0))
    ltt = mesh[::4].contiguous()
    model = tasks.pit_cylinder(2, 3, 3, 32, 1, 2, ltt.cuda(), 0.05, 0.05).cuda()
    x = torch.from_numpy(gio.synth((3, 300, 3), 85))
    y = torch.from_numpy(gio.synth((3, 300, 3), 86))
    out = model(mesh.cuda(), x.cuda(), mesh.cuda())
    loss = utils.RelLpNorm(3, 2)(out, y.cuda())
    loss.backward()
    p = _oracle_params(model)
    f = orc.with_coords(mesh, x)
    ref = orc.pit_apply(p, "euclid", False, 2, 0.05, 0.05, mesh, f, ltt, mesh) + x
    rl = orc.rel_lp_loss(ref, y, 3, 2)
    rl.backward()
    assert gio.rel_l2(ref.detach().numpy(), out.detach().cpu().numpy()) <= TOL_OUT
    assert abs(float(loss.detach()) - float(rl.detach())) <= 1e-5 * abs(float(rl.detach()))
    for k, q in model.named_parameters():
        tol = TOL_HEAD if k.endswith("lmda") else TOL_GRAD
        assert gio.rel_l2(p[k].grad.numpy(), q.grad.cpu().numpy()) <= tol, k


def test_deferred_head_finish_matches_per_layer_finish():
    """ops.DEFER_HEAD_FINISH: one pit_posatt_dhead_finish launch at the end of the backward pass must
    give the lmda gradients of the per-layer finishing kernels (in-place .grad path), accumulate over
    two passes, and leave nothing pending."""
    from position_induced_transformer_amd import ops, tasks, utils
    from position_induced_transformer_amd.ddp import FlatGradients
    model, sample, meta = tasks.make_task("darcy", seed=7)
    mesh_in, func_in, mesh_out, target = sample(3)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])
    flat = FlatGradients(model.parameters())
    heads = [k for k, _ in model.named_parameters() if k.endswith("lmda")]
    assert len(heads) == 6

    def grads(defer, passes):
        old = ops.DEFER_HEAD_FINISH
        ops.DEFER_HEAD_FINISH = defer
        try:
            flat.zero_()
            for _ in range(passes):
                loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
            torch.cuda.synchronize()
            assert not ops._PENDING_HEADS
            return {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}
        finally:
            ops.DEFER_HEAD_FINISH = old

    ref, got = grads(False, 1), grads(True, 1)
    for k in ref:
        # (weight gradients are sums of fp32 atomics over row slabs: run-to-run order differences, ~1 ulp)
        assert gio.rel_l2(ref[k], got[k]) <= (1e-6 if k.endswith("lmda") else 0.0) + 5e-7, k
    twice = grads(True, 2)
    for k in heads:
        assert gio.rel_l2(2.0 * ref[k], twice[k]) <= 1e-5, k


def test_bench_prints_exactly_one_json_line():
    """bench.py's contract with the driver: stdout is ONE JSON line with the agreed keys (logs and
    library banners go to stderr)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3",
                        "--no-extras", "--cpu-iters", "3"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["steps"] == 20 and rec["warmup"] == 3 and rec["n_gpus"] == 1 and rec["vs_baseline"] is None
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rec["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in rec["cpu_baseline"], key
    assert "workload" in rec["config"] and rec["value"] > 0


@pytest.mark.parametrize("kind", ["fixed", "batched", "periodic1d"])
def test_dist2att_and_convolution_reproduce_forward(kind):
    """The reference's two-step API (pit.py:46-57): convolution(dist2att(...), inputs) must equal the
    fused forward, and dist2att must equal the oracle's dense attention weights."""
    from position_induced_transformer_amd import pit as P
    g = torch.Generator().manual_seed(3)
    if kind == "batched":
        layer = P.posatt_cross(2, 8, 0.1).cuda()
        mo, mi = torch.rand(2, 40, 2, generator=g), torch.rand(2, 90, 2, generator=g)
        metric, batched = "euclid", True
    elif kind == "fixed":
        layer = P.posatt_cross_fixed(2, 8, 0.1).cuda()
        mo, mi = torch.rand(40, 2, generator=g), torch.rand(90, 2, generator=g)
        metric, batched = "euclid", False
    else:
        layer = P.posatt_cross_periodic1d(2, 8, 0.1).cuda()
        mi = torch.linspace(0, 1, 129)[:-1].reshape(-1, 1)
        mo = torch.rand(40, 1, generator=g)
        metric, batched = "periodic1d", False
    x = torch.randn(2, mi.shape[-2], 8, generator=g)
    fused = layer(mo.cuda(), mi.cuda(), x.cuda())
    A = layer.dist2att(mo.cuda(), mi.cuda(), layer.lmda, layer.locality)
    two_step = layer.convolution(A, x.cuda())
    assert gio.rel_l2(fused.detach().cpu().numpy(), two_step.detach().cpu().numpy()) <= 2e-6
    c = orc.head_scale(layer.lmda.detach().cpu())
    ref = orc.attention_weights(orc.sqdist(metric, mo, mi), c, layer.locality, batched)
    assert gio.rel_l2(ref.numpy(), A.detach().cpu().numpy()) <= 1e-5
