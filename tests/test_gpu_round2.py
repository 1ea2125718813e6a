"""GPU: the boundary items of round 2 - torch.compile survival (train_darcy.py:112,150,152), the
lmda->c routes on unselected parameter seeds (regular grids: tie shells), RelMaxNorm on device,
graph replay with NEW per-sample meshes, exception safety of the deferred d(lmda) finish, a
200-step captured training run against the eager loop."""
import json
import os

import numpy as np
import pytest
import torch

import golden_io as gio
import model_cases as mc
import pit_oracle as orc
from test_gpu_models import build_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------- torch.compile
def test_torch_compile_model_matches_eager_and_traces_nothing():
    """`model = torch.compile(model)` exactly as train_darcy.py:112, forward + backward through the
    compiled wrapper == eager to 1e-6; dynamo compiles NO graph (nothing of the hot path may go to
    inductor/Triton); the checkpoint of the wrapper has the `_orig_mod.` keys (train_darcy.py:150)
    and `torch._dynamo.disable(model)` (train_darcy.py:152) still runs it."""
    import torch._dynamo as dynamo
    from position_induced_transformer_amd import pit as P, utils

    class pit_darcy(P.pit_fixed):                 # the script's class, train_darcy.py:25-59
        def forward(self, mesh_in, func_in, mesh_out):
            size = mesh_out.shape[:-1]
            mesh_in = mesh_in.reshape(-1, self.space_dim)
            func_in = func_in.reshape(func_in.shape[0], -1, self.in_dim)
            mesh_out = mesh_out.reshape(-1, self.space_dim)
            func_in = torch.cat((torch.tile(mesh_in.unsqueeze(0), [func_in.shape[0], 1, 1]), func_in), -1)
            func_ltt = self.encoder(mesh_in, func_in, self.mesh_ltt)
            func_ltt = self.processor(func_ltt, self.mesh_ltt)
            func_out = self.decoder(self.mesh_ltt, func_ltt, mesh_out)
            return func_out.reshape(func_in.shape[0], *size, self.out_dim)

    torch.manual_seed(3)
    g43, g16 = orc.grid_mesh_2d(43).reshape(43, 43, 2).cuda(), orc.grid_mesh_2d(16).reshape(16, 16, 2).cuda()
    model = pit_darcy(2, 1, 1, 64, 2, 4, g16, 0.02, 0.02).cuda()
    x, y = torch.randn(4, 43, 43, 1, device="cuda"), torch.randn(4, 43, 43, 1, device="cuda")
    loss_fn = utils.RelLpNorm(1, 2)

    out_e = model(g43, x, g43)
    loss_fn(y, out_e).backward()
    grads_e = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()

    dynamo.reset()
    dynamo.utils.counters.clear()
    compiled = torch.compile(model)
    out_c = compiled(g43, x, g43)
    loss_fn(y, out_c).backward()
    torch.cuda.synchronize()
    assert gio.rel_l2(out_e.detach().cpu().numpy(), out_c.detach().cpu().numpy()) <= 1e-6
    for k, p in model.named_parameters():
        assert gio.rel_l2(grads_e[k].cpu().numpy(), p.grad.cpu().numpy()) <= (1e-5 if k.endswith("lmda") else 1e-6), k
    assert dynamo.utils.counters["stats"].get("unique_graphs", 0) == 0, dict(dynamo.utils.counters["stats"])
    keys = list(compiled.state_dict().keys())
    assert keys[0] == "_orig_mod.down.lmda" and all(k.startswith("_orig_mod.") for k in keys)
    with torch.no_grad():
        out_d = torch._dynamo.disable(compiled)(g43, x, g43)
    assert gio.rel_l2(out_e.detach().cpu().numpy(), out_d.cpu().numpy()) <= 1e-6


# --------------------------------------------------------------------------- lmda -> c routes
# seeds of the route sweep: 200 produced profiles/r02_lmda_route_sweep.json (PIT_SWEEP_SEEDS=200 re-runs it); the suite's default
# keeps the same seeds' first 100 (Burgers, where 1 of 200 seeds is above the bound: 50) - the oracle on the host is what the sweep's
# time goes to (79 of the suite's 201 s with 200).  (The first 50 Darcy seeds alone have 16 above 1e-5 = 0.32: too few for the bound.)
SWEEP_SEEDS = int(os.environ.get("PIT_SWEEP_SEEDS", "100"))
# Fraction of unselected seeds whose device-route output is more than 1e-5 from the oracle ON THE SAME
# HOST.  It is a property of the host's libm as much as of the kernels: measured 0.22 (Darcy) on the
# MI355X box (EPYC 9575F: MKL's VML kernels for AMD CPUs disagree with the correctly rounded c for ~19 %
# of lmda, 185 of 200 seeds have at least one differing c); 0.005 (Burgers: 1 of 200) -
# profiles/r02_lmda_route_sweep.json.  The test fails if the fraction grows beyond this bound.
# Guards set to what was measured plus a margin for host-to-host variation of the libm (round 2 had 0.35 for both).
MAX_FRACTION_ABOVE_1E5 = {"F9_model_darcy": 0.25, "F11_model_burgers": 0.02}


@pytest.mark.parametrize("name", ["F9_model_darcy", "F11_model_burgers"])
def test_unselected_seed_sweep_device_and_host_scale_routes(name):
    """VERDICT r1 weak #1.  SWEEP_SEEDS (100, Burgers 50; 200 for the committed record) parameter seeds taken as they come (nothing selected) on the two
    regular-grid models, each compared with the oracle on THIS host (whose sin/tan are ATen-CPU's):
      * route 'host' (c by the reference's own torch-CPU ops, injected): every seed <= 1e-5;
      * route 'device' (c evaluated in the kernels): <= 1e-5 whenever all c are bit-equal to the
        host's; otherwise a differing c can move a tie shell of the grid across the quantile
        threshold - the fraction of seeds above 1e-5 is reported and bounded.
    The distribution goes to gpurun_out/lmda_route_sweep_<name>.json (summarised in profiles/)."""
    from position_induced_transformer_amd import ops
    cs = mc.build_case(name)
    cfg = cs["cfg"]
    mi = cs["mesh_in"].reshape(-1, cfg["space_dim"])
    func = cs["func_in"].reshape(2, -1, cfg["in_dim"])
    feats = orc.with_coords(mi, func)
    rows = []
    model = None
    for seed in range(1000, 1000 + (SWEEP_SEEDS if name == "F9_model_darcy" else max(1, SWEEP_SEEDS // 2))):
        params = gio.synth_params(cs["shapes"], seed)
        p = {k: torch.from_numpy(v) for k, v in params.items()}
        with torch.no_grad():
            ref = orc.pit_apply(p, cs["metric"], False, cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"], mi, feats,
                                cs["mesh_ltt"].reshape(-1, cfg["space_dim"]), mi).numpy()
        if model is None:
            model = build_model(cs, params)
        else:
            model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        ulps = []
        for k, v in p.items():
            if k.endswith("lmda"):
                c_host = orc.head_scale(v).numpy().reshape(-1)
                c_dev = ops.head_scale(v.cuda()).cpu().numpy().reshape(-1)
                ulps.append(int(np.abs(c_host.view(np.int32).astype(np.int64) - c_dev.view(np.int32).astype(np.int64)).max()))
        with torch.no_grad():
            out_dev = model(cs["mesh_in"].cuda(), cs["func_in"].cuda(), cs["mesh_out"].cuda()).cpu().numpy()
            with ops.head_scale_route("host"):
                out_host = model(cs["mesh_in"].cuda(), cs["func_in"].cuda(), cs["mesh_out"].cuda()).cpu().numpy()
        rows.append(dict(seed=seed, max_ulp=max(ulps), n_layers_differ=int(sum(u > 0 for u in ulps)),
                         err_device=gio.rel_l2(ref.reshape(-1), out_dev.reshape(-1)),
                         err_host=gio.rel_l2(ref.reshape(-1), out_host.reshape(-1))))
    e_dev = np.array([r["err_device"] for r in rows])
    e_host = np.array([r["err_host"] for r in rows])
    equal = np.array([r["max_ulp"] == 0 for r in rows])
    summary = dict(model=name, seeds=len(rows), seeds_with_all_c_bit_equal=int(equal.sum()),
                   max_ulp_seen=int(max(r["max_ulp"] for r in rows)),
                   host_route=dict(max=float(e_host.max()), median=float(np.median(e_host)), above_1e5=int((e_host > 1e-5).sum())),
                   device_route_c_equal=dict(n=int(equal.sum()), max=float(e_dev[equal].max()) if equal.any() else None,
                                             above_1e5=int((e_dev[equal] > 1e-5).sum())),
                   device_route_c_differs=dict(n=int((~equal).sum()),
                                               max=float(e_dev[~equal].max()) if (~equal).any() else None,
                                               median=float(np.median(e_dev[~equal])) if (~equal).any() else None,
                                               above_1e5=int((e_dev[~equal] > 1e-5).sum()),
                                               above_1e4=int((e_dev[~equal] > 1e-4).sum())),
                   device_route_fraction_above_1e5=float((e_dev > 1e-5).mean()))
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"lmda_route_sweep_{name}.json"), "w") as f:
            json.dump(dict(summary=summary, rows=rows), f, indent=1)
    except OSError:
        pass
    print(json.dumps(summary))
    # host route: <= 1e-5, except on seeds where the REFERENCE's own fp32 arithmetic is that far from exact
    # (tiny, cancelling outputs): there the bound is twice the fp32-vs-fp64 distance of the oracle itself
    for r in rows:
        if r["err_host"] > 1e-5:
            p64 = {k: torch.from_numpy(v).double() for k, v in gio.synth_params(cs["shapes"], r["seed"]).items()}
            with torch.no_grad():
                kw = dict(dtype=torch.float64)
                a64 = orc.pit_apply(p64, cs["metric"], False, cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"], mi.to(**kw),
                                    feats.to(**kw), cs["mesh_ltt"].reshape(-1, cfg["space_dim"]).to(**kw), mi.to(**kw))
                p32 = {k: v.float() for k, v in p64.items()}
                a32 = orc.pit_apply(p32, cs["metric"], False, cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"], mi, feats,
                                    cs["mesh_ltt"].reshape(-1, cfg["space_dim"]), mi)
            own = float((a32.double() - a64).norm() / a64.norm())
            assert r["err_host"] <= 2.0 * own, (r, own)
            r["oracle_fp32_vs_fp64"] = own
    assert summary["host_route"]["above_1e5"] <= 2, summary
    assert summary["device_route_c_equal"]["above_1e5"] == 0, summary
    # (the bound is the 200-seed figure; a shorter sweep gets the sampling slack of its size: 0.24 on the first 100 Darcy seeds)
    slack = 0.0 if SWEEP_SEEDS >= 200 else 0.05
    assert summary["device_route_fraction_above_1e5"] <= MAX_FRACTION_ABOVE_1E5[name] + slack, summary


def test_host_route_gradients_match_the_oracle_on_a_seed_where_c_differs():
    """Route 'host' end to end (forward, loss, every gradient incl. d lmda through the torch-CPU chain
    rule) on the first Darcy seed whose device c differs from the host's: the case the round-1
    goldens avoided by selecting seeds."""
    from position_induced_transformer_amd import ops, utils
    cs = mc.build_case("F9_model_darcy")
    cfg = cs["cfg"]
    for seed in range(5000, 5400):
        params = gio.synth_params(cs["shapes"], seed)
        differs = [k for k, v in params.items() if k.endswith("lmda") and not np.array_equal(
            orc.head_scale(torch.from_numpy(v)).numpy(), ops.head_scale(torch.from_numpy(v).cuda()).cpu().numpy())]
        if differs:
            break
    else:
        pytest.skip("no seed in range with a differing c on this host")
    model = build_model(cs, params)
    with ops.head_scale_route("host"):
        out = model(cs["mesh_in"].cuda(), cs["func_in"].cuda(), cs["mesh_out"].cuda())
        loss = utils.RelLpNorm(1, 2)(cs["target"].cuda(), out)
        loss.backward()
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in params.items()}
    mi = cs["mesh_in"].reshape(-1, 2)
    ref = orc.pit_apply(p, "euclid", False, cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"], mi,
                        orc.with_coords(mi, cs["func_in"].reshape(2, -1, 1)), cs["mesh_ltt"].reshape(-1, 2), mi)
    ref_loss = orc.rel_lp_loss(cs["target"], ref.reshape(2, 43, 43, 1), 1, 2)
    ref_loss.backward()
    assert gio.rel_l2(ref.detach().numpy().reshape(-1), out.detach().cpu().numpy().reshape(-1)) <= 1e-5
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    for k, q in model.named_parameters():
        tol = 2e-4 if k.endswith("lmda") else 2e-5
        assert gio.rel_l2(p[k].grad.numpy().reshape(-1), q.grad.cpu().numpy().reshape(-1)) <= tol, (k, differs)


# --------------------------------------------------------------------------- RelMaxNorm
@pytest.mark.parametrize("shape,out_dim", [((3, 1024, 1), 1), ((2, 221, 51, 4), 4), ((5, 7, 3), 3)])
def test_rel_max_norm_on_device(shape, out_dim):
    """utils.py:59-77 on device tensors runs pit_rel_max_norm; equal to the reference formula."""
    from position_induced_transformer_amd import utils
    g = torch.Generator().manual_seed(11)
    t, q = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
    tr, qr = t.reshape(shape[0], -1, out_dim), q.reshape(shape[0], -1, out_dim)
    want = torch.sum(torch.mean(torch.max(torch.abs(tr - qr), dim=1)[0] / torch.max(torch.abs(tr), dim=1)[0], dim=-1))
    with torch.no_grad():
        got = utils.RelMaxNorm(out_dim)(t.cuda(), q.cuda())
        again = utils.RelMaxNorm(out_dim)(t.cuda(), q.cuda())          # the accumulator cleans itself
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
    assert float(got) == float(again)
    with pytest.raises(NotImplementedError):
        utils.RelMaxNorm(out_dim)(t.cuda(), q.cuda().requires_grad_(True))


# --------------------------------------------------------------------------- captured step, new meshes
@pytest.mark.parametrize("task,batch", [("elasticity", 2), ("naca", 2)])
@pytest.mark.parametrize("math", ["fp32", "bf16"])
def test_captured_step_replays_new_per_sample_meshes(task, batch, math):
    """BASELINE config 5 (NACA: per-sample meshes, hipGraph-captured step) and Elasticity: the captured
    step rebuilds its selection plans from the static mesh buffers, so after set_batch(...meshes...)
    a replay must equal an eager step on the NEW clouds (train_naca.py:62-65, train_elasticity.py:46)."""
    from position_induced_transformer_amd import ops, tasks
    from position_induced_transformer_amd.engine import TrainStep
    with ops.math_mode(math):
        model, sample, meta = tasks.make_task(task, seed=7)
        first, second = sample(batch), sample(batch)
        assert not torch.equal(first[2], second[2])
        static = tuple(t.clone() for t in first)
        if task == "naca":
            static = (static[0], static[0], static[2], static[3])          # mesh_in IS func_in (train_naca.py:95)
        step = TrainStep(model, static, meta["out_dim"], meta["p"])
        step.capture()
        step.replay()
        step.set_batch(second[1], second[3], mesh_in=second[0], mesh_out=second[2])
        step.replay()
        torch.cuda.synchronize()
        got_loss, got = float(step.loss), step.flat.flat.clone()
        eager = TrainStep(model, second, meta["out_dim"], meta["p"], flat=step.flat)
        eager.run_eager()
        torch.cuda.synchronize()
        tol = 1e-5 if math == "fp32" else 1e-3         # identical kernels either way; atomics reorder sums
        assert abs(got_loss - float(eager.loss)) <= tol * abs(float(eager.loss))
        assert gio.rel_l2(eager.flat.flat.cpu().numpy(), got.cpu().numpy()) <= 10 * tol
        # and the first batch's result differs (the replay really consumed the new clouds)
        step.set_batch(first[1], first[3], mesh_in=first[0], mesh_out=first[2])
        step.replay()
        torch.cuda.synchronize()
        assert abs(float(step.loss) - got_loss) > 1e-4 * abs(got_loss)


def test_set_batch_rejects_meshes_of_fixed_mesh_models():
    from position_induced_transformer_amd import tasks
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task("darcy", seed=1)
    b = sample(2)
    step = TrainStep(model, b, meta["out_dim"], meta["p"])
    with pytest.raises(ValueError, match="batch-free"):
        step.set_batch(b[1], b[3], mesh_in=b[0])


# --------------------------------------------------------------------------- exception safety on the device
def test_backward_that_raises_midway_does_not_poison_the_next_step():
    """ADVICE r1 (ops.py deferred finish): a backward pass that dies after some attention layers have
    loaded their fp64 d(scale) accumulators must not leak those partial sums - or a missing
    end-of-pass callback - into the next step's d(lmda)."""
    from position_induced_transformer_amd import tasks, utils
    from position_induced_transformer_amd.ddp import FlatGradients
    from position_induced_transformer_amd import ops
    for seed in range(9, 60):             # a seed whose device c equals this host's in every layer (see the sweep)
        model, sample, meta = tasks.make_task("darcy", seed=seed)
        if all(np.array_equal(orc.head_scale(v.detach().cpu()).numpy(), ops.head_scale(v.detach()).cpu().numpy())
               for k, v in model.named_parameters() if k.endswith("lmda")):
            break
    mesh_in, func_in, mesh_out, target = sample(2)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])
    flat = FlatGradients(model.parameters())               # in-place accumulation + deferred finishes

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    # die inside the encoder's backward: decoder + processor layers have already deferred
    orig = model.encoder
    model.encoder = lambda a, b, c: Boom.apply(orig(a, b, c))
    with pytest.raises(RuntimeError, match="boom"):
        loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    model.encoder = orig
    flat.zero_()
    loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    torch.cuda.synchronize()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    mi = mesh_in.cpu().reshape(-1, 2)
    ref = orc.pit_apply(p, "euclid", False, 4, 0.02, 0.02, mi, orc.with_coords(mi, func_in.cpu().reshape(2, -1, 1)),
                        model.mesh_ltt.cpu(), mi).reshape(2, 43, 43, 1)
    orc.rel_lp_loss(target.cpu(), ref, 1, 2).backward()
    for k, q in model.named_parameters():
        tol = 2e-4 if k.endswith("lmda") else 2e-5
        assert gio.rel_l2(p[k].grad.numpy().reshape(-1), q.grad.cpu().numpy().reshape(-1)) <= tol, k
    # a third step equals the second (nothing accumulates across steps)
    second = flat.flat.clone()
    flat.zero_()
    loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    torch.cuda.synchronize()
    assert gio.rel_l2(second.cpu().numpy(), flat.flat.cpu().numpy()) <= 1e-5


# --------------------------------------------------------------------------- 200 captured steps
def test_200_captured_steps_train_and_equal_the_eager_loop():
    """engine.TrainStep + ddp.FlatAdam: 200 replays of the captured step (fixed batch) bring the loss
    down and land on the same parameters as 200 eager steps of the same step function."""
    from position_induced_transformer_amd import tasks
    from position_induced_transformer_amd.ddp import FlatAdam, FlatGradients
    from position_induced_transformer_amd.engine import TrainStep

    def run(graphed):
        model, sample, meta = tasks.make_task("darcy", seed=21)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(4, 43, 43, 1, generator=g).cuda()
        k = torch.ones(1, 1, 5, 5, device="cuda") / 25.0
        y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), k, padding=2).permute(0, 2, 3, 1).contiguous()
        mesh = sample(1)[0]
        flat = FlatGradients(model.parameters(), flatten_params=True)
        opt = FlatAdam(flat, lr=1e-3, cosine_t_max=203, zero_grads=True)
        step = TrainStep(model, (mesh, x, mesh, y), meta["out_dim"], meta["p"], optimizer=opt, flat=flat)
        if graphed:
            step.capture(warmup=3)                       # three eager steps, then the capture (records, runs nothing)
        else:
            for _ in range(3):
                step.run_eager()
        first = None
        for i in range(200):
            step.replay() if graphed else step.run_eager()
            if i == 0:
                first = float(step.loss)
        torch.cuda.synchronize()
        return (first, float(step.loss)), flat.flat_params.clone(), int(opt.step_count)

    l_e, p_e, n_e = run(False)
    l_g, p_g, n_g = run(True)
    assert n_e == 203 and n_g == 203
    assert l_e[1] < 0.8 * l_e[0] and l_g[1] < 0.8 * l_g[0], (l_e, l_g)
    assert abs(l_g[0] - l_e[0]) <= 1e-4 * abs(l_e[0])     # same state after the three warm-up steps
    assert abs(l_g[1] - l_e[1]) <= 5e-2 * abs(l_e[1])     # fp32 atomics reorder sums; 200 Adam steps amplify


# --------------------------------------------------------------------------- row f3: the 20-step rollout
def _vorticity_full(batch, seed):
    from position_induced_transformer_amd import tasks
    model, sample, meta = tasks.make_task("vorticity", seed=seed)
    g = torch.Generator().manual_seed(seed)
    mesh = sample(1)[0]
    x = torch.randn(batch, 64, 64, 10, generator=g).cuda()
    y = torch.randn(batch, 64, 64, 20, generator=g).cuda()
    return model, mesh, x, y, meta


def test_rollout_20_steps_full_size_matches_oracle_bptt():
    """train_vorticity.py:118-126 at the script's size (64x64 grid -> 16x16 latent, hid 256, H=2, 4 blocks,
    InstanceNorm, 20 autoregressive steps, loss(out, y_t) summed, ONE backward through all 20 forwards) on a
    reduced batch of 1: loss and every parameter gradient against the oracle's BPTT on the CPU.  Head scales
    through the host route (20 chained forwards on a periodic GRID: one moved tie shell would swamp the
    comparison - see the sweep).  Tolerances are those of a 20-fold composition of the 1e-5 model."""
    from position_induced_transformer_amd import ops, tasks, utils
    model, mesh, x, y, meta = _vorticity_full(1, seed=31)
    loss_fn = utils.RelLpNorm(1, 2)
    with ops.head_scale_route("host"):
        loss = tasks.rollout_loss(model, mesh, x, y, 20, loss_fn)
        loss.backward()
    torch.cuda.synchronize()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    mi, ltt = mesh.cpu().reshape(-1, 2), model.mesh_ltt.cpu()
    xc, yc = x.cpu(), y.cpu()
    ref_loss = 0.0
    for t in range(20):
        f = orc.with_coords(mi, xc.reshape(1, -1, 10))
        out = orc.pit_apply(p, "periodic2d", False, 4, 0.02, 0.02, mi, f, ltt, mi, norm_after_enc_proc=True).reshape(1, 64, 64, 1)
        ref_loss = ref_loss + orc.rel_lp_loss(out, yc[..., t:t + 1], 1, 2)
        xc = torch.cat((xc[..., 1:], out), dim=-1)
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss)) <= 2e-5 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    worst = 0.0
    for k, q in model.named_parameters():
        e = gio.rel_l2(p[k].grad.numpy().reshape(-1), q.grad.cpu().numpy().reshape(-1))
        worst = max(worst, e if not k.endswith("lmda") else 0.0)
        assert e <= (5e-3 if k.endswith("lmda") else 5e-4), (k, e)
    print("rollout-20 full size: loss", float(loss.detach()), "max weight-grad rel-L2", worst)


@pytest.mark.parametrize("recompute", [False, True])
def test_rollout_step_graph_equals_eager_and_recompute_equals_plain(recompute):
    """engine.RolloutStep: the captured 20-step optimiser step == the eager one (3 steps here, full size,
    batch 2), with and without activation recompute; recompute must not change the gradient."""
    from position_induced_transformer_amd.engine import RolloutStep
    model, mesh, x, y, meta = _vorticity_full(2, seed=32)
    eager = RolloutStep(model, (mesh, x, y[..., :3].contiguous()), 3, 1, 2, recompute=False)
    eager.run_eager()
    torch.cuda.synchronize()
    ref_loss, ref = float(eager.loss), eager.flat.flat.clone()
    step = RolloutStep(model, (mesh, x, y[..., :3].contiguous()), 3, 1, 2, recompute=recompute, flat=eager.flat)
    step.run_eager()
    torch.cuda.synchronize()
    assert abs(float(step.loss) - ref_loss) <= 1e-6 * abs(ref_loss)
    assert gio.rel_l2(ref.cpu().numpy(), step.flat.flat.cpu().numpy()) <= 2e-5
    # (round 3: the recomputing rollout is an autograd.Function that re-runs the step - capturable, unlike
    # torch.utils.checkpoint)
    step.capture()
    step.replay()
    step.replay()
    torch.cuda.synchronize()
    assert abs(float(step.loss) - ref_loss) <= 1e-6 * abs(ref_loss)
    assert gio.rel_l2(ref.cpu().numpy(), step.flat.flat.cpu().numpy()) <= 2e-5


# --------------------------------------------------------------------------- row f4: long-J inference, Cylinder
def test_zssr_421_forward_matches_oracle():
    """train_darcy.py:152-178: the 43x43-trained Darcy model evaluated at 421x421 (J = 177 241 keys per encoder
    row, 177 241 decoder rows) under no_grad, batch 1, against the oracle's dense evaluation on the CPU
    (2 x 256 x 177 241 attention weights per direction).  Streaming selection (rows > 4096 keys) and candidate
    lists of ~4.4 k keys per latent point are what this exercises."""
    import time
    from position_induced_transformer_amd import ops, tasks
    model, _, _ = tasks.make_task("darcy", seed=41)
    g = torch.Generator().manual_seed(41)
    mesh = tasks.grid_mesh_2d(421, True, "cuda")
    x = torch.randn(1, 421, 421, 1, generator=g).cuda()
    with torch.no_grad(), ops.head_scale_route("host"):
        out = model(mesh, x, mesh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(mesh, x, mesh)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) * 1e3
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    mi = mesh.cpu().reshape(-1, 2)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = orc.pit_apply(p, "euclid", False, 4, 0.02, 0.02, mi, orc.with_coords(mi, x.cpu().reshape(1, -1, 1)),
                            model.mesh_ltt.cpu(), mi).reshape(1, 421, 421, 1)
    cpu_s = time.perf_counter() - t0
    err = gio.rel_l2(ref.numpy().reshape(-1), out.cpu().numpy().reshape(-1))
    print(f"ZSSR 421x421 b=1: rel-L2 {err:.3e}; HIP forward {gpu_ms:.2f} ms (plans cached), oracle on this host {cpu_s:.1f} s")
    assert out.shape == (1, 421, 421, 1) and err <= 1e-5, err


def test_cylinder_full_size_matches_oracle():
    """train_cylinder.py:55-84 at the script's size: 4390 unstructured points -> 896 latent points -> 4390,
    hid 256, 1 head, 4 blocks, locality 0.01, residual connection (train_cylinder.py:52); batch 2 here
    (the script's 200 is timed by profiles/r02_tasks): forward, loss (myloss(out, y), :101) and all gradients."""
    from position_induced_transformer_amd import ops, tasks, utils
    model, sample, meta = tasks.make_task("cylinder", seed=43)
    mesh_in, func_in, mesh_out, target = sample(2)
    with ops.head_scale_route("host"):
        out = model(mesh_in, func_in, mesh_out)
        loss = utils.RelLpNorm(3, 2)(out, target)
        loss.backward()
    torch.cuda.synchronize()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    mi = mesh_in.cpu()
    f = orc.with_coords(mi, func_in.cpu())
    ref = orc.pit_apply(p, "euclid", False, 4, 0.01, 0.01, mi, f, model.mesh_ltt.cpu(), mi) + func_in.cpu()
    ref_loss = orc.rel_lp_loss(ref, target.cpu(), 3, 2)
    ref_loss.backward()
    assert gio.rel_l2(ref.detach().numpy().reshape(-1), out.detach().cpu().numpy().reshape(-1)) <= 1e-5
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    for k, q in model.named_parameters():
        tol = 2e-4 if k.endswith("lmda") else 2e-5
        assert gio.rel_l2(p[k].grad.numpy().reshape(-1), q.grad.cpu().numpy().reshape(-1)) <= tol, k


# --------------------------------------------------------------------------- data parallel with the HIP compute
def _dp_worker(rank, world, port, out_dir):
    import os
    import sys
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    for pth in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from position_induced_transformer_amd import tasks
    from position_induced_transformer_amd.ddp import FlatGradients, broadcast_parameters, shard_batch
    from position_induced_transformer_amd.engine import TrainStep
    torch.cuda.set_device(0)                              # both ranks share the one GPU of the test box
    model, sample, meta = tasks.make_task("darcy", seed=50 + rank)     # different initialisations ...
    broadcast_parameters(model)                                         # ... made rank 0's
    g = torch.Generator().manual_seed(77)
    mesh = sample(1)[0]
    x, y = torch.randn(6, 43, 43, 1, generator=g).cuda(), torch.randn(6, 43, 43, 1, generator=g).cuda()
    sl = shard_batch(6, rank, world)
    step = TrainStep(model, (mesh, x[sl].contiguous(), mesh, y[sl].contiguous()), meta["out_dim"], meta["p"], all_reduce=True)
    step.run_eager()                                      # HIP forward/backward on the shard + ONE flat all-reduce
    torch.cuda.synchronize()
    if rank == 0:
        single = TrainStep(model, (mesh, x, mesh, y), meta["out_dim"], meta["p"], all_reduce=False,
                           flat=FlatGradients(model.parameters()))
        got = step.flat.flat.clone()                      # (FlatGradients above re-pointed .grad: keep the reduced copy)
        single.run_eager()
        torch.cuda.synchronize()
        want = single.flat.flat
        err = float((got - want).norm() / want.norm())
        np.save(os.path.join(out_dir, "dp_err.npy"), np.asarray([err, float(want.abs().sum())]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_flat_allreduce_equals_single_rank_hip_gradient(tmp_path):
    """the data-parallel step with the HIP compute on every rank (two processes sharing
    this box's one GPU, gloo for the exchange): shard -> forward/backward kernels -> ONE all-reduce of the flat
    buffer == the single-process gradient of the whole batch (SUM reduction: RelLpNorm sums over the batch)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    err, mass = np.load(os.path.join(tmp_path, "dp_err.npy"))
    assert mass > 0 and err <= 2e-5, (err, mass)


# --------------------------------------------------------------------------- full-size task configurations vs the oracle
def _compare_with_oracle(model, out, loss, ref, ref_loss, p, tol_out=1e-5):
    assert gio.rel_l2(ref.detach().numpy().reshape(-1), out.detach().cpu().numpy().reshape(-1)) <= tol_out
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-5 * abs(float(ref_loss.detach()))
    he, hg = [], []
    for k, q in model.named_parameters():
        if k.endswith("lmda"):               # judged as ONE vector: a decoder's d(lmda) can be 1e-9 of the others
            he.append(p[k].grad.numpy().reshape(-1)); hg.append(q.grad.cpu().numpy().reshape(-1))
        else:
            assert gio.rel_l2(p[k].grad.numpy().reshape(-1), q.grad.cpu().numpy().reshape(-1)) <= 2e-5, k
    assert gio.rel_l2(np.concatenate(he), np.concatenate(hg)) <= 2e-4


@pytest.mark.parametrize("batch", [2, 3])
def test_naca_full_size_matches_oracle(batch):
    """train_naca.py:17-89 at the script's size (120-point outline -> 728 latent points cut out of the
    221x51 body-fitted grid -> 11 271 output points, hid 128, 1 head, 4 blocks, per-sample meshes):
    forward, RelL2 loss and every gradient against the oracle (was a finite-and-shape check).
    Batch 2 = 22 542 decoder rows builds its plan with plan_rows_reg (a wave per row); batch 3 = 33 813 rows is past the
    32 768-row switch and runs plan_rows_lane (a row per lane) - the kernel the batch-20 bench runs."""
    from position_induced_transformer_amd import ops, tasks, utils
    model, sample, meta = tasks.make_task("naca", seed=61)
    mesh_in, func_in, mesh_out, target = sample(batch)
    with ops.head_scale_route("host"):
        out = model(mesh_in, func_in, mesh_out)
        loss = utils.RelLpNorm(4, 2)(target, out)
        loss.backward()
    torch.cuda.synchronize()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    mo = mesh_out.cpu()
    ltt = mo[:, ::4, ::4, :][:, :56, :13, :].reshape(batch, -1, 2)              # train_naca.py:62-65
    ref = orc.pit_apply(p, "euclid", True, 4, 0.02, 0.02, mesh_in.cpu(), func_in.cpu(), ltt,
                        mo.reshape(batch, -1, 2)).reshape(batch, 221, 51, 4)
    ref_loss = orc.rel_lp_loss(target.cpu(), ref, 4, 2)
    ref_loss.backward()
    _compare_with_oracle(model, out, loss, ref, ref_loss, p)


def test_sod_and_elasticity_full_size_match_oracle():
    """train_sod.py:55-76 (1024 -> 256 -> 1024 on [-5,5), hid 32, 1 head, 2 blocks, 3 channels, RelL1) and
    train_elasticity.py:56-75 (972-point clouds, latent = output = input mesh, hid 256, 2 heads, en_layer
    88 -> 256 -> 256) at the scripts' sizes, batch 2."""
    from position_induced_transformer_amd import ops, tasks, utils
    # Sod
    model, sample, meta = tasks.make_task("sod", seed=62)
    mesh_in, func_in, mesh_out, target = sample(2)
    with ops.head_scale_route("host"):
        out = model(mesh_in, func_in, mesh_out)
        loss = utils.RelLpNorm(3, 1)(target, out)
        loss.backward()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    mi = mesh_in.cpu()
    ref = orc.pit_apply(p, "euclid", False, 2, 0.02, 0.02, mi, orc.with_coords(mi, func_in.cpu()), model.mesh_ltt.cpu(), mi)
    ref_loss = orc.rel_lp_loss(target.cpu(), ref, 3, 1)
    ref_loss.backward()
    _compare_with_oracle(model, out, loss, ref, ref_loss, p)
    # Elasticity
    model, sample, meta = tasks.make_task("elasticity", seed=63)
    mesh_in, func_in, mesh_out, target = sample(2)
    with ops.head_scale_route("host"):
        out = model(mesh_in, func_in, mesh_out)
        loss = utils.RelLpNorm(1, 2)(target, out)
        loss.backward()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    xy = mesh_in.cpu()
    ref = orc.pit_apply(p, "euclid", True, 4, 0.02, 0.02, xy, func_in.cpu(), xy, xy)
    ref_loss = orc.rel_lp_loss(target.cpu(), ref, 1, 2)
    ref_loss.backward()
    # hid 256 (K = 768 sums), random-init weights: the oracle's OWN fp32 result is 7.0e-6 from its fp64 evaluation
    # on this case, so two correct fp32 evaluations sit ~1e-5 apart (measured here 1.05e-5): bound = 2 x that distance
    _compare_with_oracle(model, out, loss, ref, ref_loss, p, tol_out=1.4e-5)


# --------------------------------------------------------------------------- concat buffer hand-off (MLP -> self-attention)
def test_mlp_output_in_concat_buffer_equals_the_copy_path_and_is_consumed_once():
    """ops.mlp_apply(..., concat_heads=H) writes straight into the next self-attention's concat buffer (no input copy
    in the attention epilogue, pit.py:44).  Same numbers as the plain path, forward and backward; a SECOND consumer of
    the same MLP output must not reuse (and overwrite) the buffer the first one returned."""
    from position_induced_transformer_amd import ops, pit as P
    torch.manual_seed(5)
    mlp = P.kaiming_mlp(24, 64, 64).cuda()
    att1, att2 = P.posatt_fixed(2, 64, 1.0).cuda(), P.posatt_fixed(2, 64, 1.0).cuda()
    mesh = torch.rand(200, 2, device="cuda")
    x = torch.randn(3, 200, 24, device="cuda")

    def run(concat_heads):
        for m in (mlp, att1, att2):
            m.zero_grad()
        xx = x.clone().requires_grad_(True)
        y = mlp(xx, out_gelu=True, concat_heads=concat_heads)
        a = att1(mesh, y)                    # consumes the buffer (if any)
        b = att2(mesh, y)                    # second consumer: must take the copy path
        (a.square().sum() + (b * 0.5).sum()).backward()
        return a.detach().clone(), b.detach().clone(), xx.grad.clone(), [p.grad.clone() for p in mlp.parameters()]

    ref = run(0)
    got = run(2)
    assert hasattr(mlp(x, out_gelu=True, concat_heads=2), "_pit_concat")
    for r, g in zip(ref[:3], got[:3]):
        assert gio.rel_l2(r.cpu().numpy(), g.cpu().numpy()) <= 1e-6
    for r, g in zip(ref[3], got[3]):
        assert gio.rel_l2(r.cpu().numpy(), g.cpu().numpy()) <= 1e-5
    # the first consumer's output really is the buffer (shares the MLP output's storage), the second's is not
    y = mlp(x, out_gelu=True, concat_heads=2)
    a = att1(mesh, y)
    b = att2(mesh, y)
    assert a.data_ptr() == y.data_ptr() and b.data_ptr() != y.data_ptr()
    assert torch.equal(a[..., :64], y) and torch.equal(b[..., :64], y)


# --------------------------------------------------------------------------- coordinate concat fused into the encoder kernel
@pytest.mark.parametrize("task", ["darcy", "burgers", "sod"])
def test_fused_coordinate_channels_equal_the_materialised_concat(task):
    """The task forwards' `func_in = cat((tile(mesh_in), func_in), -1)` (train_darcy.py:51-55, train_burgers.py:44,
    train_sod.py:49) is not materialised: the encoder's candidate-list kernels take the coordinate channels from
    mesh_in (pit_posatt_fwd/bwd coord_dims).  Same output, parameter gradients and d(func_in) as the explicit concat
    through the same modules."""
    from position_induced_transformer_amd import ops, tasks, utils
    model, sample, meta = tasks.make_task(task, seed=71)
    mesh_in, func_in, mesh_out, target = sample(3)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])

    def run(fused):
        model.zero_grad()
        x = func_in.clone().requires_grad_(True)
        if fused:
            out = model(mesh_in, x, mesh_out)
        else:
            mi = mesh_in.reshape(-1, model.space_dim)
            func = x.reshape(3, -1, model.in_dim)
            feats = torch.cat((mi.unsqueeze(0).expand(3, -1, -1), func), -1)          # the reference's own line
            ltt = model.encoder(mi, feats, model.mesh_ltt)
            ltt = model.processor(ltt, model.mesh_ltt)
            out = model.decoder(model.mesh_ltt, ltt, mesh_out.reshape(-1, model.space_dim)).reshape(target.shape)
        loss = loss_fn(target, out)
        loss.backward()
        return out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in model.named_parameters()}

    launched = []
    orig = ops.posatt_apply

    def spy(*a, **kw):
        launched.append(kw.get("coord_dims", 0))
        return orig(*a, **kw)
    ops.posatt_apply = spy
    try:
        got = run(True)
    finally:
        ops.posatt_apply = orig
    assert model.space_dim in launched, "the fused path was not taken"
    ref = run(False)
    assert gio.rel_l2(ref[0].cpu().numpy(), got[0].cpu().numpy()) <= 1e-6
    assert gio.rel_l2(ref[1].cpu().numpy(), got[1].cpu().numpy()) <= 1e-5
    he, hg = [], []
    for k in ref[2]:
        if k.endswith("lmda"):
            he.append(ref[2][k].cpu().numpy().reshape(-1)); hg.append(got[2][k].cpu().numpy().reshape(-1))
        else:
            assert gio.rel_l2(ref[2][k].cpu().numpy(), got[2][k].cpu().numpy()) <= 1e-5, k
    assert gio.rel_l2(np.concatenate(he), np.concatenate(hg)) <= 1e-4


# --------------------------------------------------------------------------- self-cleaning state after many replays
@pytest.mark.parametrize("task,batch", [("darcy", 8), ("naca", 2)])
def test_replays_leave_every_self_cleaning_buffer_clean(task, batch):
    """The soak invariants after 400 replays of the captured training step (fwd + loss + bwd + fused
    Adam): the Adam step counter counts exactly, its arrival ticket is back at zero, the gradients it consumed are
    cleared, the fp64 d(scale) accumulators and the loss workspace (partial sums, per-pair and global tickets) are
    zero, parameters and loss finite."""
    from position_induced_transformer_amd import ops, tasks
    from position_induced_transformer_amd.ddp import FlatAdam, FlatGradients
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task(task, seed=3)
    flat = FlatGradients(model.parameters(), flatten_params=True)
    opt = FlatAdam(flat, lr=1e-4, cosine_t_max=500, zero_grads=True)
    step = TrainStep(model, sample(batch), meta["out_dim"], meta["p"], optimizer=opt, flat=flat)
    step.capture()
    torch.cuda.synchronize()
    base = int(opt.step_count)
    for _ in range(400):
        step.replay()
    torch.cuda.synchronize()
    assert int(opt.step_count) == base + 400
    assert int(opt.scalars[3].view(torch.int32)) == 0, "Adam arrival ticket not reset"
    assert float(flat.flat.abs().max()) == 0.0, "gradients not cleared by the fused Adam"
    assert any(True for _ in ops._LAYER_WS), "no deferred d(lmda) accumulators were used"
    for ws in ops._LAYER_WS.values():
        assert float(ws.abs().max()) == 0.0, "d(scale) accumulators not drained"
    for ws in ops._LOSS_WS.values():
        assert float(ws.abs().max()) == 0.0, "loss workspace not reset"
    assert torch.isfinite(flat.flat_params).all() and torch.isfinite(step.loss)


# --------------------------------------------------------------------------- weight gradients riding with the attention backward
@pytest.mark.parametrize("task,batch,math", [("darcy", 8, "fp32"), ("darcy", 8, "bf16"), ("burgers", 8, "fp32"),
                                             ("elasticity", 2, "fp32"), ("vorticity", 2, "fp32")])
def test_mlp_weight_gradients_carried_by_the_attention_backward_equal_their_own_launch(task, batch, math):
    """pit_hip.h `rider`: the postponed pit_mlp_bwd_params of a block's MLP performed by the following
    pit_posatt_bwd (inside its launch when both are small) against the same pass with the MLP backward issuing
    its own reductions.  Everything but the atomics' summation order is identical."""
    from position_induced_transformer_amd import ops, tasks, utils
    from position_induced_transformer_amd.ddp import FlatGradients
    model, sample, meta = tasks.make_task(task, seed=11)
    batch_t = sample(batch)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])
    flat = FlatGradients(model.parameters())
    got = {}
    ops._PENDING_DW.clear()                      # (entries of passes that earlier tests aborted on purpose: other models' slots)
    with ops.math_mode(math):
        for rider in (True, False, True):
            ops.MLP_PARAMS_RIDER = rider
            try:
                flat.zero_()
                loss_fn(batch_t[-1], model(*batch_t[:-1])).backward()
                torch.cuda.synchronize()
            finally:
                ops.MLP_PARAMS_RIDER = True
            assert not ops._PENDING_DW
            got.setdefault(rider, []).append(flat.flat.clone())
    assert float(got[True][0].abs().max()) > 0
    assert gio.rel_l2(got[False][0].cpu().numpy(), got[True][0].cpu().numpy()) <= 2e-6
    assert gio.rel_l2(got[True][0].cpu().numpy(), got[True][1].cpu().numpy()) <= 2e-6
    for (k, p), v in zip(model.named_parameters(), flat._views):        # parameter by parameter, not only in norm
        a = got[False][0][v.storage_offset():v.storage_offset() + v.numel()].cpu().numpy()
        b = got[True][0][v.storage_offset():v.storage_offset() + v.numel()].cpu().numpy()
        c = got[True][1][v.storage_offset():v.storage_offset() + v.numel()].cpu().numpy()
        # (the union-tile attention backward sums d(values) with atomics: two identical passes differ by `noise`,
        #  largest on the bias sums, which cancel heavily)
        noise = gio.rel_l2(c, b)
        assert gio.rel_l2(a, b) <= max(1e-5, 4 * noise), (k, noise)
        assert noise <= 1e-4, (k, noise)


def test_a_postponed_weight_gradient_job_of_an_aborted_pass_is_dropped():
    """A pass that raises between an MLP backward (job postponed) and the attention backward that would have
    carried it leaves the job behind: the next pass must neither run it (its buffers are gone) nor lose its own."""
    from position_induced_transformer_amd import ops, tasks, utils
    from position_induced_transformer_amd.ddp import FlatGradients
    model, sample, meta = tasks.make_task("darcy", seed=4)
    mesh_in, func_in, mesh_out, target = sample(8)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])
    flat = FlatGradients(model.parameters())
    loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    torch.cuda.synchronize()
    want = flat.flat.clone()

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    # die right after the last block's MLP backward postponed its reductions (before its attention backward)
    last = model.mlp[-1]
    orig = last.forward
    last.forward = lambda x, *a, **k: orig(Boom.apply(x), *a, **k)
    try:
        with pytest.raises(RuntimeError, match="boom"):
            loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    finally:
        last.forward = orig
    assert any(j is not None for j in ops._PENDING_DW.values()), "the scenario did not leave a postponed job"
    flat.zero_()
    loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
    torch.cuda.synchronize()
    assert not ops._PENDING_DW
    assert gio.rel_l2(want.cpu().numpy(), flat.flat.cpu().numpy()) <= 2e-6


@pytest.mark.parametrize("n,hid,batch,locality", [(256, 64, 8, 1.0), (256, 64, 8, 0.3), (972, 64, 2, 1.0), (300, 128, 1, 1.0)])
def test_cabi_posatt_bwd_with_a_rider_equals_the_two_separate_calls(n, hid, batch, locality):
    """include/pit_hip.h, `rider`: one pit_posatt_bwd call carrying a pit_mlp_params_job must leave the same
    d_values / d_head as the plain call and the same d_w1, d_b1, d_w2, d_b2 as a separate pit_mlp_bwd_params -
    whether it merges the reductions into its launch (the small cases) or falls back to separate launches."""
    import ctypes
    from position_induced_transformer_amd import _lib, ops
    torch.manual_seed(5)
    H, d = 2, hid
    mesh = torch.rand(n, 2, device="cuda")
    plan = ops.MeshPlan("euclid", mesh, mesh, locality, True)
    u = torch.randn(batch, n, d, device="cuda", requires_grad=True)
    lm = torch.rand(H, device="cuda", requires_grad=True)
    out = ops.posatt_apply(u, lm, plan, H, True)
    values, head, rowstat, scale = out.grad_fn.saved_tensors
    d_out = torch.randn_like(out)
    rows, n0, n1, n2 = batch * n, (1 + H) * d, d, d
    x2, hh = torch.randn(rows, n0, device="cuda"), torch.randn(rows, n1, device="cuda")
    scratch = torch.randn(rows * (n1 + n2), device="cuda")
    L = _lib.lib()

    def run(with_rider):
        d_values = torch.empty_like(u)
        d_head = torch.zeros(H, device="cuda")
        work = torch.zeros(H * 1024, device="cuda", dtype=torch.float64)
        g = [torch.zeros(n1, n0, device="cuda"), torch.zeros(n1, device="cuda"),
             torch.zeros(n2, n1, device="cuda"), torch.zeros(n2, device="cuda")]
        job = _lib.MlpParamsJob(x2.data_ptr(), n0, rows, n0, n1, n2, hh.data_ptr(), 1, d_out.data_ptr(), d_out.stride(1),
                                g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), 1, scratch.data_ptr(), 0)
        rc = L.pit_posatt_bwd(
            plan.mesh_out.data_ptr(), plan.mesh_in.data_ptr(), plan.mesh_batch, plan.n_out, plan.n_in,
            plan.sdim, plan.metric_id, plan.period,
            values.data_ptr(), batch, d, values.stride(1), values.stride(0),
            head.data_ptr(), H, 0, scale.data_ptr(), rowstat.data_ptr(), 1 if plan.masked else 0,
            d_out.data_ptr(), d_out.stride(1), d_out.stride(0), d,
            d_values.data_ptr(), d_values.stride(1), d_values.stride(0), 1,
            d_head.data_ptr(), 0, work.data_ptr(),
            _lib.ptr(plan.nbr_idx), _lib.ptr(plan.nbr_cnt), plan.nbr_cap, plan.lists_complete(),
            _lib.ptr(plan.rev_ptr), _lib.ptr(plan.rev_row),
            ctypes.cast(ctypes.pointer(job), ctypes.c_void_p) if with_rider else None, 0, 0, _lib.stream_ptr())
        _lib.check(rc, "pit_posatt_bwd")
        if not with_rider:
            rc = L.pit_mlp_bwd_params(x2.data_ptr(), n0, rows, n0, n1, n2, hh.data_ptr(), 1, d_out.data_ptr(), d_out.stride(1),
                                      g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), 1, scratch.data_ptr(),
                                      0, _lib.stream_ptr())
            _lib.check(rc, "pit_mlp_bwd_params")
        torch.cuda.synchronize()
        assert float(work.abs().max()) == 0.0, "d(scale) accumulators not drained"
        return [d_values, d_head] + g

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]), "d_values must not depend on the rider"
    assert gio.rel_l2(b[1].cpu().numpy(), a[1].cpu().numpy()) <= 1e-6
    for x, y, name in zip(a[2:], b[2:], ("d_w1", "d_b1", "d_w2", "d_b2")):
        assert float(y.abs().max()) > 0, name
        assert gio.rel_l2(y.cpu().numpy(), x.cpu().numpy()) <= 2e-6, name      # (fp32 atomics: summation order)
