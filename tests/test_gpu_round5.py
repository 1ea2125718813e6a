"""GPU: round 5 - the fused encoder-side and decoder-side launches of the small regime (csrc/pit_edge.hip) against the
oracle, the static slab plans they run on, and the RelLp loss with non-finite terms."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_io as gio
import pit_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------- RelLp loss: non-finite terms
@pytest.mark.parametrize("npts,out_dim", [(1849, 1), (300, 3)])
def test_rel_lp_loss_reports_non_finite_terms_and_recovers(npts, out_dim):
    """utils.py:80-98 with an all-zero target series (||y|| = 0 -> x/0) and with a NaN prediction: the reference's LpLoss
    reports inf / nan; the single-workgroup loss kernel must too (round 4 packed the arrival count and a fixed-point sum into
    one word: inf became a finite wrong loss and could leave the accumulator armed) - and the NEXT call on the same workspace
    must be exact again."""
    from position_induced_transformer_amd import utils
    g = torch.Generator().manual_seed(5)
    b = 4
    t, q = torch.randn(b, npts, out_dim, generator=g), torch.randn(b, npts, out_dim, generator=g)
    loss = utils.RelLpNorm(out_dim, 2)
    want = float(orc.rel_lp_loss(t, q, out_dim, 2))
    good = float(loss(t.cuda(), q.cuda()))
    assert abs(good - want) <= 1e-6 * abs(want)
    t0 = t.clone()
    t0[1] = 0.0                                          # ||y_1|| = 0: the reference divides by zero -> inf
    ref0 = float(orc.rel_lp_loss(t0, q, out_dim, 2))
    got0 = float(loss(t0.cuda(), q.cuda()))
    assert not np.isfinite(ref0) and not np.isfinite(got0)
    assert abs(float(loss(t.cuda(), q.cuda())) - want) <= 1e-6 * abs(want)     # the workspace was left clean
    qn = q.clone()
    qn[2, 7, 0] = float("nan")
    assert np.isnan(float(orc.rel_lp_loss(t, qn, out_dim, 2))) and np.isnan(float(loss(t.cuda(), qn.cuda())))
    assert abs(float(loss(t.cuda(), q.cuda())) - want) <= 1e-6 * abs(want)
    big = q * 1e30                                       # a diverged step: the squared norm overflows fp32 but not the loss's fp64
    ref_big, got_big = float(orc.rel_lp_loss(t, big, out_dim, 2)), float(loss(t.cuda(), big.cuda()))
    assert (np.isinf(ref_big) and np.isinf(got_big)) or abs(got_big - ref_big) <= 1e-5 * abs(ref_big)
    assert abs(float(loss(t.cuda(), q.cuda())) - want) <= 1e-6 * abs(want)


# --------------------------------------------------------------------------- fused encoder- / decoder-side launches
METRIC = {"darcy": "euclid", "burgers": "periodic1d", "sod": "euclid"}
EDGE_CASES = [("darcy", 8), ("darcy", 3), ("darcy", 1), ("burgers", 8), ("sod", 8)]


class _Count:
    """Counts calls of ops.<name> (the tests must exercise the fused launches, not a fallback)."""

    def __init__(self, name):
        from position_induced_transformer_amd import ops
        self.ops, self.name, self.n = ops, name, 0

    def __enter__(self):
        self.orig = getattr(self.ops, self.name)

        def counting(*a, **k):
            self.n += 1
            return self.orig(*a, **k)
        setattr(self.ops, self.name, counting)
        return self

    def __exit__(self, *exc):
        setattr(self.ops, self.name, self.orig)
        return False


def _cpu_params(*tensors):
    return [t.detach().cpu().clone().requires_grad_(True) for t in tensors]


def test_slab_plan_records_distances_unions_and_slots():
    """pit_slab_plan_build on the Darcy decoder pair (43 x 43 output grid <- 16 x 16 latent) and a random cloud: the candidates'
    distances are the reference's fp32 expression bit for bit (pit.py:134), a slab's keys are the sorted union of its rows' lists,
    every candidate's slot points at its key, the reported maximum is the largest union."""
    from position_induced_transformer_amd import ops
    g = torch.Generator().manual_seed(3)
    pairs = [(orc.grid_mesh_2d(43), orc.grid_mesh_2d(16), 0.02), (torch.rand(700, 2, generator=g), torch.rand(300, 2, generator=g), 0.05)]
    for mo, mi, loc in pairs:
        plan = ops.MeshPlan("euclid", mo.cuda(), mi.cuda(), loc, False)
        sp, max_union, (m, slot, keys, nkeys), max_count = plan.slab_plan()
        assert max_count == int(plan.nbr_cnt.clamp(max=plan.nbr_cap).max())
        cap, n_out = plan.nbr_cap, plan.n_out
        idx, cnt = plan.nbr_idx.view(n_out, cap).cpu().long(), plan.nbr_cnt.cpu().long()
        ref_m = ((mo[:, None, :] - mi[None, :, :]) ** 2).sum(-1)
        m, slot, keys, nkeys = m.cpu(), slot.cpu().long() & 0xffff, keys.cpu().long(), nkeys.cpu().long()
        assert int(nkeys.max()) == max_union
        for s in range(sp.n_slabs):
            rows = range(16 * s, min(16 * s + 16, n_out))
            union = sorted({int(idx[r, i]) for r in rows for i in range(int(cnt[r]))})
            assert int(nkeys[s]) == len(union)
            if len(union) <= 64:
                assert keys[s, :len(union)].tolist() == union
            for r in rows:
                c = int(cnt[r])
                assert torch.equal(m[r, :c], ref_m[r, idx[r, :c]])
                if len(union) <= 64:
                    assert keys[s, slot[r, :c]].tolist() == idx[r, :c].tolist()


@pytest.mark.parametrize("task,batch", EDGE_CASES)
def test_fused_decoder_against_the_oracle(task, batch):
    """pit.decoder (pit.py:124-127) through ops.decoder_apply - ONE launch forward, ONE backward - against the oracle's
    posatt_cross + mlp on the same parameters and inputs: prediction <= 1e-5, d(values) and weight gradients <= 2e-5, d(lmda) <= 2e-4.
    Route 'host': the head scale is the reference's own torch-CPU value, so the check holds for any seed."""
    from position_induced_transformer_amd import ops, tasks
    model, sample, meta = tasks.make_task(task, seed=51)
    mesh_in, func_in, mesh_out, target = sample(batch)
    mo = mesh_out.reshape(-1, model.space_dim)
    L, hid = model.mesh_ltt.shape[0], model.hid_dim
    g = torch.Generator().manual_seed(52)
    x = torch.randn(batch, L, hid, generator=g)
    d_out = torch.randn(batch, mo.shape[0], model.out_dim, generator=g)
    xg = x.cuda().requires_grad_(True)
    with _Count("decoder_apply") as cnt, ops.head_scale_route("host"):
        out = model.decoder(model.mesh_ltt, xg, mo)
        out.backward(d_out.cuda())
    torch.cuda.synchronize()
    assert cnt.n == 1, "the fused decoder launch did not run"
    de = model.de
    xr, lm, w1, b1, w2, b2 = _cpu_params(x, model.up.lmda, de.mlp1.weight, de.mlp1.bias, de.mlp2.weight, de.mlp2.bias)
    ref = orc.mlp(orc.posatt_cross(METRIC[task], False, mo.cpu(), model.mesh_ltt.cpu(), xr, lm, 0.02), w1, b1, w2, b2)
    ref.backward(d_out)
    assert gio.rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    assert gio.rel_l2(xg.grad.cpu().numpy(), xr.grad.numpy()) <= 2e-5
    for name, a, r in (("w1", de.mlp1.weight, w1), ("b1", de.mlp1.bias, b1), ("w2", de.mlp2.weight, w2), ("b2", de.mlp2.bias, b2)):
        assert gio.rel_l2(a.grad.cpu().numpy(), r.grad.numpy()) <= 2e-5, name
    assert float((model.up.lmda.grad.cpu().reshape(-1) - lm.grad.reshape(-1)).norm()) <= 2e-4 * float(lm.grad.norm()), "d(lmda)"
    # forward only (no_grad: nothing is saved) gives the same prediction
    with torch.no_grad(), ops.head_scale_route("host"):
        again = model.decoder(model.mesh_ltt, x.cuda(), mo)
    assert torch.equal(again, out.detach())


@pytest.mark.parametrize("tagged", [True, False], ids=["coordinates-from-the-mesh", "materialised-concat"])
@pytest.mark.parametrize("task,batch", EDGE_CASES)
def test_fused_encoder_against_the_oracle(task, batch, tagged):
    """pit.encoder (pit.py:108-112) through ops.encoder_apply against the oracle's gelu(mlp(posatt_cross(...))), with the
    coordinate channels read from the mesh (ops.tag_coords, train_darcy.py:51-55) and with the concat materialised."""
    from position_induced_transformer_amd import ops, tasks
    model, sample, meta = tasks.make_task(task, seed=53)
    mesh_in, func_in, mesh_out, target = sample(batch)
    mi = mesh_in.reshape(-1, model.space_dim)
    func = func_in.reshape(batch, -1, model.in_dim)
    L, hid = model.mesh_ltt.shape[0], model.hid_dim
    d_out = torch.randn(batch, L, hid, generator=torch.Generator().manual_seed(54))
    feats = ops.tag_coords(func, mi) if tagged else torch.cat((mi.unsqueeze(0).expand(batch, -1, -1), func), -1)
    with _Count("encoder_apply") as cnt, ops.head_scale_route("host"):
        out = model.encoder(mi, feats, model.mesh_ltt)
        torch.autograd.backward(out, d_out.cuda())
    torch.cuda.synchronize()
    assert cnt.n == 1, "the fused encoder launch did not run"
    en = model.en_layer
    lm, w1, b1, w2, b2 = _cpu_params(model.down.lmda, en.mlp1.weight, en.mlp1.bias, en.mlp2.weight, en.mlp2.bias)
    f = orc.posatt_cross(METRIC[task], False, model.mesh_ltt.cpu(), mi.cpu(), orc.with_coords(mi.cpu(), func.cpu()), lm, 0.02)
    ref = F.gelu(orc.mlp(f, w1, b1, w2, b2))
    ref.backward(d_out)
    assert gio.rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    for name, a, r in (("w1", en.mlp1.weight, w1), ("b1", en.mlp1.bias, b1), ("w2", en.mlp2.weight, w2), ("b2", en.mlp2.bias, b2)):
        assert gio.rel_l2(a.grad.cpu().numpy(), r.grad.numpy()) <= 2e-5, name
    assert float((model.down.lmda.grad.cpu().reshape(-1) - lm.grad.reshape(-1)).norm()) <= 2e-4 * float(lm.grad.norm()), "d(lmda)"


def _oracle_step(task, model, batch, affine, meta):
    """Forward + RelLp loss + backward of the whole model with the oracle; returns (prediction, loss, gradients by name)."""
    mesh_in, func_in, mesh_out, target = batch
    b = func_in.shape[0]
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    mi, mo = mesh_in.cpu().reshape(-1, model.space_dim), mesh_out.cpu().reshape(-1, model.space_dim)
    f = orc.with_coords(mi, func_in.cpu().reshape(b, -1, model.in_dim))
    ref = orc.pit_apply(sd, METRIC[task], False, model.n_blocks, model.en_local, model.de_local, mi, f, model.mesh_ltt.cpu(), mo)
    ref = ref.reshape(target.shape)
    pred = ref if affine is None else ref * affine[0].cpu() + affine[1].cpu()
    loss = orc.rel_lp_loss(target.cpu(), pred, meta["out_dim"], meta["p"])
    loss.backward()
    return ref.detach(), float(loss), {k: v.grad for k, v in sd.items()}


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
@pytest.mark.parametrize("task,batch,affine", [("darcy", 8, True), ("darcy", 3, False), ("burgers", 8, False), ("sod", 8, False)])
def test_train_step_with_the_loss_inside_the_decoder_launches_matches_the_oracle(task, batch, affine, graph):
    """engine.TrainStep on a model whose encoder and decoder sides are fused launches: the RelLp loss (p = 2 with the Darcy loop's
    affine de-normalisation, train_darcy.py:129-130; p = 1: Burgers, Sod with three channels) is accumulated by the decoder's
    forward launch and differentiated by its backward launch - no loss launch - and prediction, loss and every gradient of the
    flat buffer equal the oracle's forward + loss + backward."""
    from position_induced_transformer_amd import ops, tasks
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task(task, seed=55)
    b4 = sample(batch)
    aff = None
    if affine:
        g = torch.Generator().manual_seed(56)
        aff = (torch.rand(b4[3].shape[1:], generator=g).cuda() + 0.5, torch.randn(b4[3].shape[1:], generator=g).cuda())
    with ops.head_scale_route("host"), _Count("decoder_apply") as cd, _Count("encoder_apply") as ce:
        step = TrainStep(model, b4, meta["out_dim"], meta["p"], pred_affine=aff)
        fused = {"n": 0}
        orig = ops._FusedLoss.apply

        def counting(*a, **k):
            fused["n"] += 1
            return orig(*a, **k)
        ops._FusedLoss.apply = counting
        try:
            if graph:
                step.capture()
                for _ in range(3):
                    step.replay()
            else:
                step.run_eager()
                step.run_eager()
        finally:
            ops._FusedLoss.apply = orig
    torch.cuda.synchronize()
    assert cd.n >= 1 and ce.n >= 1 and fused["n"] >= 1, "the fused launches / the loss inside them did not run"
    ref, ref_loss, ref_grads = _oracle_step(task, model, b4, aff, meta)
    assert gio.rel_l2(step.out.cpu().numpy(), ref.numpy()) <= 1e-5
    assert abs(float(step.loss) - ref_loss) <= 1e-5 * abs(ref_loss)
    lk = [k for k in ref_grads if k.endswith("lmda")]
    for k, q in model.named_parameters():
        if not k.endswith("lmda"):
            assert gio.rel_l2(q.grad.cpu().numpy(), ref_grads[k].numpy()) <= 2e-5, k
    got = torch.cat([dict(model.named_parameters())[k].grad.cpu().reshape(-1) for k in lk])
    want = torch.cat([ref_grads[k].reshape(-1) for k in lk])
    assert float((got - want).norm()) <= 2e-4 * float(want.norm()), "d(lmda)"


def test_prediction_that_is_post_processed_keeps_the_loss_launch():
    """The loss rides in the decoder launches only when the model's prediction reaches it through views; Cylinder adds its input
    back onto the prediction (train_cylinder.py:52): the step must fall back to the loss kernel - and still be right."""
    from position_induced_transformer_amd import ops, tasks
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task("darcy", seed=57)
    model.residual = True
    b4 = sample(4)
    step = TrainStep(model, b4, meta["out_dim"], meta["p"])
    with ops.head_scale_route("host"):
        step.run_eager()
        step.run_eager()
    torch.cuda.synchronize()
    model.residual = False
    t, o = b4[3].cpu().reshape(4, -1), step.out.cpu().reshape(4, -1)
    want = float(((t - o).norm(dim=1) / t.norm(dim=1)).sum())
    assert abs(float(step.loss) - want) <= 1e-5 * abs(want)
    assert all(torch.isfinite(q.grad).all() and float(q.grad.abs().max()) > 0 for q in model.parameters())


@pytest.mark.parametrize("decoder", ["folded", "fused"])
def test_large_batch_step_on_the_large_regime_kernels_matches_the_oracle(decoder):
    """Darcy at batch 256 (the large-regime paths had no direct test): 65 536 latent rows take the precomputed-weights
    self-attention (pit_posatt_pre_fwd / _bwd) and the 64-row-slab MLP kernels (pit_mlp_slab.hip); the 473 344 decoder rows the
    folded decoder (round 6: from ops.FOLD_EDGE_ROWS rows) or - the threshold moved away - the fused decoder launches with the
    loss inside: forward, loss and every gradient of the flat buffer against the oracle.  (The weight-gradient tolerance grows
    with the square root of the rows summed, as in bench.parity_vs_oracle.)"""
    from position_induced_transformer_amd import ops, tasks
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task("darcy", seed=58)
    b4 = sample(256)
    calls = {"pre": 0}
    orig = ops.posatt_pre_apply

    def counting(*a, **k):
        calls["pre"] += 1
        return orig(*a, **k)
    ops.posatt_pre_apply = counting
    saved = ops.FOLD_EDGE_ROWS
    if decoder == "fused":
        ops.FOLD_EDGE_ROWS = 1 << 62
    try:
        with ops.head_scale_route("host"), _Count("decoder_apply") as cd, _Count("fold_decoder_apply") as cf:
            step = TrainStep(model, b4, meta["out_dim"], meta["p"])
            step.run_eager()
            step.run_eager()
    finally:
        ops.posatt_pre_apply = orig
        ops.FOLD_EDGE_ROWS = saved
    torch.cuda.synchronize()
    assert calls["pre"] >= len(model.conv)
    assert (cf.n >= 1 and cd.n == 0) if decoder == "folded" else (cd.n >= 1 and cf.n == 0), (cd.n, cf.n)
    ref, ref_loss, ref_grads = _oracle_step("darcy", model, b4, None, meta)
    assert gio.rel_l2(step.out.cpu().numpy(), ref.numpy()) <= 1e-5
    assert abs(float(step.loss) - ref_loss) <= 1e-5 * abs(ref_loss)
    wtol = 2e-5 * (256 * 1849 / 16384.0) ** 0.5
    lk = [k for k in ref_grads if k.endswith("lmda")]
    for k, q in model.named_parameters():
        if not k.endswith("lmda"):
            assert gio.rel_l2(q.grad.cpu().numpy(), ref_grads[k].numpy()) <= wtol, k
    got = torch.cat([dict(model.named_parameters())[k].grad.cpu().reshape(-1) for k in lk])
    want = torch.cat([ref_grads[k].reshape(-1) for k in lk])
    assert float((got - want).norm()) <= 5e-4 * float(want.norm()), "d(lmda)"


def test_two_bucket_decision_is_made_once_and_a_later_change_raises():
    """whether a data-parallel step reduces an early bucket is the SHAPE of its collective sequence - decided at the
    first step (by all ranks together: MIN over the group) and never silently switched: a tail gradient that stops being written
    in place afterwards raises."""
    from position_induced_transformer_amd import tasks
    from position_induced_transformer_amd.engine import TrainStep
    model, sample, meta = tasks.make_task("darcy", seed=59)
    step = TrainStep(model, sample(4), meta["out_dim"], meta["p"], all_reduce=True, all_reduce_buckets=2)
    step.flat.all_reduce = lambda average=False, group=None, part="all": None          # (one process: the sum over one rank)
    step.run_eager()
    assert step._early_decision is True
    handle = model.de.mlp1.weight.register_hook(lambda g: g)
    try:
        with pytest.raises(RuntimeError, match="no longer written in place"):
            step.run_eager()
    finally:
        handle.remove()
        step._early_hook.remove()
    torch.cuda.synchronize()


@pytest.mark.parametrize("metric,n_out_side,n_in_side,dim,heads,batch,loc", [
    ("periodic2d", 64, 16, 256, 2, 2, 0.02),          # Vorticity's up-projection (train_vorticity.py:98-106)
    ("euclid", 43, 16, 128, 1, 3, 0.02),
    ("euclid", 30, 12, 192, 2, 2, 0.05),
])
def test_union_attention_of_any_width_against_the_oracle(metric, n_out_side, n_in_side, dim, heads, batch, loc):
    """Masked cross attention on a batch-free mesh pair at widths 128 / 192 / 256 (pit_union_att_fwd / _bwd: the decoder's
    union-tile contraction without the MLP, a workgroup per 64-column chunk) against the oracle's posatt_cross: output <= 1e-6,
    d(values) <= 1e-5, d(lmda) <= 1e-4; and with bf16-stored out / d_out (PIT_IO_*) within the bf16 tolerances."""
    from position_induced_transformer_amd import ops
    per = metric != "euclid"
    mo, mi = orc.grid_mesh_2d(n_out_side, not per).reshape(-1, 2), orc.grid_mesh_2d(n_in_side, not per).reshape(-1, 2)
    g = torch.Generator().manual_seed(61)
    values = torch.randn(batch, mi.shape[0], dim, generator=g)
    lmda = torch.rand(heads, 1, 1, generator=g)
    d_out = torch.randn(batch, mo.shape[0], heads * dim, generator=g)
    v0, l0 = values.clone().requires_grad_(True), lmda.clone().requires_grad_(True)
    ref = orc.posatt_cross(metric, False, mo, mi, v0, l0, loc)
    ref.backward(d_out)
    plan = ops.MeshPlan(metric, mo.cuda(), mi.cuda(), loc, False)
    calls = {"n": 0}
    orig = ops._launch_decoder_weights

    def counting(w):
        calls["n"] += 1
        return orig(w)
    ops._launch_decoder_weights = counting
    try:
        with ops.head_scale_route("host"):
            v1, l1 = values.cuda().requires_grad_(True), lmda.cuda().requires_grad_(True)
            out = ops.posatt_apply(v1, l1, plan, heads, concat=False)
            out.backward(d_out.cuda())
    finally:
        ops._launch_decoder_weights = orig
    torch.cuda.synchronize()
    assert calls["n"] == 1, "the union-tile attention did not run"
    assert gio.rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-6
    assert gio.rel_l2(v1.grad.cpu().numpy(), v0.grad.numpy()) <= 1e-5
    assert float((l1.grad.cpu().reshape(-1) - l0.grad.reshape(-1)).norm()) <= 1e-4 * float(l0.grad.norm())
    # bf16-stored output and gradient (bf16 math mode): storage rounding only
    with ops.math_mode("bf16"), ops.head_scale_route("host"):
        v2, l2 = values.cuda().requires_grad_(True), lmda.cuda().requires_grad_(True)
        out16 = ops.posatt_apply(v2, l2, plan, heads, concat=False, out_bf16=True)
        assert out16.dtype == torch.bfloat16
        out16.backward(d_out.cuda().to(torch.bfloat16))
    torch.cuda.synchronize()
    assert gio.rel_l2(out16.detach().float().cpu().numpy(), ref.detach().numpy()) <= 1e-2
    assert gio.rel_l2(v2.grad.cpu().numpy(), v0.grad.numpy()) <= 1e-2
    assert float((l2.grad.cpu().reshape(-1) - l0.grad.reshape(-1)).norm()) <= 5e-2 * float(l0.grad.norm())


# --------------------------------------------------------------------------- determinism flag
@pytest.mark.gpu
def test_deterministic_algorithms_flag_keeps_the_atomic_free_data_path():
    """d(values) of the fused decoder launch and of the union attention leave as fp32 atomic adds; under
    torch.use_deterministic_algorithms both layers keep the candidate-list kernels (transposed lists, fixed summation order):
    the gradient of the input function is bit-identical from pass to pass."""
    import torch
    from position_induced_transformer_amd import ops, tasks, utils
    model, sample, meta = tasks.make_task("darcy", seed=5)
    mesh_in, func_in, mesh_out, target = sample(4)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])
    plan = model.up._plan(mesh_out.reshape(-1, 2), model.mesh_ltt, False)
    assert ops.edge_fusion_supported(plan, model.up.n_head, model.hid_dim, 4, True)
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        assert not ops.edge_fusion_supported(plan, model.up.n_head, model.hid_dim, 4, True)
        assert not ops._union_att_ok(plan, 2, 256, 4, torch.empty(4, plan.n_in, 256, device="cuda"))
        grads = []
        for _ in range(3):
            f = func_in.clone().requires_grad_(True)
            model.zero_grad(set_to_none=True)
            loss_fn(target, model(mesh_in, f, mesh_out)).backward()
            grads.append(f.grad.clone())
        assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
        assert float(grads[0].abs().max()) > 0
    finally:
        torch.use_deterministic_algorithms(False)


# --------------------------------------------------------------------------- operands as unit-stride copies
@pytest.mark.gpu
def test_weights_and_e_rows_through_lds_give_the_bits_of_the_fragment_loads():
    """block_fwd_kernel<H, WLDS> (the MLP's weights and the slab's E rows reach the MFMAs through LDS when the launch has at most 256
    workgroups) against the same kernel with per-lane fragment loads (PIT_NO_BLOCK_WLDS, read once per process: a child each),
    and the decoder forward on the step's fragment-order copy of W1 against the row-major W1: the same operands in the same
    order - the prediction is bit-identical."""
    import hashlib, os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, hashlib; sys.path.insert(0, %r); import torch\n"
        "from position_induced_transformer_amd import ops, tasks\n"
        "if 'NOW1F' in sys.argv:\n"
        "    real = ops._new_decoder_weights\n"
        "    ops._new_decoder_weights = lambda *a, **k: real(*a[:6])\n"
        "model, sample, meta = tasks.make_task('darcy', seed=21)\n"
        "mesh_in, func_in, mesh_out, target = sample(8)\n"
        "with torch.no_grad():\n"
        "    out = model(mesh_in, func_in, mesh_out)\n"
        "print('DIGEST', hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())\n" % os.path.dirname(here))
    digests = {}
    for name, env, args in (("lds", {}, []), ("fragments", {"PIT_NO_BLOCK_WLDS": "1"}, []), ("row-major-w1", {}, ["NOW1F"])):
        r = subprocess.run([sys.executable, "-c", code] + args, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        digests[name] = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0]
    assert digests["lds"] == digests["fragments"] == digests["row-major-w1"], digests
