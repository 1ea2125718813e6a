"""GPU: round 4 - the persistent latent kernels (csrc/pit_latent.hip: the whole processor as one launch per direction,
per-sample hand-offs inside the launch) and the fused processor against the ORACLE itself (not against
the unfused HIP path)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_io as gio
import pit_oracle as orc

pytestmark = pytest.mark.gpu
LATENT_DEFAULT = os.environ.get("PIT_LATENT_FUSION", "0") != "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METRIC = {"darcy": "euclid", "burgers": "periodic1d", "sod": "euclid"}


def _processor_inputs(task, seed, batch):
    """The task's model (train_*.py hyper-parameters, synthetic init) and a latent activation of the right shape."""
    from position_induced_transformer_amd import tasks
    model, _sample, _meta = tasks.make_task(task, seed=seed)
    L = model.mesh_ltt.shape[0]
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(batch, L, model.hid_dim, generator=g)
    lmdas = [a.lmda for a in model.conv]
    mlps = [(w.mlp1.weight, w.mlp1.bias, w.mlp2.weight, w.mlp2.bias) for w in model.mlp]
    return model, x, lmdas, mlps


def _oracle_processor(metric, mesh, x, lmdas, mlps):
    """pit.processor (pit.py:114-122) with the oracle's operators."""
    f = x
    for lm, (w1, b1, w2, b2) in zip(lmdas, mlps):
        f = orc.posatt_self(metric, False, mesh, f, lm, 1.0)
        f = F.gelu(orc.mlp(f, w1, b1, w2, b2))
    return f


def _run_processor(model, x, lmdas, mlps, d_out):
    from position_induced_transformer_amd import ops
    plan = model.conv[0]._plan(model.mesh_ltt, model.mesh_ltt, True)
    xg = x.cuda().requires_grad_(True)
    for p in list(lmdas) + [t for m in mlps for t in m]:
        p.grad = None
    out = ops.processor_apply(xg, plan, model.conv[0].n_head, lmdas, mlps)
    out.backward(d_out.cuda())
    torch.cuda.synchronize()
    grads = [xg.grad] + [p.grad for p in lmdas] + [t.grad for m in mlps for t in m]
    return out.detach().cpu(), [g.detach().cpu().clone() for g in grads]


@pytest.mark.parametrize("latent", [False, True], ids=["per-block", "persistent"])
@pytest.mark.parametrize("task,batch", [("darcy", 8), ("darcy", 3), ("burgers", 8), ("darcy", 1)])
def test_fused_processor_against_the_oracle(task, batch, latent):
    """ops.processor_apply (block weights + the persistent latent launch, or one launch per block where that does not
    apply) against the oracle's processor - posatt_self + mlp + gelu per block, pit.py:114-122 - on the same parameters
    and inputs: output <= 1e-5, d(input) and every weight gradient <= 2e-5, d(lmda) <= 2e-4 of the largest one.  Route
    'host': the head scale is the reference's own torch-CPU value, so the check holds for any seed."""
    from position_induced_transformer_amd import ops
    model, x, lmdas, mlps = _processor_inputs(task, 31, batch)
    g = torch.Generator().manual_seed(77)
    d_out = torch.randn(x.shape, generator=g)
    try:
        ops.LATENT_FUSION = latent
        with ops.head_scale_route("host"):
            out, grads = _run_processor(model, x, lmdas, mlps, d_out)
    finally:
        ops.LATENT_FUSION = LATENT_DEFAULT
    assert ops.latent_status() == 0
    xr = x.clone().requires_grad_(True)
    lm_r = [p.detach().cpu().clone().requires_grad_(True) for p in lmdas]
    ml_r = [tuple(t.detach().cpu().clone().requires_grad_(True) for t in m) for m in mlps]
    ref = _oracle_processor(METRIC[task], model.mesh_ltt.cpu(), xr, lm_r, ml_r)
    ref.backward(d_out)
    ref_grads = [xr.grad] + [p.grad for p in lm_r] + [t.grad for m in ml_r for t in m]
    assert gio.rel_l2(out.numpy(), ref.detach().numpy()) <= 1e-5
    n = len(lmdas)
    lm_scale = max(float(p.norm()) for p in ref_grads[1:1 + n])
    for k, (a, r) in enumerate(zip(grads, ref_grads)):
        if 1 <= k <= n:
            assert float((a.reshape(-1) - r.reshape(-1)).norm()) <= 2e-4 * lm_scale, f"d(lmda) of block {k - 1}"
        else:
            assert gio.rel_l2(a.numpy(), r.numpy()) <= 2e-5, f"gradient {k}"


@pytest.mark.parametrize("task,batch", [("darcy", 8), ("darcy", 3), ("darcy", 16), ("burgers", 8), ("darcy", 1)])
@pytest.mark.parametrize("linear_map", [0, 1], ids=["xcd-local", "spread-over-xcds"])
def test_latent_launch_is_bit_identical_to_one_launch_per_block(task, batch, linear_map):
    """The persistent launch runs the arithmetic of the per-block launches phase by phase: prediction and every gradient
    are EQUAL bit for bit (weight gradients: to the atomics' summation order, <= 1e-6) - with a sample's slabs on one XCD
    and, through PIT_LATENT_LINEAR_MAP, spread over all eight (the hand-off must not depend on placement)."""
    from position_induced_transformer_amd import ops
    model, x, lmdas, mlps = _processor_inputs(task, 32, batch)
    H, D, L = model.conv[0].n_head, model.hid_dim, model.mesh_ltt.shape[0]
    ops.LATENT_FUSION = True
    if not ops.latent_fusion_supported(L, H, D, batch, len(lmdas)):
        ops.LATENT_FUSION = LATENT_DEFAULT
        pytest.skip("shape not covered by the persistent kernels on this device")
    d_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    res = {}
    try:
        for latent in (True, False):
            ops.LATENT_FUSION, ops.LATENT_FLAGS = latent, linear_map
            res[latent] = _run_processor(model, x, lmdas, mlps, d_out)
    finally:
        ops.LATENT_FUSION, ops.LATENT_FLAGS = LATENT_DEFAULT, 0
    assert ops.latent_status() == 0
    assert torch.equal(res[True][0], res[False][0])
    n = len(lmdas)
    for k, (a, r) in enumerate(zip(res[True][1], res[False][1])):
        if k == 0:
            assert torch.equal(a, r), "d(input)"
        elif k <= n:
            assert float((a - r).abs().max()) <= 1e-5 * max(1e-30, float(r.abs().max())) + 1e-12, f"d(lmda) {k - 1}"
        else:
            assert gio.rel_l2(a.numpy(), r.numpy()) <= 1e-6, f"gradient {k}"


@pytest.mark.parametrize("linear_map", [0, 1], ids=["xcd-local", "spread-over-xcds"])
def test_latent_launch_replayed_under_load_never_reads_stale_rows(linear_map):
    """400 graph replays of the persistent forward + backward with another stream hammering the memory system in between
    (uneven load: workgroups of a sample reach their waits at different times): every replay reproduces the first one
    bit for bit, and no wait timed out."""
    from position_induced_transformer_amd import ops
    model, x, lmdas, mlps = _processor_inputs("darcy", 33, 8)
    plan = model.conv[0]._plan(model.mesh_ltt, model.mesh_ltt, True)
    H = model.conv[0].n_head
    d_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).cuda()
    xg = x.cuda().requires_grad_(True)
    try:
        ops.LATENT_FUSION, ops.LATENT_FLAGS = True, linear_map
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                xg.grad = None
                ops.processor_apply(xg, plan, H, lmdas, mlps).backward(d_out)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            xg.grad = None
            out = ops.processor_apply(xg, plan, H, lmdas, mlps)
            out.backward(d_out)
        graph.replay()
        torch.cuda.synchronize()
        want_out, want_dx = out.detach().clone(), xg.grad.detach().clone()
        noise = torch.empty(64 << 20, device="cuda")
        other = torch.cuda.Stream()
        for it in range(400):
            if it % 3 == 0:
                with torch.cuda.stream(other):
                    noise.mul_(1.0001)
            graph.replay()
            if it % 50 == 49:
                torch.cuda.synchronize()
                assert torch.equal(out, want_out) and torch.equal(xg.grad, want_dx), f"replay {it}"
        torch.cuda.synchronize()
        assert torch.equal(out, want_out) and torch.equal(xg.grad, want_dx)
        assert ops.latent_status() == 0
    finally:
        ops.LATENT_FUSION, ops.LATENT_FLAGS = LATENT_DEFAULT, 0
