"""GPU: randomised shape sweep of the attention operator against the CPU oracle - odd sizes that
the fixture cases do not hit (key counts that leave partial steps / partial chunks, rows and
columns that are not multiples of the 32-wide MFMA tiles, 1-3 coordinates, 1-3 heads, batch-free
and per-sample meshes, masked and unmasked, self and cross attention, both math modes)."""
import numpy as np
import pytest
import torch

import golden_io as gio
import pit_oracle as orc

pytestmark = pytest.mark.gpu


def _case(seed):
    r = np.random.RandomState(seed)
    metric = ["euclid", "euclid", "periodic1d", "periodic2d"][r.randint(4)]
    batched = bool(r.randint(2)) and metric == "euclid"
    self_attn = bool(r.randint(2))
    sdim = {"periodic1d": 1, "periodic2d": 2}.get(metric, int(r.randint(1, 4)))
    n_in = int(r.choice([2, 7, 31, 33, 64, 100, 129, 257, 400, 515, 1000, 2100]))
    if metric == "periodic2d":
        res = int(r.choice([3, 5, 8, 11, 16, 23]))
        n_in = res * res
    n_out = n_in if self_attn else int(r.choice([1, 5, 32, 45, 96, 130, 300, 700]))
    batch = int(r.choice([1, 2, 3, 5]))
    dim = int(r.choice([1, 3, 8, 17, 32, 40, 64, 100]))
    n_head = int(r.choice([1, 2, 3]))
    q = float(r.choice([1.0, 1.0, 0.5, 0.1, 0.03]))
    return dict(metric=metric, batched=batched, self_attn=self_attn, sdim=sdim, n_in=n_in, n_out=n_out, batch=batch,
                dim=dim, n_head=n_head, q=q, seed=seed)


def _meshes(cs):
    g = torch.Generator().manual_seed(cs["seed"])
    lead = (cs["batch"],) if cs["batched"] else ()
    if cs["metric"] == "periodic1d":
        mi = torch.linspace(0, 1, cs["n_in"] + 1)[:-1].reshape(-1, 1)
        mo = mi if cs["self_attn"] else torch.rand(cs["n_out"], 1, generator=g)
    elif cs["metric"] == "periodic2d":
        res = int(round(cs["n_in"] ** 0.5))
        x = torch.linspace(0, 1, res + 1)[:-1]
        mi = torch.stack(torch.meshgrid(x, x, indexing="ij"), -1).reshape(-1, 2)
        mo = mi if cs["self_attn"] else torch.rand(cs["n_out"], 2, generator=g)
    else:
        mi = torch.rand(*lead, cs["n_in"], cs["sdim"], generator=g)
        mo = mi if cs["self_attn"] else torch.rand(*lead, cs["n_out"], cs["sdim"], generator=g)
    return mo.contiguous(), mi.contiguous()


@pytest.mark.parametrize("seed", range(48))
@pytest.mark.parametrize("sparse", [True, False])
def test_random_attention_case_vs_oracle(seed, sparse):
    from position_induced_transformer_amd import ops
    cs = _case(1000 + seed)
    mo, mi = _meshes(cs)
    g = torch.Generator().manual_seed(cs["seed"] + 1)
    values = torch.randn(cs["batch"], cs["n_in"], cs["dim"], generator=g)
    lmda = torch.rand(cs["n_head"], 1, 1, generator=g) * 1.2 - 0.1
    c_ref = orc.head_scale(lmda)
    old = ops.SPARSE_MASKED
    ops.SPARSE_MASKED = sparse
    try:
        plan = ops.MeshPlan(cs["metric"], mo.cuda(), mo.cuda() if cs["self_attn"] else mi.cuda(), cs["q"], cs["self_attn"])
        vg = values.cuda().requires_grad_(True)
        cg = c_ref.reshape(-1).cuda().requires_grad_(True)             # inject the reference's scale: isolates libm
        out = ops.posatt_apply(vg, cg, plan, cs["n_head"], concat=cs["self_attn"], head_is_scale=True)
        d_out = torch.randn(out.shape, generator=g)
        out.backward(d_out.cuda())
    finally:
        ops.SPARSE_MASKED = old
    vc = values.clone().requires_grad_(True)
    cc = c_ref.clone().requires_grad_(True)
    if cs["self_attn"]:
        ref = orc.posatt_self(cs["metric"], cs["batched"], mo, vc, None, cs["q"], c=cc)
    else:
        ref = orc.posatt_cross(cs["metric"], cs["batched"], mo, mi, vc, None, cs["q"], c=cc)
    ref.backward(d_out)
    info = str(cs)
    assert gio.rel_l2(ref.detach().numpy(), out.detach().cpu().numpy()) <= 2e-6, info
    assert gio.rel_l2(vc.grad.numpy(), vg.grad.cpu().numpy()) <= 1e-5, info
    gc = cc.grad.reshape(-1).numpy()
    assert np.abs(gc - cg.grad.cpu().numpy()).max() <= 2e-4 * max(np.abs(gc).max(), 1e-6) + 1e-9, info


@pytest.mark.parametrize("seed", range(24))
def test_random_mlp_case_vs_torch(seed):
    """Random MLP shapes across the kernel regimes (register-direct, merged backward launches,
    LDS-tiled, thin output layer), with and without the trailing gelu, against torch on the CPU."""
    from position_induced_transformer_amd import ops
    r = np.random.RandomState(500 + seed)
    rows = int(r.choice([37, 300, 2048, 5000, 9000, 20000]))
    n0 = int(r.choice([3, 12, 64, 130, 192]))
    n1 = int(r.choice([8, 64, 130, 256]))
    n2 = int(r.choice([1, 4, 33, 64]))
    out_gelu = bool(r.randint(2))
    shapes = [("mlp1.weight", (n1, n0)), ("mlp1.bias", (n1,)), ("mlp2.weight", (n2, n1)), ("mlp2.bias", (n2,))]
    pc = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gio.synth_params(shapes, 600 + seed).items()}
    pg = {k: v.detach().cuda().requires_grad_(True) for k, v in pc.items()}
    xc = torch.from_numpy(gio.synth((rows, n0), 700 + seed)).requires_grad_(True)
    xg = xc.detach().cuda().requires_grad_(True)
    dy = torch.from_numpy(gio.synth((rows, n2), 800 + seed))
    yc = orc.mlp(xc, pc["mlp1.weight"], pc["mlp1.bias"], pc["mlp2.weight"], pc["mlp2.bias"])
    if out_gelu:
        yc = torch.nn.functional.gelu(yc)
    yc.backward(dy)
    yg = ops.mlp_apply(xg, pg["mlp1.weight"], pg["mlp1.bias"], pg["mlp2.weight"], pg["mlp2.bias"], out_gelu)
    yg.backward(dy.cuda())
    info = f"rows={rows} n0={n0} n1={n1} n2={n2} out_gelu={out_gelu}"
    assert gio.rel_l2(yc.detach().numpy(), yg.detach().cpu().numpy()) <= 2e-6, info
    assert gio.rel_l2(xc.grad.numpy(), xg.grad.cpu().numpy()) <= 1e-5, info
    for k in pc:
        assert gio.rel_l2(pc[k].grad.numpy(), pg[k].grad.cpu().numpy()) <= 1e-5, (info, k)


@pytest.mark.parametrize("seed", range(10))
def test_random_model_vs_oracle(seed):
    """Whole models with random hyper-parameters (heads, blocks, widths, localities, mesh kinds) and
    random-cloud meshes (no ties, so a 1-ulp difference in the head scale cannot move the mask):
    output, loss and every parameter gradient against the CPU oracle."""
    from position_induced_transformer_amd import pit as P
    from position_induced_transformer_amd import utils
    r = np.random.RandomState(9000 + seed)
    g = torch.Generator().manual_seed(9000 + seed)
    batched = bool(r.randint(2))
    sdim = int(r.choice([1, 2, 3]))
    in_dim, out_dim = int(r.choice([1, 3, 5])), int(r.choice([1, 2, 4]))
    hid, n_head, n_blocks = int(r.choice([16, 32, 48])), int(r.choice([1, 2, 3])), int(r.choice([1, 2, 3]))
    n_in, n_ltt, n_out = int(r.choice([60, 131, 300])), int(r.choice([24, 50, 97])), int(r.choice([45, 131, 257]))
    en_loc, de_loc = float(r.choice([0.05, 0.2, 1.0])), float(r.choice([0.05, 0.3]))
    b = int(r.choice([1, 2, 3]))
    lead = (b,) if batched else ()
    mesh_in = torch.rand(*lead, n_in, sdim, generator=g)
    mesh_ltt = torch.rand(*lead, n_ltt, sdim, generator=g)
    mesh_out = torch.rand(*lead, n_out, sdim, generator=g)
    func_in = torch.randn(b, n_in, in_dim + sdim, generator=g)      # the task forwards concatenate the coordinates (pit.py:100)
    target = torch.randn(b, n_out, out_dim, generator=g)
    cls = P.pit if batched else P.pit_fixed
    torch.manual_seed(100 + seed)
    model = cls(sdim, in_dim, out_dim, hid, n_head, n_blocks, None if batched else mesh_ltt.cuda(), en_loc, de_loc).cuda()
    ltt_dev = mesh_ltt.cuda()
    f = model.encoder(mesh_in.cuda(), func_in.cuda(), ltt_dev)
    f = model.processor(f, ltt_dev)
    out = model.decoder(ltt_dev, f, mesh_out.cuda())
    loss = utils.RelLpNorm(out_dim, 2)(target.cuda(), out)
    loss.backward()

    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ref = orc.pit_apply(p, "euclid", batched, n_blocks, en_loc, de_loc, mesh_in, func_in, mesh_ltt, mesh_out)
    rl = orc.rel_lp_loss(target, ref, out_dim, 2)
    rl.backward()
    info = dict(batched=batched, sdim=sdim, in_dim=in_dim, out_dim=out_dim, hid=hid, H=n_head, blocks=n_blocks,
                n=(n_in, n_ltt, n_out), loc=(en_loc, de_loc), b=b)
    assert gio.rel_l2(ref.detach().numpy(), out.detach().cpu().numpy()) <= 1e-5, info
    assert abs(float(loss.detach()) - float(rl.detach())) <= 1e-5 * abs(float(rl.detach())), info
    for k, q in model.named_parameters():
        tol = 5e-4 if k.endswith("lmda") else 5e-5
        assert gio.rel_l2(p[k].grad.numpy(), q.grad.cpu().numpy()) <= tol, (k, info)
