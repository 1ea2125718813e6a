"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(position_induced_transformer_amd.ops -> ctypes -> libpit_hip.so), against
  (1) the committed golden vectors captured from the reference, and
  (2) the CPU oracle run on the same seeded inputs (full tensors).

Tolerances (fp32 path): order statistics / thresholds / mask keep-sets
exact; forward rel-L2 <= 1e-6 per operator; d values and MLP gradients <= 1e-5;
d lmda / d c <= 1e-4 (cancellation-heavy reduction)."""
import numpy as np
import pytest
import torch

import golden_io as gio
import pit_oracle as orc

pytestmark = pytest.mark.gpu

OP_CASES = gio.list_cases(("F1_", "F2_", "F3_", "F4", "F5_", "F6_", "F7_", "E"))
MLP_CASES = gio.list_cases(("F8_",))
TOL_FWD, TOL_GRAD, TOL_HEAD = 1e-6, 1e-5, 1e-4


@pytest.fixture(scope="module", params=["sparse-masked", "dense-only"])
def ops(request):
    """Every test runs twice: masked layers on the candidate-list (sparse) kernels, and with the
    dense MFMA kernels forced everywhere."""
    from position_induced_transformer_amd import ops as _ops
    assert torch.cuda.is_available()
    old = _ops.SPARSE_MASKED
    _ops.SPARSE_MASKED = request.param == "sparse-masked"
    yield _ops
    _ops.SPARSE_MASKED = old


def dev(x):
    return torch.as_tensor(np.asarray(x)).cuda()


def load_case(name):
    fx = gio.load(name)
    seed = int(fx["seed"])
    values = gio.synth(tuple(int(v) for v in fx["values_shape"]), seed)
    return fx, dict(metric=str(fx["metric"]), batched=bool(fx["batched"]), self_attn=bool(fx["self_attn"]),
                    q=float(fx["locality"]), mesh_out=fx["mesh_out"], mesh_in=fx["mesh_in"], values=values,
                    lmda=fx["lmda"], c=fx["c"], seed=seed)


def make_plan(ops, cs):
    mo = dev(cs["mesh_out"])
    mi = mo if cs["self_attn"] else dev(cs["mesh_in"])
    return ops.MeshPlan(cs["metric"], mo, mi, cs["q"], cs["self_attn"])


def test_mfma_fragment_layout(ops):
    """A = I-like with an ASYMMETRIC B catches a transposed C/D or swapped A/B map."""
    rng = np.random.RandomState(0)
    a = rng.randint(-3, 4, size=(32, 8)).astype(np.float32)
    b = rng.randint(-3, 4, size=(8, 32)).astype(np.float32)
    d = ops.debug_mfma_tile(dev(a), dev(b)).cpu().numpy()
    assert np.array_equal(d, a @ b)


@pytest.mark.parametrize("name", OP_CASES)
def test_select_order_statistics(ops, name):
    fx, cs = load_case(name)
    plan = make_plan(ops, cs)
    if plan.nbr_idx is not None:          # candidate lists: superset of every head's keep-set, k+2 + ties
        cnt = plan.nbr_cnt.cpu().numpy().reshape(plan.mesh_batch, plan.n_out)
        kc = fx["keep_count"].reshape(plan.mesh_batch, -1, plan.n_out)
        assert (cnt[:, None, :] >= kc).all()
        assert cnt.min() >= min(plan.rank_k + 2, plan.n_in)
    assert plan.rank_k == int(fx["rank_k"]) and np.float32(plan.rank_w) == fx["rank_w"]
    if cs["metric"] != "euclid":
        assert np.float32(plan.period) == fx["period"]
    if plan.stats is None:            # unmasked self attention: nothing to select
        assert not plan.masked and plan.self_attn
        return
    st = plan.stats.cpu().numpy()
    shape = fx["m_min"].shape
    assert np.array_equal(st[2].reshape(shape), fx["m_min"])
    if plan.masked:
        assert np.array_equal(st[0].reshape(shape), fx["m_k"])
        assert np.array_equal(st[1].reshape(shape), fx["m_k1"])


@pytest.mark.parametrize("name", OP_CASES)
def test_posatt_forward_backward_injected_scale(ops, name):
    """Kernel-level parity with the reference's own c injected (isolates libm)."""
    fx, cs = load_case(name)
    plan = make_plan(ops, cs)
    n_head = cs["lmda"].shape[0]
    values = dev(cs["values"]).requires_grad_(True)
    c = dev(cs["c"].reshape(-1)).requires_grad_(True)
    out = ops.posatt_apply(values, c, plan, n_head, concat=cs["self_attn"], head_is_scale=True)
    d_out = gio.synth(tuple(out.shape), cs["seed"] + 1000)
    out.backward(dev(d_out))
    e, g, ne, ng = gio.expect(fx, "out", out.detach().cpu().numpy())
    assert gio.rel_l2(e, g) <= TOL_FWD
    e, g, _, _ = gio.expect(fx, "d_values", values.grad.cpu().numpy())
    assert gio.rel_l2(e, g) <= TOL_GRAD
    assert gio.rel_l2(fx["d_c"].reshape(-1), c.grad.cpu().numpy()) <= TOL_HEAD

    # full-tensor comparison with the oracle on the same inputs
    mo, mi = torch.from_numpy(cs["mesh_out"]), torch.from_numpy(cs["mesh_in"])
    u = torch.from_numpy(cs["values"]).requires_grad_(True)
    cc = torch.from_numpy(cs["c"]).requires_grad_(True)
    if cs["self_attn"]:
        ref = orc.posatt_self(cs["metric"], cs["batched"], mo, u, None, cs["q"], c=cc)
    else:
        ref = orc.posatt_cross(cs["metric"], cs["batched"], mo, mi, u, None, cs["q"], c=cc)
    ref.backward(torch.from_numpy(d_out))
    assert gio.rel_l2(ref.detach().numpy(), out.detach().cpu().numpy()) <= TOL_FWD
    assert gio.rel_l2(u.grad.numpy(), values.grad.cpu().numpy()) <= TOL_GRAD
    assert np.abs(ref.detach().numpy() - out.detach().cpu().numpy()).max() <= 2e-5 * np.abs(ref.detach().numpy()).max()


def _c_host(lmda_np):
    return orc.head_scale(torch.from_numpy(lmda_np)).numpy()


def _ulp(a, b):
    return int(np.abs(a.reshape(-1).view(np.int32).astype(np.int64) - b.reshape(-1).view(np.int32).astype(np.int64)).max())


@pytest.mark.parametrize("draw", ["fixture", "differing"])
@pytest.mark.parametrize("name", OP_CASES)
def test_posatt_lmda_path(ops, name, draw):
    """lmda -> c inside the kernel (route 'device': fp64 evaluation) and d lmda, against the ORACLE run
    live on this host with the same lmda.

    ATen-CPU's sin/tan are MKL VML high-accuracy kernels (not correctly rounded, chosen by the host
    CPU), so the device c differs from the host's for a few per cent of lmda, by up to ~12 ulp per ulp
    of sin (tan amplifies).  draw='fixture' uses the golden case's lmda; draw='differing' SEARCHES for
    an lmda whose device c differs from this host's (the case round 1 never drew).  Asserted either way:
      (1) the kernel is exact given its c: output == oracle with the DEVICE's c injected, at TOL_FWD;
      (2) route 'host' reproduces the reference's lmda path: output, d(values), d(lmda) == oracle(lmda);
      (3) when the two c are bit-equal, route 'device' == oracle(lmda) too, d(lmda) included."""
    fx, cs = load_case(name)
    plan = make_plan(ops, cs)
    n_head = cs["lmda"].shape[0]
    lmda = cs["lmda"]
    if draw == "differing":
        rng = np.random.default_rng(cs["seed"])
        for _ in range(4000):
            cand = rng.random(lmda.shape, dtype=np.float32)
            if _ulp(_c_host(cand), ops.head_scale(dev(cand)).cpu().numpy()) > 0:
                lmda = cand
                break
        else:
            pytest.skip("no lmda with a differing c found on this host")
    c_dev = ops.head_scale(dev(lmda)).cpu().numpy()
    ulp = _ulp(_c_host(lmda), c_dev)
    assert (ulp > 0) if draw == "differing" else True
    # analytic bound: a 1-ulp difference in sin(lmda) can move fl(1+s) - and with it u = K(1+s) - by one ulp; near
    # lmda -> 1 (u = 1.446, c = 7.98) d c = (1+c^2) d u = 64.7 * 2^-23 = 16 ulp of c, plus the host tan's own < 1 ulp
    # and the device's final rounding: <= 18 (17 was observed on the build container's Xeon); 24 leaves a margin
    assert ulp <= 24, f"lmda->c differs from this host's ATen by {ulp} ulp"
    d_out = gio.synth((cs["values"].shape[0], plan.n_out, (n_head + (1 if cs["self_attn"] else 0)) * cs["values"].shape[2]),
                      cs["seed"] + 1000)

    def oracle(**kw):
        mo, mi = torch.from_numpy(cs["mesh_out"]), torch.from_numpy(cs["mesh_in"])
        u = torch.from_numpy(cs["values"]).requires_grad_(True)
        lm = torch.from_numpy(lmda).requires_grad_(True)
        if cs["self_attn"]:
            ref = orc.posatt_self(cs["metric"], cs["batched"], mo, u, lm, cs["q"], **kw)
        else:
            ref = orc.posatt_cross(cs["metric"], cs["batched"], mo, mi, u, lm, cs["q"], **kw)
        if not kw:
            ref.backward(torch.from_numpy(d_out))
        return ref.detach().numpy(), u.grad, lm.grad

    def device(route):
        lm = dev(lmda).requires_grad_(True)
        values = dev(cs["values"]).requires_grad_(True)
        with ops.head_scale_route(route):
            out = ops.posatt_apply(values, lm, plan, n_head, concat=cs["self_attn"])
        out.backward(dev(d_out))
        return out.detach().cpu().numpy(), values.grad.cpu(), lm.grad.cpu()

    ref, ref_du, ref_dl = oracle()
    out_d, du_d, dl_d = device("device")
    out_h, du_h, dl_h = device("host")
    ref_cdev, _, _ = oracle(c=torch.from_numpy(c_dev.reshape(lmda.shape)))
    assert gio.rel_l2(ref_cdev, out_d) <= TOL_FWD                                     # (1)
    assert gio.rel_l2(ref, out_h) <= TOL_FWD                                          # (2)
    assert gio.rel_l2(ref_du.numpy(), du_h.numpy()) <= TOL_GRAD
    assert gio.rel_l2(ref_dl.numpy().reshape(-1), dl_h.numpy().reshape(-1)) <= TOL_HEAD
    if ulp == 0:                                                                      # (3)
        assert gio.rel_l2(ref, out_d) <= TOL_FWD
        assert gio.rel_l2(ref_du.numpy(), du_d.numpy()) <= TOL_GRAD
        assert gio.rel_l2(ref_dl.numpy().reshape(-1), dl_d.numpy().reshape(-1)) <= TOL_HEAD
        if draw == "fixture":
            e, g, _, _ = gio.expect(fx, "out", out_d)
            assert gio.rel_l2(e, g) <= TOL_FWD


@pytest.mark.parametrize("name", ["F1_darcy_enc", "F3_darcy_dec", "F5_p1d_enc", "F6_p2d_enc", "E2_duplicates",
                                   "E1_small_scale", "F7_naca_enc"])
def test_mask_keep_sets_exact(ops, name):
    """Feed the identity as values: the output IS the attention matrix.  Its non-zero
    pattern must equal the reference's keep-set exactly (ties at the threshold included)."""
    fx, cs = load_case(name)
    plan = make_plan(ops, cs)
    n_head = cs["lmda"].shape[0]
    j = plan.n_in
    mb = plan.mesh_batch
    eye = torch.eye(j, device="cuda").unsqueeze(0).repeat(mb, 1, 1).contiguous()
    att = ops.posatt_apply(eye, dev(cs["c"].reshape(-1)), plan, n_head, concat=False, head_is_scale=True)
    att = att.reshape(mb, plan.n_out, n_head, j).permute(0, 2, 1, 3).cpu()          # (mb,H,N,J)
    m = orc.sqdist(cs["metric"], torch.from_numpy(cs["mesh_out"]), torch.from_numpy(cs["mesh_in"]))
    ref = orc.attention_weights(m, torch.from_numpy(cs["c"]), cs["q"], cs["batched"])
    if not cs["batched"]:
        ref = ref.unsqueeze(0)
    assert torch.equal(att > 0, ref > 0), "mask keep-set differs from the reference"
    assert np.array_equal((att > 0).sum(-1).numpy().astype(np.int16).reshape(fx["keep_count"].shape), fx["keep_count"])
    assert gio.rel_l2(ref.numpy(), att.numpy()) <= TOL_FWD
    assert np.allclose(att.sum(-1).numpy(), 1.0, atol=2e-6)


@pytest.mark.parametrize("name", MLP_CASES)
def test_mlp_forward_backward(ops, name):
    fx = gio.load(name)
    n0, n1, n2 = (int(v) for v in fx["dims"])
    seed = int(fx["seed"])
    shapes = [("mlp1.weight", (n1, n0)), ("mlp1.bias", (n1,)), ("mlp2.weight", (n2, n1)), ("mlp2.bias", (n2,))]
    p = {k: dev(v).requires_grad_(True) for k, v in gio.synth_params(shapes, seed).items()}
    rows = tuple(int(v) for v in fx["rows"])
    x = dev(gio.synth(rows + (n0,), seed + 1)).requires_grad_(True)
    y = ops.mlp_apply(x, p["mlp1.weight"], p["mlp1.bias"], p["mlp2.weight"], p["mlp2.bias"])
    y.backward(dev(gio.synth(tuple(y.shape), seed + 2)))
    got = {"y": y.detach(), "d_x": x.grad, "d_w1": p["mlp1.weight"].grad, "d_b1": p["mlp1.bias"].grad,
           "d_w2": p["mlp2.weight"].grad, "d_b2": p["mlp2.bias"].grad}
    for key, val in got.items():
        e, g, _, _ = gio.expect(fx, key, val.cpu().numpy())
        assert gio.rel_l2(e, g) <= (TOL_FWD if key == "y" else TOL_GRAD), key


@pytest.mark.parametrize("rows,n0,n1,n2", [(300, 24, 32, 32), (1000, 192, 64, 64), (77, 5, 130, 3),
                                            (3000, 200, 130, 70), (2304, 768, 256, 256),    # LDS-tiled GEMM
                                            (20000, 128, 64, 1), (9000, 64, 128, 4),        # thin output layer (n2 <= 4)
                                            (8200, 32, 256, 3)])
def test_mlp_trailing_gelu_vs_oracle(ops, rows, n0, n1, n2):
    """The fused trailing gelu of pit.py:111,121 (out_gelu=True) against torch on the CPU."""
    shapes = [("mlp1.weight", (n1, n0)), ("mlp1.bias", (n1,)), ("mlp2.weight", (n2, n1)), ("mlp2.bias", (n2,))]
    pc = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gio.synth_params(shapes, 7).items()}
    pg = {k: v.detach().cuda().requires_grad_(True) for k, v in pc.items()}
    xc = torch.from_numpy(gio.synth((rows, n0), 8)).requires_grad_(True)
    xg = xc.detach().cuda().requires_grad_(True)
    dy = torch.from_numpy(gio.synth((rows, n2), 9))
    yc = torch.nn.functional.gelu(orc.mlp(xc, pc["mlp1.weight"], pc["mlp1.bias"], pc["mlp2.weight"], pc["mlp2.bias"]))
    yc.backward(dy)
    yg = ops.mlp_apply(xg, pg["mlp1.weight"], pg["mlp1.bias"], pg["mlp2.weight"], pg["mlp2.bias"], True)
    yg.backward(dy.cuda())
    assert gio.rel_l2(yc.detach().numpy(), yg.detach().cpu().numpy()) <= TOL_FWD
    assert gio.rel_l2(xc.grad.numpy(), xg.grad.cpu().numpy()) <= TOL_GRAD
    for k in pc:
        assert gio.rel_l2(pc[k].grad.numpy(), pg[k].grad.cpu().numpy()) <= TOL_GRAD, k


@pytest.mark.parametrize("p,affine", [(2, False), (1, False), (2, True), (3, False)])
def test_rel_lp_loss(ops, p, affine):
    t = torch.from_numpy(gio.synth((3, 50, 2), 31))
    q = torch.from_numpy(gio.synth((3, 50, 2), 32)).requires_grad_(True)
    sc = torch.from_numpy(gio.synth((50, 2), 33, 0.5, 1.5)) if affine else None
    sh = torch.from_numpy(gio.synth((50, 2), 34)) if affine else None
    qq = q * sc + sh if affine else q
    ref = orc.rel_lp_loss(t, qq, 2, p)
    ref.backward()
    qg = q.detach().cuda().requires_grad_(True)
    got = ops.rel_lp_loss(t.cuda(), qg, 2, p, sc.cuda() if affine else None, sh.cuda() if affine else None)
    (got * 1.0).backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref))
    assert gio.rel_l2(q.grad.numpy(), qg.grad.cpu().numpy()) <= TOL_GRAD


@pytest.mark.parametrize("p", [1, 2, 3])
def test_rel_lp_loss_gradient_wrt_true_argument(ops, p):
    """train_vorticity.py:124 / train_cylinder.py:101 call myloss(out, y): the model output is the
    FIRST ('true') argument, so the loss must differentiate through the denominator too."""
    t = torch.from_numpy(gio.synth((3, 40, 2), 35)).requires_grad_(True)
    q = torch.from_numpy(gio.synth((3, 40, 2), 36)).requires_grad_(True)
    ref = orc.rel_lp_loss(t, q, 2, p)
    ref.backward()
    tg = t.detach().cuda().requires_grad_(True)
    qg = q.detach().cuda().requires_grad_(True)
    got = ops.rel_lp_loss(tg, qg, 2, p)
    got.backward()
    assert abs(float(got.detach()) - float(ref.detach())) <= 2e-6 * abs(float(ref.detach()))
    assert gio.rel_l2(t.grad.numpy(), tg.grad.cpu().numpy()) <= TOL_GRAD
    assert gio.rel_l2(q.grad.numpy(), qg.grad.cpu().numpy()) <= TOL_GRAD


def test_cpu_tensors_fail_loudly(ops):
    with pytest.raises(RuntimeError):
        ops.MeshPlan("euclid", torch.zeros(4, 2), torch.zeros(5, 2), 0.5, False)
    with pytest.raises(RuntimeError):
        ops.mlp_apply(torch.zeros(3, 4), torch.zeros(5, 4), torch.zeros(5), torch.zeros(2, 5), torch.zeros(2))


def test_abi_argument_errors(ops):
    from position_induced_transformer_amd import _lib
    L = _lib.lib()
    assert L.pit_version() >= 1
    assert L.pit_select_fwd(0, 0, 1, 4, 4, 2, 0, 0.0, 0, 1, 0, 0) == -1           # NULL pointers
    buf = torch.zeros(64, device="cuda")
    assert L.pit_select_fwd(buf.data_ptr(), buf.data_ptr(), 1, 4, 4, 7, 0, 0.0, 0, 1, buf.data_ptr(), 0) == -2
    assert L.pit_select_fwd(buf.data_ptr(), buf.data_ptr(), 1, 4, 4, 2, 9, 0.0, 0, 1, buf.data_ptr(), 0) == -3
    assert b"NULL" in L.pit_error_string(-1)


# --------------------------------------------------------------------------- full-size properties
@pytest.mark.parametrize("task,batch", [("darcy", 8), ("elasticity", 10)])
def test_full_size_properties(ops, task, batch):
    """At BASELINE.json's sizes the oracle is too slow for a per-test budget, so check
    size-independent properties of the fused operator: rows of the attention sum to one
    (values == 1 -> output == 1), linearity in the values, and - for the full model -
    hipGraph replay == eager."""
    from position_induced_transformer_amd import tasks
    model, sample, meta = tasks.make_task(task, seed=3)
    mesh_in, func_in, mesh_out, target = sample(batch)
    layer = model.up
    if task == "darcy":
        mo, mi = mesh_out.reshape(-1, 2), model.mesh_ltt
    else:
        mo, mi = mesh_out, mesh_out
    plan = layer._plan(mo, mi, False)
    ones = torch.ones(batch, plan.n_in, 64, device="cuda")
    out = ops.posatt_apply(ones, layer.lmda, plan, layer.n_head, False)
    assert torch.allclose(out.detach(), torch.ones_like(out), atol=3e-6)
    u = torch.randn(batch, plan.n_in, 64, device="cuda")
    v = torch.randn(batch, plan.n_in, 64, device="cuda")
    lin = ops.posatt_apply(2.0 * u - 3.0 * v, layer.lmda, plan, layer.n_head, False)
    sep = 2.0 * ops.posatt_apply(u, layer.lmda, plan, layer.n_head, False) \
        - 3.0 * ops.posatt_apply(v, layer.lmda, plan, layer.n_head, False)
    assert gio.rel_l2(sep.detach().cpu().numpy(), lin.detach().cpu().numpy()) <= 2e-6
    # adjoint identity <dO, A U> == <A^T dO, U> ties forward and d-values kernels together
    u.requires_grad_(True)
    o = ops.posatt_apply(u, layer.lmda, plan, layer.n_head, False)
    d_o = torch.randn_like(o)
    o.backward(d_o)
    lhs = float((d_o.double() * o.detach().double()).sum())
    rhs = float((u.grad.double() * u.detach().double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("n_out,n_in,q", [(70, 5000, 0.01), (40, 9000, 0.002)])
def test_long_rows_stream_select_and_chunked_keys(ops, n_out, n_in, q):
    """J > 4096 (zero-shot super-resolution regime, train_darcy.py:152-178): the streaming
    select kernel and the multi-chunk key loop, against the oracle on the same inputs."""
    mo = torch.from_numpy(gio.synth((n_out, 2), 41, 0.0, 1.0))
    mi = torch.from_numpy(gio.synth((n_in, 2), 42, 0.0, 1.0))
    u = torch.from_numpy(gio.synth((2, n_in, 5), 43))
    c = torch.tensor([1.7, 4.2]).reshape(2, 1, 1)
    plan = ops.MeshPlan("euclid", mo.cuda(), mi.cuda(), q, False)
    m = orc.sqdist_euclid(mo, mi)
    mk, mk1, mmin = orc.row_order_stats(m, q)
    st = plan.stats.cpu()
    assert torch.equal(st[0, 0], mk) and torch.equal(st[1, 0], mk1) and torch.equal(st[2, 0], mmin)
    ug = u.cuda().requires_grad_(True)
    out = ops.posatt_apply(ug, c.reshape(-1).cuda(), plan, 2, False, head_is_scale=True)
    uc = u.clone().requires_grad_(True)
    ref = orc.posatt_cross("euclid", False, mo, mi, uc, None, q, c=c)
    d_out = torch.from_numpy(gio.synth(tuple(ref.shape), 44))
    ref.backward(d_out)
    out.backward(d_out.cuda())
    assert gio.rel_l2(ref.detach().numpy(), out.detach().cpu().numpy()) <= TOL_FWD
    assert gio.rel_l2(uc.grad.numpy(), ug.grad.cpu().numpy()) <= TOL_GRAD


@pytest.mark.parametrize("batched,q,self_attn,d,b,big", [(False, 1.0, True, 70, 6, False), (False, 0.3, False, 70, 6, False),
                                                          (True, 1.0, True, 300, 2, False), (True, 0.4, False, 260, 2, False),
                                                          (False, 1.0, False, 64, 20, False),
                                                          (False, 1.0, True, 64, 12, True), (False, 0.5, False, 64, 12, True)])
@pytest.mark.parametrize("force_rt", ["1", "2", "4"])
def test_wide_column_kernels_vs_oracle(ops, monkeypatch, batched, q, self_attn, d, b, big, force_rt):
    """>= 8 column tiles per row tile: the large-regime kernels (weights in LDS, whole tiles per
    wave, 1/2/4 row tiles per workgroup), forward, d(values) and d(scale), against the oracle;
    ragged sizes, 3 heads.  (The work threshold is bypassed so small test shapes take this path.)"""
    monkeypatch.setenv("PIT_FORCE_TILES", "1")
    monkeypatch.setenv("PIT_FORCE_RT", force_rt)
    n_out, n_in = (150, 150) if self_attn else (100, 310)
    if big:                               # complete 256-row / 256-key chunks: the unchecked two-groups-per-trip loops
        n_out, n_in = (300, 300) if self_attn else (530, 270)
    shape_o = (b, n_out, 2) if batched else (n_out, 2)
    shape_i = (b, n_in, 2) if batched else (n_in, 2)
    mo = torch.from_numpy(gio.synth(shape_o, 51, 0.0, 1.0))
    mi = mo if self_attn else torch.from_numpy(gio.synth(shape_i, 52, 0.0, 1.0))
    u = torch.from_numpy(gio.synth((b, n_in, d), 53))
    c = torch.tensor([0.8, 2.1, 4.4]).reshape(3, 1, 1)
    plan = ops.MeshPlan("euclid", mo.cuda(), (mo if self_attn else mi).cuda(), q, self_attn)
    ug = u.cuda().requires_grad_(True)
    cg = c.reshape(-1).cuda().requires_grad_(True)
    out = ops.posatt_apply(ug, cg, plan, 3, concat=self_attn, head_is_scale=True)
    uc = u.clone().requires_grad_(True)
    cc = c.clone().requires_grad_(True)
    if self_attn:
        ref = orc.posatt_self("euclid", batched, mo, uc, None, q, c=cc)
    else:
        ref = orc.posatt_cross("euclid", batched, mo, mi, uc, None, q, c=cc)
    d_out = torch.from_numpy(gio.synth(tuple(ref.shape), 54))
    ref.backward(d_out)
    out.backward(d_out.cuda())
    assert gio.rel_l2(ref.detach().numpy(), out.detach().cpu().numpy()) <= TOL_FWD
    assert gio.rel_l2(uc.grad.numpy(), ug.grad.cpu().numpy()) <= TOL_GRAD
    assert gio.rel_l2(cc.grad.reshape(-1).numpy(), cg.grad.cpu().numpy()) <= TOL_HEAD


def test_raw_ctypes_binding_as_in_integration_md():
    """The stub of INTEGRATION.md section 2, verbatim in spirit: bind libpit_hip.so with plain
    ctypes (no ops.py), run posatt_cross_fixed.forward through pit_select_fwd + pit_posatt_fwd
    and compare with the oracle."""
    import ctypes
    from position_induced_transformer_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    P, I, F, LG = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long
    L.pit_select_fwd.argtypes = [P, P, I, I, I, I, I, F, I, I, P, P]
    L.pit_posatt_fwd.argtypes = [P, P, I, I, I, I, I, F, P, I, I, LG, LG, P, I, I, P, F, I, I,
                                 P, LG, LG, I, I, P, P, P, P, I, I, I, P]
    fx, cs = load_case("F1_darcy_enc")
    mesh_out, mesh_in = dev(cs["mesh_out"]).contiguous(), dev(cs["mesh_in"]).contiguous()   # the ABI takes dense rows
    inputs, lmda = dev(cs["values"]).contiguous(), dev(cs["lmda"]).contiguous()
    n, j, s = mesh_out.shape[0], mesh_in.shape[0], mesh_out.shape[1]
    b, _, d = inputs.shape
    h, q = lmda.shape[0], cs["q"]
    rank = torch.tensor(q, dtype=torch.float32) * torch.tensor(j - 1, dtype=torch.float32)
    k = int(rank.floor())
    w = float(rank - k)
    stream = torch.cuda.current_stream().cuda_stream
    stats = torch.empty(3, 1, n, device="cuda")
    assert L.pit_select_fwd(mesh_out.data_ptr(), mesh_in.data_ptr(), 1, n, j, s, 0, 0.0, k, 1, stats.data_ptr(), stream) == 0
    out = torch.empty(b, n, h * d, device="cuda")
    rowstat = torch.empty(1, h, n, 4, device="cuda")
    scale = torch.empty(h, device="cuda")

    def forward(head, head_is_scale):
        rc = L.pit_posatt_fwd(mesh_out.data_ptr(), mesh_in.data_ptr(), 1, n, j, s, 0, 0.0,
                              inputs.data_ptr(), b, d, inputs.stride(1), inputs.stride(0),
                              head.data_ptr(), h, head_is_scale, stats.data_ptr(), w, 1, 0,
                              out.data_ptr(), out.stride(1), out.stride(0), 0, 0,
                              rowstat.data_ptr(), scale.data_ptr(), None, None, 0, 0, 0, stream)   # coord_dims 0, PIT_MATH_FP32
        assert rc == 0
        torch.cuda.synchronize()
        return out.cpu().numpy(), scale.cpu().numpy()

    # (1) the reference's own c (stored with the fixture) injected: the golden output, no libm in the way
    got, c_used = forward(dev(cs["c"].reshape(-1)).contiguous(), 1)
    assert np.array_equal(c_used, cs["c"].reshape(-1))
    e, g, _, _ = gio.expect(fx, "out", got)
    assert gio.rel_l2(e, g) <= TOL_FWD
    # (2) lmda handed over, c evaluated in the kernel (within a few ulp of any host's libm, see test_posatt_lmda_path):
    #     the output is the oracle's for exactly THAT c - whatever it is, on every host
    got, c_used = forward(lmda, 0)
    assert _ulp(c_used, cs["c"].reshape(-1)) <= 24
    with torch.no_grad():
        ref = orc.posatt_cross("euclid", False, torch.from_numpy(cs["mesh_out"]), torch.from_numpy(cs["mesh_in"]),
                               torch.from_numpy(cs["values"]), None, q, c=torch.from_numpy(c_used.reshape(cs["lmda"].shape)))
    assert gio.rel_l2(ref.numpy(), got) <= TOL_FWD


def test_overflowed_candidate_lists_take_the_overflow_pass():
    """MeshPlan.lists_complete(): 1 for a mesh without ties (the d(values) overflow pass is skipped),
    0 when rows overflow their list capacity (many coincident keys) - and then the sparse backward
    must still equal the dense one."""
    from position_induced_transformer_amd import ops as O
    g = torch.Generator().manual_seed(5)
    n_out, n_in, dim, n_head = 96, 400, 16, 2
    mesh_out = torch.rand(n_out, 2, generator=g).cuda()
    clean = torch.rand(n_in, 2, generator=g)
    dup = clean.clone()
    dup[:120] = dup[0]                                  # 120 coincident keys: ties far beyond any capacity
    old = O.SPARSE_MASKED
    try:
        res = {}
        for label, mesh_in in (("clean", clean.cuda()), ("dup", dup.cuda())):
            for sparse in (True, False):
                O.SPARSE_MASKED = sparse
                plan = O.MeshPlan("euclid", mesh_out, mesh_in, 0.05, False)
                if sparse:
                    assert plan.nbr_idx is not None
                    assert plan.lists_complete() == (1 if label == "clean" else 0)
                    assert (plan.nbr_cnt > plan.nbr_cap).any().item() == (label == "dup")
                values = torch.from_numpy(gio.synth((3, n_in, dim), 77)).cuda().requires_grad_(True)
                lm = torch.from_numpy(gio.synth((n_head,), 78)).cuda().requires_grad_(True)
                out = O.posatt_apply(values, lm, plan, n_head, concat=False)
                out.backward(torch.from_numpy(gio.synth(tuple(out.shape), 79)).cuda())
                res[(label, sparse)] = (out.detach().cpu().numpy(), values.grad.cpu().numpy(), lm.grad.cpu().numpy())
        for label in ("clean", "dup"):
            s, d = res[(label, True)], res[(label, False)]
            assert gio.rel_l2(d[0], s[0]) <= TOL_FWD and gio.rel_l2(d[1], s[1]) <= TOL_GRAD
            assert gio.rel_l2(d[2], s[2]) <= TOL_HEAD
    finally:
        O.SPARSE_MASKED = old


@pytest.mark.parametrize("b,npts,nch", [(3, 256, 256), (2, 100, 70), (1, 7, 1)])
def test_instance_norm_points_vs_torch(b, npts, nch):
    """pit_instance_norm_fwd/bwd against nn.InstanceNorm1d applied as train_vorticity.py:56 does
    (permute - norm - permute), on the CPU."""
    from position_induced_transformer_amd import ops as O
    xc = torch.from_numpy(gio.synth((b, npts, nch), 31) * 3.0 + 0.5).requires_grad_(True)
    dy = torch.from_numpy(gio.synth((b, npts, nch), 32))
    ref = torch.nn.InstanceNorm1d(nch)(xc.permute(0, 2, 1)).permute(0, 2, 1)
    ref.backward(dy)
    xg = xc.detach().cuda().requires_grad_(True)
    got = O.instance_norm_points(xg, 1e-5)
    got.backward(dy.cuda())
    assert gio.rel_l2(ref.detach().numpy(), got.detach().cpu().numpy()) <= TOL_FWD
    assert gio.rel_l2(xc.grad.numpy(), xg.grad.cpu().numpy()) <= TOL_GRAD
    # a strided view (columns of a wider buffer) goes through the row stride, not a copy
    wide = torch.zeros(b, npts, nch + 5, device="cuda")
    wide[..., 2:2 + nch] = xc.detach().cuda()
    got2 = O.instance_norm_points(wide[..., 2:2 + nch], 1e-5)
    assert torch.equal(got2, got.detach())


def test_checked_load_variant_in_a_subprocess():
    """Tensors of 2 GiB and more cannot use the scalar-offset value loads (offset wrap); that variant
    of the dense kernels is forced with PIT_NO_FAST_LOADS=1 and must pass the same operator parity
    tests (run in a child process: the library reads the variable per call, the parent stays clean)."""
    import os, subprocess, sys
    env = dict(os.environ, PIT_NO_FAST_LOADS="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_ops.py"), "-q", "-m", "gpu", "-x",
                        "-k", "posatt_forward_backward_injected_scale and dense-only or mask_keep_sets_exact and dense-only"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("rows,n0,n1,n2", [(20000, 128, 64, 1), (16400, 256, 128, 4)])
def test_mlp_thin_output_layer_vs_oracle(ops, rows, n0, n1, n2):
    """The decoder's output MLP (pit.py:106, out_dim in {1,4}) at large row counts takes the thin
    kernels (row dots / outer product / column sums instead of MFMA tiles); no trailing gelu."""
    shapes = [("mlp1.weight", (n1, n0)), ("mlp1.bias", (n1,)), ("mlp2.weight", (n2, n1)), ("mlp2.bias", (n2,))]
    pc = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gio.synth_params(shapes, 17).items()}
    pg = {k: v.detach().cuda().requires_grad_(True) for k, v in pc.items()}
    xc = torch.from_numpy(gio.synth((rows, n0), 18)).requires_grad_(True)
    xg = xc.detach().cuda().requires_grad_(True)
    dy = torch.from_numpy(gio.synth((rows, n2), 19))
    yc = orc.mlp(xc, pc["mlp1.weight"], pc["mlp1.bias"], pc["mlp2.weight"], pc["mlp2.bias"])
    yc.backward(dy)
    yg = ops.mlp_apply(xg, pg["mlp1.weight"], pg["mlp1.bias"], pg["mlp2.weight"], pg["mlp2.bias"], False)
    yg.backward(dy.cuda())
    assert gio.rel_l2(yc.detach().numpy(), yg.detach().cpu().numpy()) <= TOL_FWD
    assert gio.rel_l2(xc.grad.numpy(), xg.grad.cpu().numpy()) <= TOL_GRAD
    for k in pc:
        assert gio.rel_l2(pc[k].grad.numpy(), pg[k].grad.cpu().numpy()) <= TOL_GRAD, k


@pytest.mark.parametrize("p,affine", [(2, True), (1, False), (3, False)])
def test_rel_lp_loss_unit_seed_and_clear(p, affine):
    """pit_rel_lp_loss_fwd_grad: gradients written by the forward launch for a seed of ones equal the
    general backward's, a different upstream gradient falls back to the backward kernel, and the
    `clear` buffer is zeroed."""
    from position_induced_transformer_amd import ops as O
    true = torch.from_numpy(gio.synth((3, 50, 2), 41)).cuda().requires_grad_(True)
    pred0 = torch.from_numpy(gio.synth((3, 50, 2), 42)).cuda()
    sc = (torch.from_numpy(gio.synth((50, 2), 43)).abs() + 0.5).cuda() if affine else None
    sh = torch.from_numpy(gio.synth((50, 2), 44)).cuda() if affine else None
    seed = torch.ones((), device="cuda")

    def grads(**kw):
        pred = pred0.clone().requires_grad_(True)
        t = true.detach().clone().requires_grad_(True)
        loss = O.rel_lp_loss(t, pred, 2, p, sc, sh, **kw)
        torch.autograd.backward(loss, grad_tensors=seed)
        return float(loss), pred.grad.clone(), t.grad.clone()

    buf = torch.full((1000,), 3.0, device="cuda")
    l0, gp0, gt0 = grads()
    l1, gp1, gt1 = grads(unit_seed=seed, clear=buf)
    assert abs(l0 - l1) <= 1e-6 * abs(l0) and float(buf.abs().max()) == 0.0     # (the loss is summed with float atomics)
    assert gio.rel_l2(gp0.cpu().numpy(), gp1.cpu().numpy()) <= 1e-6
    assert gio.rel_l2(gt0.cpu().numpy(), gt1.cpu().numpy()) <= 1e-6
    # a different upstream gradient must not use the stored unit gradients
    pred = pred0.clone().requires_grad_(True)
    loss = O.rel_lp_loss(true.detach(), pred, 2, p, sc, sh, unit_seed=seed)
    torch.autograd.backward(loss, grad_tensors=torch.full((), 2.5, device="cuda"))
    assert gio.rel_l2(2.5 * gp0.cpu().numpy(), pred.grad.cpu().numpy()) <= 1e-6
