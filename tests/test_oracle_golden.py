"""CPU: the oracle (oracle/pit_oracle.py) against the committed golden vectors that
oracle/make_golden.py captured from the reference (pit.py imported unmodified).

Bars: thresholds, order statistics and keep counts exact; outputs and gradients
rel-L2 <= 2e-6 (they were bit-equal to the reference on the generating machine; the
slack only absorbs BLAS/vector-ISA differences between hosts)."""
import numpy as np
import pytest
import torch

import golden_io as gio
import model_cases as mc
import pit_oracle as orc

OP_CASES = gio.list_cases(("F1_", "F2_", "F3_", "F4", "F5_", "F6_", "F7_", "E"))
MLP_CASES = gio.list_cases(("F8_",))
TOL = 2e-6


def _op_inputs(fx):
    seed = int(fx["seed"])
    values = torch.from_numpy(gio.synth(tuple(int(v) for v in fx["values_shape"]), seed))
    return (str(fx["metric"]), bool(fx["batched"]), bool(fx["self_attn"]), float(fx["locality"]),
            torch.from_numpy(fx["mesh_out"]), torch.from_numpy(fx["mesh_in"]), values,
            torch.from_numpy(fx["lmda"]), seed)


@pytest.mark.parametrize("name", OP_CASES)
def test_operator_case(name):
    fx = gio.load(name)
    metric, batched, self_attn, q, mo, mi, values, lmda, seed = _op_inputs(fx)
    lm = lmda.clone().requires_grad_(True)
    u = values.clone().requires_grad_(True)
    if self_attn:
        out = orc.posatt_self(metric, batched, mo, u, lm, q)
    else:
        out = orc.posatt_cross(metric, batched, mo, mi, u, lm, q)
    d_out = torch.from_numpy(gio.synth(tuple(out.shape), seed + 1000))
    out.backward(d_out)
    for key, got in (("out", out.detach()), ("d_values", u.grad)):
        e, g, ne, ng = gio.expect(fx, key, got.numpy())
        assert gio.rel_l2(e, g) <= TOL, key
        if ne is not None:
            assert abs(ne - ng) <= TOL * ne
    assert gio.rel_l2(fx["d_lmda"], lm.grad.numpy()) <= 1e-5

    # the selection form the HIP kernels use: unscaled order statistics + A.3 lerp
    m = orc.sqdist(metric, mo, mi)
    mk, mk1, mmin = orc.row_order_stats(m, q)
    assert np.array_equal(mk.numpy(), fx["m_k"]) and np.array_equal(mk1.numpy(), fx["m_k1"])
    assert np.array_equal(mmin.numpy(), fx["m_min"])
    c = torch.from_numpy(fx["c"])
    assert np.array_equal(orc.head_scale(lmda).numpy(), fx["c"])
    k, w = orc.quantile_rank(q, m.shape[-1])
    assert k == int(fx["rank_k"]) and np.float32(w) == fx["rank_w"]
    a = (mk.unsqueeze(1) if batched else mk.unsqueeze(0)) * c.squeeze(-1)
    b = (mk1.unsqueeze(1) if batched else mk1.unsqueeze(0)) * c.squeeze(-1)
    thr = orc.lerp_threshold(a, b, w)
    assert np.array_equal(thr.numpy(), fx["thr"])
    scaled = (m.unsqueeze(1) * c) if batched else (m * c)
    assert np.array_equal((scaled <= thr.unsqueeze(-1)).sum(-1).numpy().astype(np.int16), fx["keep_count"])
    assert torch.equal(orc.quantile_threshold_explicit(scaled, q), orc.quantile_threshold(scaled, q))


@pytest.mark.parametrize("name", MLP_CASES)
def test_mlp_case(name):
    fx = gio.load(name)
    n0, n1, n2 = (int(v) for v in fx["dims"])
    seed = int(fx["seed"])
    shapes = [("mlp1.weight", (n1, n0)), ("mlp1.bias", (n1,)), ("mlp2.weight", (n2, n1)), ("mlp2.bias", (n2,))]
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gio.synth_params(shapes, seed).items()}
    rows = tuple(int(v) for v in fx["rows"])
    x = torch.from_numpy(gio.synth(rows + (n0,), seed + 1)).requires_grad_(True)
    y = orc.mlp(x, p["mlp1.weight"], p["mlp1.bias"], p["mlp2.weight"], p["mlp2.bias"])
    y.backward(torch.from_numpy(gio.synth(tuple(y.shape), seed + 2)))
    got = {"y": y.detach(), "d_x": x.grad, "d_w1": p["mlp1.weight"].grad, "d_b1": p["mlp1.bias"].grad,
           "d_w2": p["mlp2.weight"].grad, "d_b2": p["mlp2.bias"].grad}
    for key, val in got.items():
        e, g, _, _ = gio.expect(fx, key, val.numpy())
        assert gio.rel_l2(e, g) <= TOL, key


def oracle_model_forward(cs, params):
    cfg = cs["cfg"]
    if cs["kind"] == "cloud":
        o = orc.pit_apply(params, cs["metric"], True, cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"],
                          cs["mesh_in"], cs["func_in"], cs["mesh_out"].clone(), cs["mesh_out"])
        return o.reshape(*cs["mesh_out"].shape[:-1], cfg["out_dim"])
    sd = cfg["space_dim"]
    mi, mo = cs["mesh_in"].reshape(-1, sd), cs["mesh_out"].reshape(-1, sd)
    f = orc.with_coords(mi, cs["func_in"].reshape(cs["func_in"].shape[0], -1, cfg["in_dim"]))
    o = orc.pit_apply(params, cs["metric"], False, cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"],
                      mi, f, cs["mesh_ltt"].reshape(-1, sd), mo)
    return o.reshape(cs["func_in"].shape[0], *cs["mesh_out"].shape[:-1], cfg["out_dim"])


@pytest.mark.parametrize("name", mc.CASES)
def test_model_case(name):
    fx = gio.load(name)
    cs = mc.build_case(name)
    params = {k: torch.from_numpy(v).requires_grad_(True)
              for k, v in gio.synth_params(cs["shapes"], int(fx["param_seed"])).items()}
    assert list(params.keys()) == [str(s) for s in fx["param_names"]]
    out = oracle_model_forward(cs, params)
    loss = orc.rel_lp_loss(cs["target"], out, cs["cfg"]["out_dim"], cs["p_norm"])
    loss.backward()
    e, g, _, _ = gio.expect(fx, "out", out.detach().numpy())
    assert gio.rel_l2(e, g) <= TOL
    assert abs(float(loss.detach()) - float(fx["loss"])) <= 1e-6 * abs(float(fx["loss"]))
    for k, v in params.items():
        e, g, _, _ = gio.expect(fx, "grad/" + k, v.grad.numpy())
        tol = 1e-5 if k.endswith("lmda") else 5e-6
        assert gio.rel_l2(e, g) <= tol, k
        if k.endswith("lmda"):
            assert np.array_equal(orc.head_scale(v.detach()).numpy(), fx["c/" + k])
