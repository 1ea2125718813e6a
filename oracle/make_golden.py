"""Generate tests/golden/*.npz by importing the reference (build container only).

Run:  python oracle/make_golden.py            (needs /root/reference; CPU only)

For every case the REFERENCE (``/root/reference/pit.py`` imported unmodified) and the
oracle (``oracle/pit_oracle.py``) are run on the same seeded inputs; the script
asserts bit-equality of their forward outputs and attention weights (so the oracle
is pinned to the reference) and then stores inputs + expected outputs as data.
Nothing of the reference's source travels: fixtures hold arrays only.

Fixture list follows SURVEY.md section 8(c): F1..F8 operator level, F9/F10/F11 model
level, E* edge cases.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io as gio      # noqa: E402  synthetic inputs + compact storage
import pit as ref            # noqa: E402  the reference, read-only
import utils as ref_utils    # noqa: E402
import pit_oracle as orc     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)

REF_CROSS = {("euclid", True): ref.posatt_cross, ("euclid", False): ref.posatt_cross_fixed,
             ("periodic1d", False): ref.posatt_cross_periodic1d,
             ("periodic2d", False): ref.posatt_cross_periodic2d}
REF_SELF = {("euclid", True): ref.posatt, ("euclid", False): ref.posatt_fixed,
            ("periodic1d", False): ref.posatt_periodic1d,
            ("periodic2d", False): ref.posatt_periodic2d}


def npf(t):
    return t.detach().cpu().numpy()


def cr_head_scale(lmda: torch.Tensor) -> torch.Tensor:
    """The HIP kernels evaluate lmda->c through fp64 (correctly rounded sin/tan, fp32
    intermediate roundings as in pit.py:48).  Used to pick model-level seeds where this
    agrees bitwise with ATen's (sleef, <=1 ulp) result, see DESIGN.md 'head scale'."""
    s = torch.sin(lmda.double()).float()
    u = torch.tensor(orc.SCALE_K).float() * (1.0 + s)
    return torch.tan(u.double()).float()


def op_case(name, metric, batched, self_attn, mesh_out, mesh_in, vshape, lmda, q, seed):
    """Run one operator case through reference and oracle; save fixture.
    values = synth(vshape, seed), d_out = synth(out.shape, seed+1000) (not stored)."""
    n_head = lmda.shape[0]
    values = torch.from_numpy(gio.synth(vshape, seed))
    d = values.shape[-1]
    cls = (REF_SELF if self_attn else REF_CROSS)[(metric, batched)]
    mod = cls(n_head, d, q)
    with torch.no_grad():
        mod.lmda.copy_(lmda)
    u_ref = values.clone().requires_grad_(True)
    if self_attn:
        out_ref = mod(mesh_out, u_ref)
    else:
        out_ref = mod(mesh_out, mesh_in, u_ref)
    d_out = torch.from_numpy(gio.synth(tuple(out_ref.shape), seed + 1000))
    out_ref.backward(d_out)
    att_ref = mod.dist2att(mesh_out, mesh_in, mod.lmda, q).detach()

    # oracle, lmda path
    lm = lmda.clone().requires_grad_(True)
    u_o = values.clone().requires_grad_(True)
    if self_attn:
        out_o = orc.posatt_self(metric, batched, mesh_out, u_o, lm, q)
    else:
        out_o = orc.posatt_cross(metric, batched, mesh_out, mesh_in, u_o, lm, q)
    out_o.backward(d_out)
    assert torch.equal(out_o, out_ref), f"{name}: oracle forward != reference"
    assert torch.equal(u_o.grad, u_ref.grad), f"{name}: oracle dU != reference"
    assert torch.equal(lm.grad, mod.lmda.grad), f"{name}: oracle dlmda != reference"

    # oracle, injected-c path (gives dc for the kernel-level tests)
    c = orc.head_scale(lmda).detach()
    c_leaf = c.clone().requires_grad_(True)
    u_c = values.clone().requires_grad_(True)
    if self_attn:
        out_c = orc.posatt_self(metric, batched, mesh_out, u_c, None, q, c=c_leaf)
    else:
        out_c = orc.posatt_cross(metric, batched, mesh_out, mesh_in, u_c, None, q, c=c_leaf)
    assert torch.equal(out_c, out_ref)
    out_c.backward(d_out)

    m = orc.sqdist(metric, mesh_out, mesh_in)
    scaled = (m.unsqueeze(1) * c) if batched else (m * c)
    thr = orc.quantile_threshold(scaled, q)
    thr_x = orc.quantile_threshold_explicit(scaled, q)
    assert torch.equal(thr, thr_x), f"{name}: explicit order-statistic threshold != torch.quantile"
    keep = scaled <= thr
    att_o = orc.attention_weights(m, c, q, batched)
    assert torch.equal(att_o, att_ref), f"{name}: oracle attention != reference"
    assert torch.equal(att_ref > 0, keep), f"{name}: keep-set inconsistent"
    mk, mk1, mmin = orc.row_order_stats(m, q)
    # A.4: scaled order statistics are the scaled unscaled ones
    k, w = orc.quantile_rank(q, m.shape[-1])
    a = (mk.unsqueeze(1) if batched else mk.unsqueeze(0)) * c.squeeze(-1)   # ((b,)H,N)
    b = (mk1.unsqueeze(1) if batched else mk1.unsqueeze(0)) * c.squeeze(-1)
    assert torch.equal(orc.lerp_threshold(a, b, w).unsqueeze(-1), thr), f"{name}: A.4 failed"

    period = 0.0
    if metric == "periodic1d":
        period = float(orc.period_1d(mesh_in))
    elif metric == "periodic2d":
        period = float(orc.period_2d(mesh_in))

    st = dict(metric=metric, batched=batched, self_attn=self_attn, locality=np.float64(q),
              period=np.float32(period), rank_k=np.int32(k), rank_w=np.float32(w),
              seed=np.int64(seed), values_shape=np.asarray(vshape, dtype=np.int64),
              mesh_out=npf(mesh_out), mesh_in=npf(mesh_in), lmda=npf(lmda), c=npf(c),
              m_k=npf(mk), m_k1=npf(mk1), m_min=npf(mmin), thr=npf(thr.squeeze(-1)),
              keep_count=npf(keep.sum(-1)).astype(np.int16),
              d_lmda=npf(mod.lmda.grad), d_c=npf(c_leaf.grad))
    gio.pack(st, "out", npf(out_ref))
    gio.pack(st, "d_values", npf(u_ref.grad))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **st)
    kc = keep.sum(-1)
    print(f"{name:28s} out{tuple(out_ref.shape)} keep/row {int(kc.min())}-{int(kc.max())} "
          f"k={k} w={w:.4f} c={npf(c).ravel()}")


def rnd(shape, seed, lo=None, hi=None):
    return torch.from_numpy(gio.synth(shape, seed, lo, hi))


def make_op_cases():
    grid43, grid16 = orc.grid_mesh_2d(43), orc.grid_mesh_2d(16)
    lm2 = torch.tensor([0.3123, 0.8765]).reshape(2, 1, 1)
    lm1 = torch.tensor([0.5531]).reshape(1, 1, 1)
    # F1 Darcy encoder, F2 processor, F3 decoder (train_darcy.py shapes)
    op_case("F1_darcy_enc", "euclid", False, False, grid16, grid43, (2, 1849, 3), lm2, 0.02, 1)
    op_case("F2_darcy_proc", "euclid", False, True, grid16, grid16, (2, 256, 64), lm2, 1.0, 2)
    op_case("F3_darcy_dec", "euclid", False, False, grid43, grid16, (2, 256, 64), lm2, 0.02, 3)
    # F4 batched random cloud cross + F4s self (train_elasticity.py shapes, narrower values)
    cloud = rnd((2, 972, 2), 4, 0.0, 1.0)
    op_case("F4_cloud_cross", "euclid", True, False, cloud, cloud, (2, 972, 12), lm2, 0.02, 5)
    cl2 = rnd((2, 300, 2), 6, 0.0, 1.0)
    op_case("F4s_cloud_self", "euclid", True, True, cl2, cl2, (2, 300, 64), lm2, 1.0, 7)
    # F5 periodic 1d (train_burgers.py shapes)
    l1024, l256 = orc.line_mesh_1d(1024), orc.line_mesh_1d(256)
    op_case("F5_p1d_enc", "periodic1d", False, False, l256, l1024, (2, 1024, 2), lm2, 0.02, 8)
    op_case("F5_p1d_dec", "periodic1d", False, False, l1024, l256, (2, 256, 64), lm2, 0.02, 9)
    op_case("F5_p1d_proc", "periodic1d", False, True, l256, l256, (2, 256, 64), lm2, 1.0, 10)
    # F6 periodic 2d (train_vorticity.py shapes)
    p64, p16 = orc.grid_mesh_2d(64, endpoint=False), orc.grid_mesh_2d(16, endpoint=False)
    op_case("F6_p2d_enc", "periodic2d", False, False, p16, p64, (2, 4096, 12), lm2, 0.02, 11)
    op_case("F6_p2d_dec", "periodic2d", False, False, p64, p16, (1, 256, 32), lm2, 0.02, 12)
    op_case("F6_p2d_proc", "periodic2d", False, True, p16, p16, (2, 256, 32), lm2, 1.0, 13)
    # F7 NACA-like: small J with fractional rank 2.38, one head, and a tall slice
    a_in = rnd((2, 120, 2), 14, -0.5, 0.5)
    a_ltt = rnd((2, 728, 2), 15, -1.0, 1.0)
    op_case("F7_naca_enc", "euclid", True, False, a_ltt, a_in, (2, 120, 2), lm1, 0.02, 16)
    a_out = rnd((1, 2048, 2), 17, -1.0, 1.0)
    op_case("F7_naca_dec", "euclid", True, False, a_out, a_ltt[:1], (1, 728, 32), lm1, 0.02, 18)
    # ---- edge cases
    # E1: scale < 1 (lmda ~ -1.2)
    op_case("E1_small_scale", "euclid", False, False, grid16, grid43, (1, 1849, 5),
            torch.tensor([-1.2, -0.4]).reshape(2, 1, 1), 0.02, 19)
    # E2: duplicated points (exact ties in distance), batched
    dup = rnd((1, 40, 2), 20, 0.0, 1.0)
    dup = torch.cat((dup, dup, dup[:, :20]), dim=1)            # 100 points, many duplicates
    op_case("E2_duplicates", "euclid", True, False, dup[:, :64].contiguous(), dup, (1, 100, 7), lm2, 0.05, 21)
    # E3: J = 2
    op_case("E3_J2", "euclid", True, False, rnd((2, 33, 2), 22, 0, 1), rnd((2, 2, 2), 23, 0, 1),
            (2, 2, 4), lm2, 0.02, 24)
    # E4: locality 1.0 on a cross attention (nothing masked, row min not zero)
    op_case("E4_loc1_cross", "euclid", False, False, grid16, grid43[:500].contiguous(), (2, 500, 9), lm2, 1.0, 25)
    # E5: ragged sizes (N, J, D not multiples of any tile), 3 heads, mid locality
    op_case("E5_ragged", "euclid", True, False, rnd((3, 77, 2), 26, 0, 1), rnd((3, 131, 2), 27, 0, 1),
            (3, 131, 37), torch.tensor([0.1, 0.5, 0.9]).reshape(3, 1, 1), 0.3, 28)
    # E6: 1-d euclidean fixed mesh (Sod-like, train_sod.py:61-62) on [-5,5)
    s_in, s_ltt = orc.line_mesh_1d(512, -5.0, 5.0), orc.line_mesh_1d(256, -5.0, 5.0)
    op_case("E6_sod_enc", "euclid", False, False, s_ltt, s_in, (2, 512, 4), lm1, 0.02, 29)
    # E7: 3-d point cloud (space_dim 3 is legal for pit.py:47), batched
    op_case("E7_cloud3d", "euclid", True, False, rnd((2, 90, 3), 30, 0, 1), rnd((2, 150, 3), 31, 0, 1),
            (2, 150, 8), lm2, 0.1, 32)


def mlp_case(name, n0, n1, n2, rows, seed):
    """kaiming_mlp fwd/bwd; parameters = synth_params(seed), x = synth(seed+1),
    d_y = synth(seed+2)."""
    shapes = [("mlp1.weight", (n1, n0)), ("mlp1.bias", (n1,)), ("mlp2.weight", (n2, n1)), ("mlp2.bias", (n2,))]
    p = {k: torch.from_numpy(v) for k, v in gio.synth_params(shapes, seed).items()}
    mod = ref.kaiming_mlp(n0, n1, n2)
    mod.load_state_dict(p)
    x = rnd(rows + (n0,), seed + 1).requires_grad_(True)
    y = mod(x)
    dy = rnd(tuple(y.shape), seed + 2)
    y.backward(dy)
    x2 = x.detach().clone().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    y2 = orc.mlp(x2, ps["mlp1.weight"], ps["mlp1.bias"], ps["mlp2.weight"], ps["mlp2.bias"])
    assert torch.equal(y2, y), f"{name}: oracle mlp != reference"
    y2.backward(dy)
    assert torch.equal(x2.grad, x.grad)
    st = dict(seed=np.int64(seed), dims=np.asarray([n0, n1, n2], dtype=np.int64),
              rows=np.asarray(rows, dtype=np.int64))
    gio.pack(st, "y", npf(y))
    gio.pack(st, "d_x", npf(x.grad))
    gio.pack(st, "d_w1", npf(mod.mlp1.weight.grad))
    gio.pack(st, "d_b1", npf(mod.mlp1.bias.grad))
    gio.pack(st, "d_w2", npf(mod.mlp2.weight.grad))
    gio.pack(st, "d_b2", npf(mod.mlp2.bias.grad))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **st)
    print(f"{name:28s} x{tuple(x.shape)} -> y{tuple(y.shape)}")


# ----------------------------------------------------------------------------- models
def fixed_task_forward(model, mesh_in, func_in, mesh_out):
    """Shape handling of the fixed-mesh task forwards (train_darcy.py:46-59 and the
    same lines of train_burgers/vorticity): flatten, prepend coordinates, enc/proc/dec."""
    sd, b = model.space_dim, func_in.shape[0]
    size = mesh_out.shape[:-1]
    mi, mo = mesh_in.reshape(-1, sd), mesh_out.reshape(-1, sd)
    f = func_in.reshape(b, -1, model.in_dim)
    f = torch.cat((mi.unsqueeze(0).repeat(b, 1, 1), f), -1)
    z = model.encoder(mi, f, model.mesh_ltt)
    z = model.processor(z, model.mesh_ltt)
    return model.decoder(model.mesh_ltt, z, mo).reshape(b, *size, model.out_dim)


def cloud_task_forward(model, mesh_in, func_in, mesh_out):
    """train_elasticity.py:41-54: latent mesh = output mesh (per sample)."""
    ltt = mesh_out.clone()
    z = model.encoder(mesh_in, func_in, ltt)
    z = model.processor(z, ltt)
    return model.decoder(ltt, z, mesh_out).reshape(*mesh_out.shape[:-1], model.out_dim)


def pick_param_seed(shapes, lmda_names, start):
    """First synth_params seed whose lmda->c agrees bitwise between ATen CPU (sleef)
    and the fp64 route the HIP kernels take (DESIGN.md 'head scale')."""
    for seed in range(start, start + 500):
        p = {k: torch.from_numpy(v) for k, v in gio.synth_params(shapes, seed).items()}
        if all(torch.equal(orc.head_scale(p[n]), cr_head_scale(p[n])) for n in lmda_names):
            return seed, p
    raise RuntimeError("no seed found")


def model_case(name, build, fwd, metric, batched, shapes, inputs, target, out_dim, p_norm, n_blocks,
               en_loc, de_loc, seed0):
    lm_names = ["down.lmda"] + [f"conv.{i}.lmda" for i in range(n_blocks)] + ["up.lmda"]
    seed, sd = pick_param_seed(shapes, lm_names, seed0)
    model = build()
    model.load_state_dict(sd)
    assert [k for k, _ in shapes] == list(model.state_dict().keys()), "state_dict key order drifted"
    mesh_in, func_in, mesh_out = inputs
    out = fwd(model, mesh_in, func_in, mesh_out)
    loss = ref_utils.RelLpNorm(out_dim, p_norm)(target, out)
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in model.named_parameters()}

    # oracle on the same state dict
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    if batched:
        o = orc.pit_apply(p, metric, True, n_blocks, en_loc, de_loc, mesh_in, func_in,
                          mesh_out.clone(), mesh_out).reshape(*mesh_out.shape[:-1], out_dim)
    else:
        sdim = model.space_dim
        mi, mo = mesh_in.reshape(-1, sdim), mesh_out.reshape(-1, sdim)
        f = orc.with_coords(mi, func_in.reshape(func_in.shape[0], -1, model.in_dim))
        o = orc.pit_apply(p, metric, False, n_blocks, en_loc, de_loc, mi, f, model.mesh_ltt, mo)
        o = o.reshape(func_in.shape[0], *mesh_out.shape[:-1], out_dim)
    assert torch.equal(o, out), f"{name}: oracle model forward != reference"
    lo = orc.rel_lp_loss(target, o, out_dim, p_norm)
    assert torch.equal(lo, loss)
    lo.backward()
    for k in grads:
        assert torch.equal(p[k].grad, grads[k]), f"{name}: oracle grad {k} != reference"

    st = {"param_seed": np.int64(seed), "loss": npf(loss),
          "param_names": np.asarray([k for k, _ in shapes])}
    gio.pack(st, "out", npf(out))
    for k in lm_names:
        st["c/" + k] = npf(orc.head_scale(sd[k]))
    for k, gq in grads.items():
        gio.pack(st, "grad/" + k, npf(gq))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **st)
    print(f"{name:28s} param_seed={seed} out{tuple(out.shape)} loss={float(loss.detach()):.6f} "
          f"params={sum(v.numel() for v in sd.values())}")


# Model-level case definitions are shared with the tests (tests/model_cases.py) so both
# sides build identical inputs.
def make_model_cases():
    import model_cases as mc
    for name in mc.CASES:
        cs = mc.build_case(name)
        cfg = cs["cfg"]
        if cs["kind"] == "fixed":
            cls = {"euclid": ref.pit_fixed, "periodic1d": ref.pit_periodic1d,
                   "periodic2d": ref.pit_periodic2d}[cs["metric"]]
            ltt = cs["mesh_ltt"]
            build = lambda: cls(cfg["space_dim"], cfg["in_dim"], cfg["out_dim"], cfg["hid_dim"],  # noqa: E731
                                cfg["n_head"], cfg["n_blocks"], ltt, cfg["en_loc"], cfg["de_loc"])
            fwd = fixed_task_forward
        else:
            def build():
                m = ref.pit(cfg["space_dim"], cfg["in_dim"], cfg["out_dim"], cfg["hid_dim"], cfg["n_head"],
                            cfg["n_blocks"], None, cfg["en_loc"], cfg["de_loc"])
                m.en_layer = ref.kaiming_mlp(cfg["n_head"] * cfg["in_dim"], cfg["hid_dim"], cfg["hid_dim"])
                return m
            fwd = cloud_task_forward
        model_case(name, build, fwd, cs["metric"], cs["kind"] == "cloud", cs["shapes"],
                   (cs["mesh_in"], cs["func_in"], cs["mesh_out"]), cs["target"], cfg["out_dim"],
                   cs["p_norm"], cfg["n_blocks"], cfg["en_loc"], cfg["de_loc"], 0)


def init_case():
    """Seed-for-seed initialisation of the reference (pit.py:35 torch.rand, :18-19 kaiming_normal_,
    nn.Linear default biases, and the extra draws of pit_fixed re-creating down/conv/up,
    pit.py:182-184): digest of every parameter for torch.manual_seed(0)."""
    st = {}
    for tag, build in (("fixed", lambda: ref.pit_fixed(2, 1, 1, 64, 2, 4, orc.grid_mesh_2d(16), 0.02, 0.02)),
                       ("base", lambda: ref.pit(2, 12, 1, 32, 2, 2, None, 0.05, 0.05)),
                       ("p1d", lambda: ref.pit_periodic1d(1, 1, 1, 32, 2, 3, orc.line_mesh_1d(64), 0.02, 0.02))):
        torch.manual_seed(0)
        sd = build().state_dict()
        st[tag + "/names"] = np.asarray(list(sd.keys()))
        for k, v in sd.items():
            st[f"{tag}/head/{k}"] = npf(v.flatten()[:4])
            st[f"{tag}/sum/{k}"] = np.float64(v.double().sum().item())
            st[f"{tag}/shape/{k}"] = np.asarray(v.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "F13_init_parity.npz"), **st)
    print("F13_init_parity              ", len(st), "entries")


if __name__ == "__main__":
    for f in os.listdir(OUT):
        if f.endswith(".npz"):
            os.remove(os.path.join(OUT, f))
    make_op_cases()
    mlp_case("F8_mlp_192_64_64", 192, 64, 64, (2, 256), 201)
    mlp_case("F8_mlp_768_256_256", 768, 256, 256, (1, 100), 202)
    mlp_case("F8_mlp_128_64_1", 128, 64, 1, (2, 333), 203)
    mlp_case("F8_mlp_6_64_64", 6, 64, 64, (3, 50), 204)
    mlp_case("F8_mlp_88_256_3", 88, 256, 3, (1, 77), 205)
    make_model_cases()
    init_case()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"total golden bytes: {tot}")
