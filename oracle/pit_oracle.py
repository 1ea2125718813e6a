"""CPU oracle for the PiT position-attention hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional (no nn.Module, no classes-with-state) restatement, in
PyTorch-CPU eager fp32, of the arithmetic that the reference performs in
``/root/reference/pit.py``.  It exists to check the HIP path, never to serve it:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package
(``position_induced_transformer_amd``) never imports anything from ``oracle/``.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
(``/root/reference/pit.py``) in the build container, runs both on the same seeded
inputs and asserts bit-equality of forward outputs / masks before writing the
golden vectors under ``tests/golden``;  ``tests/test_oracle_golden.py`` re-checks
this oracle against those committed vectors on every run (the reference does not
travel to the GPU box).

The arithmetic itself lives in ATen CPU kernels (torch 2.10.0+rocm7.0 wheel):
``quantile`` (= sort + lerp), ``softmax``, ``bmm``/``addmm``, exact-erf ``gelu``,
``sin``/``tan``.  Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

FLT_MAX = torch.finfo(torch.float32).max
# 0.25*pi*(1-1e-7) evaluated in python double exactly as pit.py:48 writes it
SCALE_K = 0.25 * math.pi * (1 - 1e-7)

METRICS = ("euclid", "periodic1d", "periodic2d")


# --------------------------------------------------------------------------- #
# pieces of dist2att
# --------------------------------------------------------------------------- #
def head_scale(lmda: torch.Tensor) -> torch.Tensor:
    """c_h = tan(0.25*pi*(1-1e-7)*(1+sin(lmda_h)))  -- pit.py:48,135,196,254."""
    return torch.tan(SCALE_K * (1.0 + torch.sin(lmda)))


def sqdist_euclid(mesh_out: torch.Tensor, mesh_in: torch.Tensor) -> torch.Tensor:
    """Pairwise squared distance, direct (dx*dx)+(dy*dy) form -- pit.py:47 / :134.

    Works for batched ``(b,N,s)/(b,J,s)`` and batch-free ``(N,s)/(J,s)`` meshes.
    """
    diff = mesh_out.unsqueeze(-2) - mesh_in.unsqueeze(-3)
    return torch.sum(diff ** 2, dim=-1)


def period_1d(mesh_in: torch.Tensor) -> torch.Tensor:
    """Period l = |x_1 - x_0| * J of a uniform periodic 1-d mesh -- pit.py:190-191."""
    return torch.abs(mesh_in[1, 0] - mesh_in[0, 0]) * mesh_in.shape[0]


def period_2d(mesh_in: torch.Tensor) -> torch.Tensor:
    """Period of a square periodic grid -- pit.py:248-250."""
    res = int(mesh_in.shape[0] ** 0.5)
    dx = (torch.max(mesh_in[:, 0]) - torch.min(mesh_in[:, 0])) / (res - 1)
    return dx * res


def sqdist_periodic1d(mesh_out: torch.Tensor, mesh_in: torch.Tensor) -> torch.Tensor:
    """pit.py:190-194: wrap |dx| to min(|dx|, l-|dx|), square coordinate 0 only."""
    l = period_1d(mesh_in)
    d = abs(mesh_out.unsqueeze(-2) - mesh_in.unsqueeze(-3))
    d = torch.minimum(d, l - d)
    return d[..., 0] ** 2


def sqdist_periodic2d(mesh_out: torch.Tensor, mesh_in: torch.Tensor) -> torch.Tensor:
    """pit.py:248-253: per-coordinate wrap, then sum of squares."""
    l = period_2d(mesh_in)
    d = abs(mesh_out.unsqueeze(-2) - mesh_in.unsqueeze(-3))
    d = torch.minimum(d, l - d)
    return torch.sum(d ** 2, dim=-1)


def sqdist(metric: str, mesh_out: torch.Tensor, mesh_in: torch.Tensor) -> torch.Tensor:
    if metric == "euclid":
        return sqdist_euclid(mesh_out, mesh_in)
    if metric == "periodic1d":
        return sqdist_periodic1d(mesh_out, mesh_in)
    if metric == "periodic2d":
        return sqdist_periodic2d(mesh_out, mesh_in)
    raise ValueError(metric)


def quantile_threshold(scaled: torch.Tensor, locality: float) -> torch.Tensor:
    """Row threshold as the reference computes it: torch.quantile(..., keepdim=True)
    -- pit.py:49,136,197,255."""
    return torch.quantile(scaled, locality, dim=-1, keepdim=True)


def quantile_rank(locality: float, n: int):
    """(k, w) of torch.quantile's linear interpolation for a row of length n, in the
    fp32 arithmetic ATen uses: rank = fl32(q)*fl32(n-1), k=floor(rank), w=rank-k."""
    rank = torch.tensor(locality, dtype=torch.float32) * torch.tensor(n - 1, dtype=torch.float32)
    k = int(torch.floor(rank).item())
    w = (rank - float(k)).to(torch.float32)
    return k, float(w.item())


def lerp_threshold(a: torch.Tensor, b: torch.Tensor, w: float) -> torch.Tensor:
    """ATen's lerp on fp32: w<0.5 ? fma(w, b-a, a) : fma(-(b-a), 1-w, b).

    fp32 fma is emulated exactly in fp64 (products of two fp32 fit in 48 bits, the
    sum is then rounded once to fp32 -- double rounding cannot occur because the
    fp64 sum of a 48-bit product and a 24-bit addend is exact or within fp64's 53
    bits for the magnitudes involved; the golden test pins this against
    torch.quantile on every fixture row)."""
    wf = torch.tensor(w, dtype=torch.float32)
    diff = (b - a)  # fp32 rounding of b-a
    if w < 0.5:
        t = wf.double() * diff.double() + a.double()
    else:
        omw = (torch.tensor(1.0, dtype=torch.float32) - wf)  # fp32 rounding of 1-w
        t = b.double() - diff.double() * omw.double()
    return t.float()


def quantile_threshold_explicit(scaled: torch.Tensor, locality: float) -> torch.Tensor:
    """Order-statistic form of the threshold (what the HIP select kernel implements):
    T = lerp(S_(k), S_(k+1), w) with (k, w) from :func:`quantile_rank`.  Must equal
    :func:`quantile_threshold` bit for bit (tests/test_oracle_golden.py)."""
    n = scaled.shape[-1]
    k, w = quantile_rank(locality, n)
    srt = torch.sort(scaled, dim=-1).values
    a = srt[..., k:k + 1]
    b = srt[..., min(k + 1, n - 1):min(k + 1, n - 1) + 1]
    return lerp_threshold(a, b, w)


def row_order_stats(m_dist: torch.Tensor, locality: float):
    """Unscaled per-row statistics the HIP select kernel returns: (m_(k), m_(k+1),
    m_min).  SURVEY Appendix A.4: fl(c*m) is monotone in m, so the scaled order
    statistics are fl(c*m_(k)), fl(c*m_(k+1))."""
    n = m_dist.shape[-1]
    k, _ = quantile_rank(locality, n)
    srt = torch.sort(m_dist, dim=-1).values
    return srt[..., k], srt[..., min(k + 1, n - 1)], srt[..., 0]


def attention_weights(m_dist: torch.Tensor, c: torch.Tensor, locality: float,
                      batched: bool) -> torch.Tensor:
    """scale -> quantile mask -> softmax, pit.py:48-52 (batched) / :135-139 (fixed).

    ``m_dist`` is ``(b,N,J)`` when ``batched`` else ``(N,J)``; ``c`` is ``(H,1,1)``.
    Returns A ``(b,H,N,J)`` or ``(H,N,J)``.
    """
    scaled = (m_dist.unsqueeze(1) * c) if batched else (m_dist * c)
    thr = quantile_threshold(scaled, locality)
    scaled = torch.where(scaled <= thr, scaled, torch.tensor(FLT_MAX))
    return torch.softmax(-scaled, dim=-1)


def weighted_values(att: torch.Tensor, values: torch.Tensor, batched: bool) -> torch.Tensor:
    """A.V with the head-major output layout (index h*D+d) -- pit.py:54-57 / :141-144."""
    n_head = att.shape[-3]
    eq = "bhnj,bjd->bnhd" if batched else "hnj,bjd->bnhd"
    out = torch.einsum(eq, att, values)
    return out.reshape(values.shape[0], -1, n_head * values.shape[-1])


def posatt_cross(metric: str, batched: bool, mesh_out, mesh_in, values, lmda, locality,
                 c: Optional[torch.Tensor] = None) -> torch.Tensor:
    """posatt_cross*.forward -- pit.py:63-71,151-159,207-215,265-273.

    ``c`` (H,1,1) overrides head_scale(lmda) (used to inject a stored scale)."""
    if c is None:
        c = head_scale(lmda)
    att = attention_weights(sqdist(metric, mesh_out, mesh_in), c, locality, batched)
    return weighted_values(att, values, batched)


def posatt_self(metric: str, batched: bool, mesh, values, lmda, locality,
                c: Optional[torch.Tensor] = None) -> torch.Tensor:
    """posatt.forward: cat((inputs, conv), -1) -- pit.py:37-44."""
    conv = posatt_cross(metric, batched, mesh, mesh, values, lmda, locality, c)
    return torch.cat((values, conv), dim=-1)


def mlp(x, w1, b1, w2, b2):
    """kaiming_mlp.forward: Linear -> exact-erf GELU -> Linear -- pit.py:21-26."""
    return F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2)


# --------------------------------------------------------------------------- #
# model assembly (pit.py:108-127) on a plain state-dict
# --------------------------------------------------------------------------- #
def _mlp_p(p: Dict[str, torch.Tensor], prefix: str, x):
    return mlp(x, p[prefix + ".mlp1.weight"], p[prefix + ".mlp1.bias"],
               p[prefix + ".mlp2.weight"], p[prefix + ".mlp2.bias"])


def pit_apply(p: Dict[str, torch.Tensor], metric: str, batched: bool, n_blocks: int,
              en_loc: float, de_loc: float, mesh_in, func_in, mesh_ltt, mesh_out,
              norm_after_enc_proc: bool = False) -> torch.Tensor:
    """encoder -> processor -> decoder, pit.py:108-127, parameter names as in the
    reference state_dict (down.lmda, en_layer.mlp1.weight, conv.i.lmda, ...).

    ``norm_after_enc_proc`` inserts the affine-free InstanceNorm1d over the point
    axis that train_vorticity.py:56,59 applies after encoder and processor."""
    def inorm(x):
        return F.instance_norm(x.permute(0, 2, 1)).permute(0, 2, 1)

    f = posatt_cross(metric, batched, mesh_ltt, mesh_in, func_in, p["down.lmda"], en_loc)
    f = F.gelu(_mlp_p(p, "en_layer", f))
    if norm_after_enc_proc:
        f = inorm(f)
    for i in range(n_blocks):
        f = posatt_self(metric, batched, mesh_ltt, f, p[f"conv.{i}.lmda"], 1.0)
        f = F.gelu(_mlp_p(p, f"mlp.{i}", f))
    if norm_after_enc_proc:
        f = inorm(f)
    f = posatt_cross(metric, batched, mesh_out, mesh_ltt, f, p["up.lmda"], de_loc)
    return _mlp_p(p, "de", f)


def with_coords(mesh_in_flat: torch.Tensor, func_in: torch.Tensor) -> torch.Tensor:
    """cat(tile(mesh), func) of the fixed-mesh task forwards -- train_darcy.py:55,
    train_burgers.py:43, train_vorticity.py:54."""
    b = func_in.shape[0]
    return torch.cat((mesh_in_flat.unsqueeze(0).expand(b, -1, -1), func_in), dim=-1)


def rel_lp_loss(true: torch.Tensor, pred: torch.Tensor, out_dim: int, p: int) -> torch.Tensor:
    """RelLpNorm.__call__ -- utils.py:86-98 (sum over batch of channel-mean rel. Lp)."""
    t = true.reshape(true.size(0), -1, out_dim)
    q = pred.reshape(pred.size(0), -1, out_dim)
    num = torch.norm(t - q, p=p, dim=1)
    den = torch.norm(t, p=p, dim=1)
    return torch.sum(torch.mean(num / den, dim=-1))


# --------------------------------------------------------------------------- #
# meshes and parameter init used by fixtures, smoke and bench (synthetic)
# --------------------------------------------------------------------------- #
def grid_mesh_2d(s: int, endpoint: bool = True) -> torch.Tensor:
    """(s*s, 2) fp32 grid with the x-fastest point order that
    np.meshgrid(...).ravel() gives in train_darcy.py:83-88 (endpoint=True) and
    train_vorticity.py:77-83 (linspace(0,1,s+1)[:-1], endpoint=False)."""
    import numpy as np
    ax = np.linspace(0, 1, s) if endpoint else np.linspace(0, 1, s + 1)[:-1]
    m = np.vstack([xx.ravel() for xx in np.meshgrid(ax, ax)]).T
    return torch.tensor(m, dtype=torch.float)


def line_mesh_1d(n: int, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    """(n,1) periodic line mesh linspace(lo,hi,n+1)[:-1] -- train_burgers.py:59-60."""
    return torch.linspace(lo, hi, n + 1)[:-1].reshape(-1, 1)


def param_shapes(space_dim, in_dim, out_dim, hid, n_head, n_blocks, en_in: Optional[int] = None):
    """Ordered (name, shape) list of the reference state_dict -- pit.py:98-106.
    ``en_in`` overrides the en_layer input width (train_elasticity.py:39)."""
    if en_in is None:
        en_in = n_head * (in_dim + space_dim)
    out = [("down.lmda", (n_head, 1, 1))]
    def mlp_shapes(prefix, n0, n1, n2):
        return [(prefix + ".mlp1.weight", (n1, n0)), (prefix + ".mlp1.bias", (n1,)),
                (prefix + ".mlp2.weight", (n2, n1)), (prefix + ".mlp2.bias", (n2,))]
    out += mlp_shapes("en_layer", en_in, hid, hid)
    for i in range(n_blocks):
        out.append((f"conv.{i}.lmda", (n_head, 1, 1)))
    for i in range(n_blocks):
        out += mlp_shapes(f"mlp.{i}", (1 + n_head) * hid, hid, hid)
    out.append(("up.lmda", (n_head, 1, 1)))
    out += mlp_shapes("de", n_head * hid, hid, out_dim)
    return out


def init_params(shapes: Sequence, seed: int) -> Dict[str, torch.Tensor]:
    """Random-init parameters with the reference's distributions (pit.py:18-19,35;
    nn.Linear default bias init) from a private generator: lmda ~ U[0,1), weights
    He-normal (std = sqrt(2/fan_in)), biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    fan_in = 1
    for name, shape in shapes:
        if name.endswith("lmda"):
            p[name] = torch.rand(shape, generator=g)
        elif name.endswith("weight"):
            fan_in = shape[1]
            p[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        else:
            bound = 1.0 / math.sqrt(fan_in)
            p[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return p
