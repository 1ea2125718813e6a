"""Task models: the ``forward`` wrappers the reference defines inside each train_*.py
(SURVEY section 8(f)-1), with the scripts' hyper-parameters as named constructors and
synthetic-data generators of the right shapes (the datasets are not in the reference
repository: supplementary_data/*.mat are LFS stubs).

All wrappers only reshape / concatenate and call encoder -> processor -> decoder of
:mod:`position_induced_transformer_amd.pit`; the arithmetic is in the HIP kernels.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import pit as P


# ----------------------------------------------------------------------------- meshes
def grid_mesh_2d(s: int, endpoint: bool = True, device=None) -> torch.Tensor:
    """(s, s, 2) grid in the point order of train_darcy.py:83-88 (endpoint) /
    train_vorticity.py:77-83 (periodic, linspace(0,1,s+1)[:-1])."""
    ax = np.linspace(0, 1, s) if endpoint else np.linspace(0, 1, s + 1)[:-1]
    m = np.vstack([xx.ravel() for xx in np.meshgrid(ax, ax)]).T.reshape(s, s, 2)
    return torch.tensor(m, dtype=torch.float, device=device)


def line_mesh_1d(n: int, lo: float = 0.0, hi: float = 1.0, device=None) -> torch.Tensor:
    """(n, 1) mesh linspace(lo,hi,n+1)[:-1] (train_burgers.py:59-60, train_sod.py:61-62)."""
    return torch.linspace(lo, hi, n + 1)[:-1].reshape(-1, 1).to(device)


# ----------------------------------------------------------------------------- wrappers
class _FixedMeshForward:
    """forward(mesh_in, func_in, mesh_out) of the fixed-mesh tasks: flatten, prepend the
    coordinates to the input function, encoder -> processor -> decoder
    (train_darcy.py:46-59, train_burgers.py:38-49, train_sod.py:42-53)."""

    residual = False

    def forward(self, mesh_in, func_in, mesh_out):
        size = mesh_out.shape[:-1]
        batch = func_in.shape[0]
        mesh_in = mesh_in.reshape(-1, self.space_dim)
        mesh_out = mesh_out.reshape(-1, self.space_dim)
        func = func_in.reshape(batch, -1, self.in_dim)
        if isinstance(self.down, P.posatt) and func.is_cuda:
            # the coordinate concat of train_darcy.py:51-55 is deferred: the encoder's candidate-list kernels read the
            # coordinate channels from mesh_in (ops.tag_coords); any other consumer materialises the concat
            feats = ops.tag_coords(func, mesh_in)
        else:
            feats = torch.cat((mesh_in.unsqueeze(0).expand(batch, -1, -1), func), -1)
        ltt = self.encoder(mesh_in, feats, self.mesh_ltt)
        ltt = self._between(ltt)
        ltt = self.processor(ltt, self.mesh_ltt)
        ltt = self._between(ltt)
        out = self.decoder(self.mesh_ltt, ltt, mesh_out).reshape(batch, *size, self.out_dim)
        if self.residual:
            out = out + func_in.reshape(out.shape)
        return out

    def _between(self, x):
        return x


class pit_darcy(_FixedMeshForward, P.pit_fixed):
    """train_darcy.py:25-59."""


class pit_sod(_FixedMeshForward, P.pit_fixed):
    """train_sod.py:23-53."""


class pit_burgers(_FixedMeshForward, P.pit_periodic1d):
    """train_burgers.py:19-49."""


class pit_cylinder(_FixedMeshForward, P.pit_fixed):
    """train_cylinder.py:18-52: adds the input back onto the prediction."""
    residual = True


class pit_vorticity(_FixedMeshForward, P.pit_periodic2d):
    """train_vorticity.py:23-62: InstanceNorm1d over the points after encoder and processor."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.norm = nn.InstanceNorm1d(self.hid_dim)

    def _between(self, x):
        n = self.norm
        if x.is_cuda and isinstance(n, nn.InstanceNorm1d) and not n.affine and not n.track_running_stats:
            return ops.instance_norm_points(x, n.eps)          # HIP kernel on the (b, L, C) layout, no permutes
        return n(x.permute(0, 2, 1)).permute(0, 2, 1)


class pit_elasticity(P.pit):
    """train_elasticity.py:18-54: per-sample point cloud, latent mesh = output mesh."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.en_layer = P.kaiming_mlp(self.n_head * self.in_dim, self.hid_dim, self.hid_dim)

    def forward(self, mesh_in, func_in, mesh_out):
        size = mesh_out.shape[:-1]
        mesh_ltt = mesh_out
        ltt = self.encoder(mesh_in, func_in, mesh_ltt)
        ltt = self.processor(ltt, mesh_ltt)
        return self.decoder(mesh_ltt, ltt, mesh_out).reshape(*size, self.out_dim)


class pit_naca(P.pit):
    """train_naca.py:17-65: latent mesh = strided sub-grid of the body-fitted output grid."""

    def __init__(self, space_dim, in_dim, out_dim, hid_dim, n_head, n_blocks, mesh_ltt, x_downsample,
                 y_downsample, en_loc, de_loc):
        super().__init__(space_dim, in_dim, out_dim, hid_dim, n_head, n_blocks, mesh_ltt, en_loc, de_loc)
        self.x_down, self.y_down = x_downsample, y_downsample
        self.x_res = int(220 / x_downsample) + 1
        self.y_res = int(50 / y_downsample) + 1
        self.en_layer = P.kaiming_mlp(self.n_head * self.in_dim, self.hid_dim, self.hid_dim)

    def ltt_mesh(self, mesh_out):
        b = mesh_out.shape[0]
        ltt = mesh_out[:, ::self.x_down, ::self.y_down, :][:, :self.x_res, :self.y_res, :]
        return ltt.reshape(b, -1, self.space_dim), mesh_out.reshape(b, -1, self.space_dim)

    def forward(self, mesh_in, func_in, mesh_out):
        size = mesh_out.shape[:-1]
        mesh_ltt, mesh_flat = self.ltt_mesh(mesh_out)
        ltt = self.encoder(mesh_in, func_in, mesh_ltt)
        ltt = self.processor(ltt, mesh_ltt)
        return self.decoder(mesh_ltt, ltt, mesh_flat).reshape(*size, self.out_dim)


class _RecomputedStep(torch.autograd.Function):
    """One rollout step whose forward keeps ONLY its input: the prediction is computed under no_grad, and the backward
    re-runs the step's forward with autograd on and back-propagates through it (parameter gradients accumulate into
    ``.grad`` as in any backward pass, the input gradient is returned).  Nothing but kernel launches on the current
    stream, so - unlike torch.utils.checkpoint with its saved-tensor hooks and RNG bookkeeping - the whole rollout
    stays capturable into one hipGraph (engine.RolloutStep(recompute=True)).  ``anchor`` is a scalar that requires
    grad: the first step's input is data, and a node none of whose inputs requires grad would never be visited."""

    @staticmethod
    def forward(ctx, anchor, model, mesh, x):
        ctx.model, ctx.mesh = model, mesh
        ctx.save_for_backward(x)
        with torch.no_grad():
            return model(mesh, x, mesh)

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        xd = x.detach().requires_grad_(True)
        with torch.enable_grad():
            out = ctx.model(ctx.mesh, xd, ctx.mesh)
        torch.autograd.backward(out, d_out)
        return None, None, None, xd.grad


def rollout_loss(model, mesh, x, y, steps: int, loss_fn, recompute: bool = False):
    """Autoregressive training objective of train_vorticity.py:118-126: ``steps`` successive
    predictions, each appended to the input history (oldest frame dropped), loss summed over the
    steps with the arguments in the script's order ``loss_fn(out, y_t)``; back-propagates through
    the whole rollout.

    Memory: one forward of the full configuration (64x64 grid, hid 256, batch 20) saves ~0.49 GB for its
    backward (decoder attention output and decoder-MLP pre-activations dominate), so the 20-step rollout
    holds ~10 GB - 3.5 % of the MI355X's 288 GB: nothing has to be recomputed, which is the default.
    ``recompute=True`` keeps only each step's input and re-runs that step's forward inside the backward
    pass (_RecomputedStep: +1 forward per step, ~25x less activation memory, capturable) for rollouts /
    batches that would not fit."""
    loss = 0.0
    anchor = torch.ones((), device=x.device, requires_grad=True) if recompute else None
    for t in range(steps):
        if recompute:
            out = _RecomputedStep.apply(anchor, model, mesh, x)
        else:
            out = model(mesh, x, mesh)
        loss = loss + loss_fn(out, y[..., t:t + 1])
        x = torch.cat((x[..., 1:], out), dim=-1)
    return loss


# ----------------------------------------------------------------------------- configs
def make_task(name: str, device="cuda", seed: int = 0):
    """Build (model, sample_fn) for a named configuration of the reference.

    ``sample_fn(batch, seed)`` returns ``(mesh_in, func_in, mesh_out, target)`` synthetic
    tensors on ``device`` with the shapes the corresponding train script feeds."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)

    def randn(*shape):
        return torch.randn(*shape, generator=g).to(device)

    def rand(*shape):
        return torch.rand(*shape, generator=g).to(device)

    if name == "darcy":                     # train_darcy.py:64-111
        mesh, ltt = grid_mesh_2d(43, True, device), grid_mesh_2d(16, True, device)
        model = pit_darcy(2, 1, 1, 64, 2, 4, ltt, 0.02, 0.02)
        sample = lambda b: (mesh, randn(b, 43, 43, 1), mesh, randn(b, 43, 43, 1))  # noqa: E731
        meta = dict(out_dim=1, p=2, batch=8)
    elif name == "burgers":                 # train_burgers.py:51-72
        mesh, ltt = line_mesh_1d(1024, device=device), line_mesh_1d(256, device=device)
        model = pit_burgers(1, 1, 1, 64, 2, 5, ltt, 0.02, 0.02)
        sample = lambda b: (mesh, randn(b, 1024, 1), mesh, randn(b, 1024, 1))  # noqa: E731
        meta = dict(out_dim=1, p=1, batch=8)
    elif name == "sod":                     # train_sod.py:55-76 (N = 1024 points on [-5,5))
        mesh, ltt = line_mesh_1d(1024, -5, 5, device), line_mesh_1d(256, -5, 5, device)
        model = pit_sod(1, 3, 3, 32, 1, 2, ltt, 0.02, 0.02)
        sample = lambda b: (mesh, randn(b, 1024, 3), mesh, randn(b, 1024, 3))  # noqa: E731
        meta = dict(out_dim=3, p=1, batch=8)
    elif name == "vorticity":               # train_vorticity.py:65-106
        mesh, ltt = grid_mesh_2d(64, False, device), grid_mesh_2d(16, False, device)
        model = pit_vorticity(2, 10, 1, 256, 2, 4, ltt, 0.02, 0.02)
        sample = lambda b: (mesh, randn(b, 64, 64, 10), mesh, randn(b, 64, 64, 1))  # noqa: E731
        meta = dict(out_dim=1, p=2, batch=20)
    elif name == "elasticity":              # train_elasticity.py:56-75, 972-point clouds
        model = pit_elasticity(2, 44, 1, 256, 2, 4, None, 0.02, 0.02)

        def sample(b):
            xy = rand(b, 972, 2)
            feats = torch.cat((xy, rand(b, 972, 42) * 5 - 1), -1)      # 5*R-1 of train_elasticity.py:12
            return xy, feats, xy, randn(b, 972, 1)
        meta = dict(out_dim=1, p=2, batch=10)
    elif name == "naca":                    # train_naca.py:68-89: 120 -> 728 -> 221x51
        model = pit_naca(2, 2, 4, 128, 1, 4, None, 4, 4, 0.02, 0.02)

        def sample(b):
            th = torch.linspace(0, 2 * np.pi, 121)[:-1]
            ax = 0.5 + 0.1 * torch.rand(b, 1, generator=g)
            by = 0.06 + 0.03 * torch.rand(b, 1, generator=g)
            outline = torch.stack((ax * torch.cos(th), by * torch.sin(th)), -1)            # (b,120,2)
            t221 = torch.linspace(0, 2 * np.pi, 222)[:-1]
            r = torch.linspace(0, 1, 51) ** 1.5 * 4.0                                      # O-grid radii
            gx = (ax.unsqueeze(-1) + r) * torch.cos(t221).reshape(1, -1, 1)
            gy = (by.unsqueeze(-1) + r) * torch.sin(t221).reshape(1, -1, 1)
            grid = torch.stack((gx, gy), -1) + 1e-3 * torch.randn(b, 221, 51, 2, generator=g)
            return outline.to(device), outline.to(device), grid.to(device), randn(b, 221, 51, 4)
        meta = dict(out_dim=4, p=2, batch=20)
    elif name == "cylinder":                # train_cylinder.py:55-84: 4390 -> 896 -> 4390 (unstructured)
        mesh = torch.rand(4390, 2, generator=g).to(device) * torch.tensor([4.0, 2.0], device=device)
        ltt = mesh[torch.randperm(4390, generator=g)[:896].to(device)].contiguous()
        model = pit_cylinder(2, 3, 3, 256, 1, 4, ltt, 0.01, 0.01)
        sample = lambda b: (mesh, randn(b, 4390, 3), mesh, randn(b, 4390, 3))  # noqa: E731
        meta = dict(out_dim=3, p=2, batch=200)
    else:
        raise KeyError(name)
    return model.to(device), sample, meta


TASKS = ("darcy", "burgers", "sod", "vorticity", "elasticity", "naca", "cylinder")
