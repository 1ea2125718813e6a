"""Drop-in for the reference's ``utils.py`` (``from utils import *``): losses, the
pixel-wise normaliser and ``count_params`` with the same names and call signatures
(utils.py:6-98).  On GPU tensors ``RelLpNorm`` runs the HIP loss kernels
(pit_rel_lp_loss_fwd/bwd); CPU tensors - the reference's offline evaluation passes them,
train_darcy.py:178 - are evaluated with the same formula in torch on the CPU.
"""
from __future__ import annotations

import operator
from functools import reduce

import torch
import torch.nn.functional as F

from . import ops

__all__ = ["PixelWiseNormalization", "count_params", "RelMaxNorm", "RelLpNorm", "F", "reduce", "operator", "torch"]


def load_reference_checkpoint(model, checkpoint, strict: bool = True):
    """Load a checkpoint written by the reference scripts: ``torch.save({'model_state':
    model.state_dict()}, 'model.pth')`` (train_darcy.py:150), whose keys carry the ``_orig_mod.``
    prefix of the torch.compile wrapper.  ``checkpoint`` is a path or the loaded dict."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        checkpoint = torch.load(checkpoint, map_location="cpu")
    state = checkpoint.get("model_state", checkpoint)
    state = {(k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k): v for k, v in state.items()}
    return model.load_state_dict(state, strict=strict)


class PixelWiseNormalization:
    """Per-pixel mean/std over the sample axis (utils.py:6-50); bilinear resampling of the
    statistics when the resolution differs (zero-shot super-resolution)."""

    def __init__(self, x, eps=1e-5):
        self.mean = torch.mean(x, dim=0, keepdim=True)
        self.std = torch.std(x, dim=0, keepdim=True)
        self.eps = eps

    def _stats_for(self, x):
        if x.shape[1:] == self.mean.shape[1:]:
            return self.mean, self.std
        size = (x.shape[1], x.shape[2])
        up = lambda t: F.interpolate(t.permute(0, 3, 1, 2), size=size, mode="bilinear",  # noqa: E731
                                     align_corners=False).permute(0, 2, 3, 1)
        return up(self.mean), up(self.std)

    def normalize(self, x):
        mean, std = self._stats_for(x)
        return (x - mean) / (std + self.eps)

    def denormalize(self, x):
        mean, std = self._stats_for(x)
        return x * (std + self.eps) + mean

    def affine(self):
        """(scale, shift) of ``denormalize`` for fusing into the loss kernel."""
        return self.std + self.eps, self.mean

    def to(self, device):
        self.mean, self.std = self.mean.to(device), self.std.to(device)

    def cuda(self):
        self.to("cuda")

    def cpu(self):
        self.to("cpu")


def count_params(model):
    return sum(reduce(operator.mul, list(p.size())) for p in model.parameters())


class RelMaxNorm:
    """Sum over the batch of the channel-mean relative max-norm error (utils.py:59-77)."""

    def __init__(self, out_dim):
        self._out_dim = out_dim

    def __call__(self, true, pred):
        if pred.is_cuda or true.is_cuda:
            return ops.rel_max_norm(true, pred, self._out_dim)      # HIP kernel (forward only, as the scripts use it)
        # host tensors (offline evaluation of saved predictions)
        t = true.reshape(true.size(0), -1, self._out_dim)
        q = pred.reshape(pred.size(0), -1, self._out_dim)
        num = torch.max(torch.abs(t - q), dim=1)[0]
        den = torch.max(torch.abs(t), dim=1)[0]
        return torch.sum(torch.mean(num / den, dim=-1))


class RelLpNorm:
    """Sum over the batch of the channel-mean relative Lp error (utils.py:80-98)."""

    def __init__(self, out_dim, p):
        self._out_dim = out_dim
        self._ord = p

    def __call__(self, true, pred):
        if pred.is_cuda or true.is_cuda:
            # device tensors always go through the HIP kernels (no eager-PyTorch path on the GPU)
            if float(self._ord) != int(self._ord) or int(self._ord) < 1:
                raise NotImplementedError("RelLpNorm on device tensors: the HIP kernels cover integer p >= 1 "
                                          f"(the reference uses p = 1 and 2), got p = {self._ord}")
            return ops.rel_lp_loss(true, pred, self._out_dim, int(self._ord))
        # host tensors only (the scripts' offline evaluation of saved predictions, train_darcy.py:178)
        t = true.reshape(true.size(0), -1, self._out_dim)
        q = pred.reshape(pred.size(0), -1, self._out_dim)
        num = torch.norm(t - q, p=self._ord, dim=1)
        den = torch.norm(t, p=self._ord, dim=1)
        return torch.sum(torch.mean(num / den, dim=-1))
