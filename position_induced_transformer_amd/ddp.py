"""Batch-axis data parallelism for PiT: one process per GPU, one all-reduce per step.

The reference has no distributed code; the path shards on the batch axis,
every rank holds a full replica and the only exchange is the parameter-gradient sum.
All parameters' ``.grad`` are views into ONE flat fp32 buffer (79 k floats for Darcy,
1.27 M for Elasticity - a single sub-25 MB message), so a step issues exactly one
``all_reduce`` (RCCL over xGMI with backend "nccl"; gloo in the CPU tests).

``RelLpNorm`` SUMS over the batch (utils.py:98), so the single-process gradient of a
global batch equals the SUM of the per-rank gradients: the default reduction is SUM,
not DDP's mean (SURVEY section 8(e) loss-scaling note).
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class FlatGradients:
    """Owns a flat gradient buffer and points every parameter's ``.grad`` into it."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero_(self) -> None:
        self.flat.zero_()

    def all_reduce(self, average: bool = False, group=None) -> None:
        """Sum (or average) the flat buffer over the ranks; no-op without a process group."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat.div_(dist.get_world_size(group))


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s parameters (and buffers)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def shard_batch(n_items: int, rank: int, world: int) -> slice:
    """Contiguous shard of a global batch for this rank (remainder to the first ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))
