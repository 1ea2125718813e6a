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
    """Owns a flat gradient buffer and points every parameter's ``.grad`` into it.  With
    ``flatten_params=True`` the parameters themselves are also re-homed into one flat buffer
    (``flat_params``; values preserved), which is what the fused optimizer step works on."""

    def __init__(self, params: Iterable[torch.nn.Parameter], flatten_params: bool = False,
                 tail: Iterable[torch.nn.Parameter] = ()):
        """``tail``: parameters to lay out LAST, as one contiguous segment ``flat[tail_start:]`` - the bucket a
        two-bucket step all-reduces early (their gradients are complete before the backward pass ends, see
        engine.TrainStep(all_reduce_buckets=2))."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        tail_ids = {id(p) for p in tail}
        self.params = [p for p in self.params if id(p) not in tail_ids] + [p for p in self.params if id(p) in tail_ids]
        n_head_params = sum(1 for p in self.params if id(p) not in tail_ids)
        dev, dt = self.params[0].device, self.params[0].dtype
        # every parameter starts on a 64-byte boundary of the flat buffers (zero padding in between):
        # the GEMM kernels use 16-byte operand loads only on aligned weight matrices
        align = 16
        offsets, total = [], 0
        for p in self.params:
            offsets.append(total)
            total += (p.numel() + align - 1) // align * align
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.flat_params = torch.zeros(total, device=dev, dtype=dt) if flatten_params else None
        self.offsets = offsets
        self.tail_start = offsets[n_head_params] if n_head_params < len(offsets) else total
        self._views = []
        for p, off in zip(self.params, offsets):
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            self._views.append(view)
            p.grad = view
            if p.is_cuda:
                from . import ops
                ops.mark_inplace_grad(p, view)      # the backward kernels may accumulate into this view in place
            if flatten_params:
                self.flat_params[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_params[off:off + n].view_as(p)

    last_work = None          # Work handle of the last eager (not captured) all-reduce on a device backend

    def zero_(self) -> None:
        """Clear the gradients and keep them attached.  Use this (or ``zero_grad(set_to_none=False)``)
        instead of ``optimizer.zero_grad()``, whose default ``set_to_none=True`` drops the views; if
        a loop does call it, :meth:`attach` (run by ``all_reduce`` / ``FlatAdam.step``) repairs it."""
        self.attach(keep_values=False)
        self.flat.zero_()

    def attach(self, keep_values: bool = True) -> int:
        """Make every parameter's ``.grad`` the registered view into the flat buffer again.

        ``optimizer.zero_grad()`` (train_darcy.py:127; ``set_to_none=True`` by default) sets ``.grad``
        to None, after which autograd allocates fresh gradient tensors OUTSIDE the flat buffer and a
        flat all-reduce would sum stale memory.  For each detached parameter the current gradient
        (if any and ``keep_values``) is copied into its slot - a missing gradient zeroes the slot -
        and ``.grad`` points at the slot again.  Returns the number of parameters re-attached."""
        fixed = 0
        for p, view in zip(self.params, self._views):
            g = p.grad
            if g is not None and g.data_ptr() == view.data_ptr() and g.shape == view.shape:
                continue
            if g is not None and keep_values:
                view.copy_(g.detach().reshape(view.shape))
            else:
                view.zero_()
            p.grad = view
            fixed += 1
        return fixed

    def dense(self) -> torch.Tensor:
        """The gradients concatenated in parameter order WITHOUT the alignment padding (a copy)."""
        return torch.cat([p.grad.reshape(-1) for p in self.params])

    def all_reduce(self, average: bool = False, group=None, part: str = "all") -> None:
        """Sum (or average) the flat buffer over the ranks; no-op without a process group.  ``part``: "all", or one
        of the two buckets of a two-bucket step - "tail" (``flat[tail_start:]``) / "head" (``flat[:tail_start]``)."""
        if part not in ("all", "head", "tail"):
            raise ValueError(f"part must be 'all', 'head' or 'tail', got {part!r}")
        if part != "tail":
            self.attach()                   # .grad tensors that left the flat buffer are copied back first
        if not (dist.is_available() and dist.is_initialized()):
            return
        buf = self.flat if part == "all" else (self.flat[self.tail_start:] if part == "tail" else self.flat[:self.tail_start])
        if buf.numel() == 0:
            return
        if buf.is_cuda and dist.get_backend(group) == "gloo":
            # gloo has no device collectives here: stage through the host (tests that share ONE GPU between
            # ranks use this; the production backend is "nccl" = RCCL, which reduces the device buffer in place)
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            buf.copy_(host)
        else:
            # eager calls keep their Work handle (async_op + wait() is stream-ordered exactly like the blocking call):
            # TrainStep.capture() polls the LAST eager collective for completion before it starts capturing.  Captured
            # calls stay the plain blocking form.
            if buf.is_cuda and not torch.cuda.is_current_stream_capturing():
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
                if work is not None:
                    work.wait()
                    self.last_work = work
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        if average:
            buf.div_(dist.get_world_size(group))


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s parameters (and buffers)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_cuda and dist.get_backend(group) == "gloo":       # (tests that share one GPU between ranks: through the host)
            host = t.data.cpu()
            dist.broadcast(host, src=src, group=group)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src, group=group)


def shard_batch(n_items: int, rank: int, world: int) -> slice:
    """Contiguous shard of a global batch for this rank (remainder to the first ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


class FlatAdam:
    """Adam (+ optional CosineAnnealingLR) as ONE fused update over FlatGradients' flat buffers
    (pit_adam_step): same arithmetic as torch.optim.Adam(lr, betas, eps, weight_decay) followed by
    scheduler.step(), with the step counter on the device (hipGraph-replayable, no host sync)."""

    def __init__(self, flat: FlatGradients, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, cosine_t_max: int = 0, eta_min: float = 0.0, zero_grads: bool = False):
        if flat.flat_params is None:
            raise ValueError("FlatAdam needs FlatGradients(..., flatten_params=True)")
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.cosine_t_max, self.eta_min = cosine_t_max, eta_min
        # zero_grads: the update clears each gradient as it consumes it (= optimizer.zero_grad() fused in);
        # engine.TrainStep then skips its own zeroing launch
        self.zero_grads = bool(zero_grads)
        dev = flat.flat.device
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = torch.zeros((), device=dev, dtype=torch.int64)
        self.scalars = torch.zeros(4, device=dev, dtype=torch.float32)

    def step(self) -> None:
        from . import _lib, ops
        f = self.flat
        f.attach()
        ops.parameters_changed()            # raw-pointer update: cached host-evaluated head scales are void
        rc = _lib.lib().pit_adam_step(f.flat_params.data_ptr(), f.flat.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), f.flat.numel(), self.step_count.data_ptr(),
                                      self.lr, self.eta_min, self.cosine_t_max, self.betas[0], self.betas[1],
                                      self.eps, self.weight_decay, 1 if self.zero_grads else 0,
                                      self.scalars.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "pit_adam_step")
