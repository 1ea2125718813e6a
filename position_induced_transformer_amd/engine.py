"""Sync-free training step: forward -> RelLp loss -> backward -> (grad all-reduce) -> (Adam),
captured once into a hipGraph and replayed.

The reference loop calls ``loss.item()`` every step and creates tensors inside forward
(train_darcy.py:124-134, pit.py:50), both capture-hostile; here the loss stays on the
device, every launch goes to the capture stream through the C ABI, and the gradients live
in one flat buffer (ddp.FlatGradients) so a data-parallel step is ONE all-reduce.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .ddp import FlatGradients


class TrainStep:
    def __init__(self, model: torch.nn.Module, batch: Sequence[torch.Tensor], out_dim: int, p: int,
                 pred_affine: Optional[Sequence[torch.Tensor]] = None, all_reduce: bool = False,
                 optimizer: Optional[torch.optim.Optimizer] = None, flat: Optional[FlatGradients] = None):
        self.model = model
        self.mesh_in, self.func_in, self.mesh_out, self.target = batch
        self.out_dim, self.p = out_dim, p
        self.affine = pred_affine
        self.all_reduce = all_reduce
        self.optimizer = optimizer
        self.flat = flat if flat is not None else FlatGradients(model.parameters())
        self.loss = torch.zeros((), device=self.func_in.device)
        self._seed = torch.ones((), device=self.func_in.device)      # d loss / d loss, allocated once
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    def _step(self) -> None:
        out = self.model(self.mesh_in, self.func_in, self.mesh_out)
        sc, sh = self.affine if self.affine is not None else (None, None)
        # the loss launch also writes its own gradient for the seed of ones below and clears the flat
        # gradient accumulators (unless FlatAdam(zero_grads=True) already did): no loss-backward launch,
        # no memset launch
        clear = None if getattr(self.optimizer, "zero_grads", False) else self.flat.flat
        loss = ops.rel_lp_loss(self.target, out, self.out_dim, self.p, sc, sh, unit_seed=self._seed, clear=clear)
        torch.autograd.backward(loss, grad_tensors=self._seed)        # no ones_like fill per step
        self.loss = loss.detach()            # same storage every replay (graph-private pool): no copy
        if self.all_reduce:
            self.flat.all_reduce()
        if self.optimizer is not None:
            self.optimizer.step()          # torch optimizer (capturable) or ddp.FlatAdam (fused HIP step)

    def run_eager(self) -> None:
        self._step()

    def capture(self, warmup: int = 3) -> None:
        """Warm up (allocator, mesh-plan caches, the per-stream accumulators of ops) and capture ON THE SAME
        side stream: a workspace first requested during capture would be allocated - and zero-filled by a
        captured memset on every replay - inside the graph."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._step()
        self._capture_stream = side

    def replay(self) -> None:
        self.graph.replay()

    def set_batch(self, func_in: torch.Tensor, target: torch.Tensor, mesh_in: Optional[torch.Tensor] = None,
                  mesh_out: Optional[torch.Tensor] = None) -> None:
        """Copy a new batch into the captured static buffers.  Tasks with per-sample meshes
        (train_elasticity.py:46, train_naca.py:62-65) also pass the batch's meshes: the captured step
        rebuilds its selection plans from the static mesh buffers on every replay, so a replay after
        ``set_batch`` computes on the new clouds.  (When mesh_in and func_in are one tensor, as in
        NACA, one copy serves both.)"""
        if (mesh_in is not None or mesh_out is not None) and not getattr(self.model.down, "_batched", True):
            raise ValueError("this model's meshes are batch-free (fixed): their selection plans were built once and "
                             "are part of the captured step; only per-sample meshes can change between replays")
        self.func_in.copy_(func_in)
        self.target.copy_(target)
        if mesh_in is not None and self.mesh_in.data_ptr() != self.func_in.data_ptr():
            self.mesh_in.copy_(mesh_in)
        if mesh_out is not None and self.mesh_out.data_ptr() != self.mesh_in.data_ptr():
            self.mesh_out.copy_(mesh_out)


class RolloutStep(TrainStep):
    """The autoregressive optimiser step of train_vorticity.py:118-129 as ONE hipGraph: ``steps`` forward
    passes (each prediction appended to the input history), the summed RelLp loss with the script's
    argument order ``myloss(out, y_t)``, one backward through the whole rollout, (all-reduce), (Adam).
    ``batch`` = (mesh, x, y) with x (b, .., memory) and y (b, .., steps)."""

    def __init__(self, model, batch, steps: int, out_dim: int, p: int, recompute: bool = False, **kw):
        mesh, x, y = batch
        super().__init__(model, (mesh, x, mesh, y), out_dim, p, **kw)
        self.steps, self.recompute = steps, recompute

    def _step(self) -> None:
        from .tasks import rollout_loss
        clear = None if getattr(self.optimizer, "zero_grads", False) else self.flat.flat
        if clear is not None:
            clear.zero_()
        loss = rollout_loss(self.model, self.mesh_in, self.func_in, self.target, self.steps,
                            lambda out, y_t: ops.rel_lp_loss(out, y_t, self.out_dim, self.p), self.recompute)
        torch.autograd.backward(loss, grad_tensors=self._seed)
        self.loss = loss.detach()
        if self.all_reduce:
            self.flat.all_reduce()
        if self.optimizer is not None:
            self.optimizer.step()
