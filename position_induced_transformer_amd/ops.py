"""Host-side operators of the PiT hot path: thin autograd wrappers over the C ABI.

Each wrapper allocates outputs / workspaces with torch (caller-owned memory, see
include/pit_hip.h), launches on torch's current HIP stream and never synchronises, so a
whole training step can be captured into a hipGraph (torch.cuda.graph).  There is no
CPU or eager-PyTorch path: tensors must live on the GPU and the shared library must load.
"""
from __future__ import annotations

import ctypes
import math
import os
import threading
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from . import _lib

METRIC_ID = {"euclid": 0, "periodic1d": 1, "periodic2d": 2}

# In-place gradient accumulation is OPT-IN per parameter: ddp.FlatGradients marks the parameters whose
# ``.grad`` it owns (``mark_inplace_grad``), and only for those - while the ``.grad`` is still the
# registered view and the parameter carries no autograd hooks - the backward kernels accumulate
# straight into ``.grad`` and autograd receives ``None`` for that input (no memset of a temporary,
# no separate ``grad += tmp`` launch).  Every other parameter gets its gradient RETURNED to autograd,
# so tensor hooks, post-accumulate hooks, torch DDP's reducer and ``torch.autograd.grad`` behave as
# with any torch op.  Set to False to disable the in-place path altogether.
FUSED_GRAD_ACCUMULATION = True

# Masked (locality < 1) layers run on per-row candidate lists (O(N*k) work) when the lists are
# much shorter than the key axis; False forces the dense MFMA kernels everywhere.
SPARSE_MASKED = True

_DSCALE_WS = {}     # (device index, stream) -> fp64 accumulators
_LOSS_WS = {}       # (device index, stream) -> loss accumulator + ticket
# Tensors whose addresses were baked into a hipGraph under capture (mesh plans, accumulators): never
# released, so a cache eviction cannot hand their memory to someone else while a graph still replays.
_PINNED = []


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


_PINNED_IDS = set()


def _pin(*objs) -> None:
    for o in objs:
        if id(o) not in _PINNED_IDS:          # (pinned objects are never released, so ids stay unique)
            _PINNED_IDS.add(id(o))
            _PINNED.append(o)


def _ws_key(device):
    return (device.index, torch.cuda.current_stream(device).cuda_stream)


def _graph_task() -> int:
    """Id of the backward pass being executed (-1 outside of one): what the end-of-pass callbacks are
    keyed on, so a pass that died with an exception cannot leave the NEXT pass without its callback."""
    return torch._C._current_graph_task_id()


def _dscale_workspace(device, n_head: int) -> torch.Tensor:
    """fp64 accumulators for d(scale): zero on entry and left zero by pit_posatt_bwd, so one
    zeroed buffer per device serves every layer (launches are stream-ordered)."""
    key = _ws_key(device)                     # per (device, stream): concurrent streams never share accumulators
    ws = _DSCALE_WS.get(key)
    if ws is None or ws.numel() < n_head * 1024 + 8:
        ws = torch.zeros(max(8, n_head) * 1024 + 8, device=device, dtype=torch.float64)   # slots + counter
        _DSCALE_WS[key] = ws
    if _capturing():
        _pin(ws)
    return ws


# d(lmda) of every attention layer of a backward pass is finished by ONE launch at the end of the
# pass (pit_posatt_dhead_finish) instead of one small kernel per layer; applies when the gradient
# goes in place into lmda.grad (the training path), needs one accumulator buffer per layer.
DEFER_HEAD_FINISH = os.environ.get("PIT_DEFER_HEAD_FINISH", "1") != "0"
# Both this table and the postponed weight-gradient jobs below are keyed on autograd's graph-task id: backward passes
# of DIFFERENT models may interleave (two Python threads calling backward(): the engine's one device thread runs their
# nodes alternately) without seeing each other's entries.  An entry whose pass raised before its end-of-pass callback
# ran is recognised when another pass wants the same accumulators / gradient slots (two live passes accumulating
# into the same .grad are a race in torch itself): its partial sums are zeroed and it is forgotten.
_PENDING_HEADS = {}      # graph-task id -> [(workspace, d_head, lmda, scale, n_head, flags), ...]
_LAYER_WS = {}
_MAX_PENDING_TASKS = 64


def _layer_workspace(slot: torch.Tensor, n_head: int) -> torch.Tensor:
    key = (slot.device.index, slot.data_ptr(), n_head)
    ws = _LAYER_WS.get(key)
    if ws is None:
        while len(_LAYER_WS) >= 1024:         # gradient buffers were re-created many times: drop the oldest
            _LAYER_WS.pop(next(iter(_LAYER_WS)))
        ws = _LAYER_WS[key] = torch.zeros(n_head * 1024, device=slot.device, dtype=torch.float64)
    if _capturing():
        _pin(ws)
    return ws


def _flush_head_finishes(task: int) -> None:
    """End-of-backward callback of graph task ``task``: one launch finishing d(lmda) of all its deferred layers."""
    # a layer applied several times in one pass (the autoregressive rollout: train_vorticity.py:122-126)
    # deferred once per application into the SAME accumulators: drain them once - duplicates in one batch
    # would race on the non-atomic "d_head += ..." of the finishing kernel
    seen, pend = set(), []
    for entry in _PENDING_HEADS.pop(task, ()):
        key = (entry[0].data_ptr(), entry[1].data_ptr())
        if key not in seen:
            seen.add(key)
            pend.append(entry)
    # a weight-gradient job the pass still holds (the encoder MLP's: its fused backward is the pass's last launch, nobody is left
    # to carry it) rides in the finishing launch when it sits on the same device and stream
    job = _PENDING_DW.get(task)
    for dev in sorted({p[0].device.index for p in pend}):
        grp = [p for p in pend if p[0].device.index == dev]
        for i in range(0, len(grp), 32):
            part = grp[i:i + 32]
            n = len(part)
            rider = None
            if job is not None and job[2].device.index == dev and job[2] == torch.cuda.current_stream(job[2].device):
                rider, job = job, None
                _PENDING_DW[task] = None
            ws = (ctypes.c_void_p * n)(*[p[0].data_ptr() for p in part])
            dh = (ctypes.c_void_p * n)(*[p[1].data_ptr() for p in part])
            hd = (ctypes.c_void_p * n)(*[p[2].data_ptr() for p in part])
            sc = (ctypes.c_void_p * n)(*[p[3].data_ptr() for p in part])
            nh = (ctypes.c_int * n)(*[p[4] for p in part])
            fl = (ctypes.c_int * n)(*[p[5] for p in part])
            with torch.cuda.device(dev):
                rc = _lib.lib().pit_posatt_dhead_finish(n, ws, dh, hd, sc, nh, fl,
                                                        ctypes.cast(ctypes.pointer(rider[0]), ctypes.c_void_p) if rider is not None else None,
                                                        _lib.stream_ptr())
            _lib.check(rc, "pit_posatt_dhead_finish")


def _defer_head_begin(work: torch.Tensor) -> None:
    """Call BEFORE the backward kernel of a deferred layer loads its accumulators ``work``.  Entries of ANOTHER pass
    on the same accumulators mean that pass raised before its end-of-pass callback ran (autograd skips the
    callbacks then): they hold partial sums of an aborted pass - zero them, forget the entries.  On the first
    deferred layer of this pass: queue its flush."""
    task = _graph_task()
    ptr = work.data_ptr()
    for other in [t for t in _PENDING_HEADS if t != task]:
        stale = [e for e in _PENDING_HEADS[other] if e[0].data_ptr() == ptr]
        if stale:
            stale[0][0].zero_()
            _PENDING_HEADS[other] = [e for e in _PENDING_HEADS[other] if e[0].data_ptr() != ptr]
            if not _PENDING_HEADS[other]:
                del _PENDING_HEADS[other]
    if task not in _PENDING_HEADS:
        while len(_PENDING_HEADS) >= _MAX_PENDING_TASKS:      # passes that died long ago and whose layers never ran again
            for e in _PENDING_HEADS.pop(next(iter(_PENDING_HEADS))):
                e[0].zero_()
        _PENDING_HEADS[task] = []
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_head_finishes(task))


def _defer_head_finish(work, d_head, head, scale, n_head: int, flags: int) -> None:
    _defer_head_begin(work)
    _PENDING_HEADS[_graph_task()].append((work, d_head, head, scale, n_head, flags))


# The weight-gradient reductions of an MLP backward (dW2|db2, dW1|db1: nothing downstream in the pass reads
# them) are postponed and handed to the NEXT attention backward as its `rider` (pit_hip.h): pit.py:116-121
# runs mlp -> attention, so in the backward the attention launch that consumes the MLP's d_x can carry the
# MLP's reductions along - three small latency-bound grids in one launch.  In-place gradient mode only.
MLP_PARAMS_RIDER = os.environ.get("PIT_DW_RIDER", "1") != "0"
# Per-THREAD step state (round 4; was module-global and toggled per step: two TrainSteps on two threads raced on it).
# engine.TrainStep sets it around its forward; the autograd nodes read it in their FORWARD (which runs on the calling
# thread) and keep it on their ctx - the backward runs on autograd's device thread, where a thread-local of the caller
# is not visible.  `processor_hook`: see _Processor.
_STEP = threading.local()


def _processor_hook():
    return getattr(_STEP, "processor_hook", None)


class step_state:
    """Context manager: per-thread overrides for the autograd nodes built inside it (engine.TrainStep._step)."""

    def __init__(self, processor_hook=None, loss=None, clear=None):
        """``loss``: a LossSpec the step will apply to the model's prediction (the fused decoder takes it); ``clear``: a contiguous
        fp32 tensor the step wants zeroed before its backward (the flat gradient buffer: the fused encoder launch clears it on
        the way).  Whoever takes either resets it to None; what is left when the forward returns is the caller's to do."""
        self.new = (processor_hook, loss, clear)

    def __enter__(self):
        self.old = (getattr(_STEP, "processor_hook", None), getattr(_STEP, "loss", None), getattr(_STEP, "clear", None),
                    getattr(_STEP, "loss_issued", None))
        _STEP.processor_hook, _STEP.loss, _STEP.clear = self.new
        _STEP.loss_issued = None
        return self

    def __exit__(self, *exc):
        _STEP.processor_hook, _STEP.loss, _STEP.clear, _STEP.loss_issued = self.old
        return False


def take_pending_clear():
    """The buffer step_state(clear=...) asked to have zeroed, if no launch of the forward took it (then the caller zeroes it)."""
    t = getattr(_STEP, "clear", None)
    _STEP.clear = None
    return t


_PENDING_DW = {}          # graph-task id -> job = (MlpParamsJob, keep-alive tensors, stream it was prepared on) or None
_DEFERRABLE = {}


def _dw_run(job) -> None:
    """Perform a postponed pit_mlp_bwd_params on the stream its inputs were produced on."""
    st, keep, stream = job
    cur = torch.cuda.current_stream(stream.device)
    with torch.cuda.stream(stream):
        rc = _lib.lib().pit_mlp_bwd_params(st.x, st.ldx, st.rows, st.n0, st.n1, st.n2, st.h, st.out_gelu, st.d_y,
                                           st.ld_dy, st.d_w1, st.d_b1, st.d_w2, st.d_b2, st.accumulate, st.scratch,
                                           st.math_mode, stream.cuda_stream)
    _lib.check(rc, "pit_mlp_bwd_params")
    if cur != stream:
        cur.wait_stream(stream)


def _dw_flush(task: int) -> None:
    """A second MLP backward of the pass with no attention in between (and the end of the pass)."""
    job = _PENDING_DW.get(task)
    if job is not None:
        _PENDING_DW[task] = None
        _dw_run(job)


def _dw_end_of_pass(task: int) -> None:
    _dw_flush(task)
    _PENDING_DW.pop(task, None)


def _dw_defer(st, keep, device) -> None:
    task = _graph_task()
    # a job ANOTHER pass left for the same gradient slots: that pass raised before its end-of-pass callback ran - its
    # gradients are void (two live passes accumulating into one .grad would be a race in torch itself)
    for other in [t for t, j in _PENDING_DW.items() if t != task and j is not None and j[0].d_w1 == st.d_w1]:
        del _PENDING_DW[other]
    if task not in _PENDING_DW:
        while len(_PENDING_DW) >= _MAX_PENDING_TASKS:         # entries of passes that died long ago
            del _PENDING_DW[next(iter(_PENDING_DW))]
        _PENDING_DW[task] = None
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _dw_end_of_pass(task))
    else:
        _dw_flush(task)
    _PENDING_DW[task] = (st, keep, torch.cuda.current_stream(device))


def _dw_take(device):
    """The job the current attention backward should carry (None if there is none for this pass / stream)."""
    task = _graph_task()
    job = _PENDING_DW.get(task)
    if job is None:
        return None
    if job[2] != torch.cuda.current_stream(device):
        _dw_flush(task)                           # produced on another stream: run it there
        return None
    _PENDING_DW[task] = None
    return job


BIG_RIDER_ROWS = 8192


def _dw_pending_rows(device) -> int:
    """Rows of the job the running pass has postponed (0: none)."""
    job = _PENDING_DW.get(_graph_task())
    return int(job[0].rows) if job is not None else 0


def _dw_slices(job, parts: int):
    """``job`` (a postponed pit_mlp_bwd_params without trailing gelu: its reductions are plain row sums) cut into
    ``parts`` row slices, each a job of its own with accumulate = 1."""
    st = job[0]
    out, per = [], -(-st.rows // parts // 16) * 16
    for r0 in range(0, st.rows, per):
        n = min(per, st.rows - r0)
        out.append(_lib.MlpParamsJob(st.x + 4 * r0 * st.ldx, st.ldx, n, st.n0, st.n1, st.n2, st.h + 4 * r0 * st.n1, 0,
                                     st.d_y + 4 * r0 * st.ld_dy, st.ld_dy, st.d_w1, st.d_b1, st.d_w2, st.d_b2, 1,
                                     st.scratch + 4 * r0 * st.n1, st.math_mode))
    return out


def _dw_deferrable(rows: int, n0: int, n1: int, n2: int, out_gelu: int, ld_dy: int) -> bool:
    key = (rows, n0, n1, n2, out_gelu, ld_dy)
    v = _DEFERRABLE.get(key)
    if v is None:
        if len(_DEFERRABLE) > 4096:
            _DEFERRABLE.clear()
        v = _DEFERRABLE[key] = bool(_lib.lib().pit_mlp_bwd_params_deferrable(*key))
    return v


def mark_inplace_grad(param, grad_view) -> None:
    """Opt ``param`` in to in-place gradient accumulation into ``grad_view`` (ddp.FlatGradients)."""
    param._pit_grad_ptr = grad_view.data_ptr()


def _grad_slot(param) -> Optional[torch.Tensor]:
    """The parameter's own .grad if the kernels may accumulate into it in place."""
    if not FUSED_GRAD_ACCUMULATION or param is None:
        return None
    want = getattr(param, "_pit_grad_ptr", None)
    g = getattr(param, "grad", None)
    if want is None or g is None or g.data_ptr() != want:
        return None                      # not opted in, or .grad was dropped / replaced (zero_grad(set_to_none=True))
    if not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous():
        return None
    if param._backward_hooks or getattr(param, "_post_accumulate_grad_hooks", None):
        return None                      # hooks must see the gradient: return it to autograd
    return g


# bf16 STORAGE flags of the `math_mode` argument (include/pit_hip.h, PIT_IO_*)
IO_X_BF16, IO_SAVE_BF16, IO_DX_BF16, IO_OUT_BF16, IO_DOUT_BF16 = 0x100, 0x200, 0x400, 0x800, 0x1000
# bf16 mode: the decoder tail (up-projection output, the decoder MLP's saved activations, their gradients) is kept in
# memory as bf16 when the shapes take the kernels that implement it (pit.decoder asks); PIT_BF16_STORAGE=0: fp32 tensors
BF16_STORAGE = os.environ.get("PIT_BF16_STORAGE", "1") != "0"


def mlp_bf16_io_supported(rows: int, n0: int, n1: int, n2: int) -> bool:
    return BF16_STORAGE and get_math_mode() == "bf16" and bool(_lib.lib().pit_mlp_bf16_io_supported(rows, n0, n1, n2, 0))


def _need_gpu_bf16_ok(t) -> None:
    if t is not None and not t.is_cuda:
        raise RuntimeError("position_induced_transformer_amd: the PiT hot path runs on the HIP device only; "
                           "got a CPU tensor (move the model and its inputs to 'cuda')")
    if t is not None and t.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"position_induced_transformer_amd: fp32 (or bf16-stored) tensor expected, got {t.dtype}")


def _need_gpu(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("position_induced_transformer_amd: the PiT hot path runs on the HIP device only; "
                               "got a CPU tensor (move the model and its inputs to 'cuda')")
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"position_induced_transformer_amd: fp32 tensors expected, got {t.dtype}")


def quantile_rank(locality: float, n_in: int):
    """(k, w) of torch.quantile's linear interpolation in ATen's fp32 arithmetic
    (SURVEY appendix A.3): rank = fl32(q)*fl32(n-1), k = floor(rank), w = rank - k."""
    rank = np.float32(locality) * np.float32(n_in - 1)
    k = int(np.floor(rank))
    w = np.float32(rank - np.float32(k))
    return k, float(w)


def mesh_period(metric: str, mesh_in: torch.Tensor) -> float:
    """Period of the periodic metrics, evaluated with the reference's own torch ops
    (pit.py:190-191 / 248-250).  One host sync; callers cache it per mesh."""
    if metric == "periodic1d":
        return float(torch.abs(mesh_in[1, 0] - mesh_in[0, 0]) * mesh_in.shape[0])
    if metric == "periodic2d":
        res = int(mesh_in.shape[0] ** 0.5)
        dx = (torch.max(mesh_in[:, 0]) - torch.min(mesh_in[:, 0])) / (res - 1)
        return float(dx * res)
    return 0.0


SLAB_UNION_MAX = 64                                          # PIT_SLAB_UNION_MAX
FOLD_SLAB_ROWS = (256, 128, 64)                              # slab heights MeshPlan.fold_plan tries, tallest first
PLAN_FLAGS = 0                                               # pit_plan_fwd's `flags` (tests: 1 = PIT_PLAN_WAVE_PER_ROW, 2 = PIT_PLAN_TWO_PASSES)
UNION_TILES = os.environ.get("PIT_UNION_TILES", "auto")      # "auto" (probe per kind of plan), "0", "1"
UNION_DV = os.environ.get("PIT_UNION_DV", "auto")            # d(values) of union-tile layers: "auto", "lists" (transposed lists)
_UNION_DECISIONS = {}
ATT_UNION = 0x2000                                           # PIT_ATT_UNION


class MeshPlan:
    """Everything about a (mesh_out, mesh_in, metric, locality) pair that does not depend on
    lmda: contiguous 3-d meshes, period, quantile rank and the selection statistics
    (m_(k), m_(k+1), m_min per row).  Fixed meshes build it once and reuse it every step."""

    __slots__ = ("mesh_out", "mesh_in", "mesh_batch", "n_out", "n_in", "sdim", "metric", "metric_id", "period",
                 "rank_k", "rank_w", "masked", "self_attn", "stats", "nbr_idx", "nbr_cnt", "nbr_cap", "rev_ptr",
                 "rev_row", "_complete", "_union", "_slab", "_fold")

    def __init__(self, metric: str, mesh_out: torch.Tensor, mesh_in: torch.Tensor, locality: float,
                 self_attn: bool, period: Optional[float] = None):
        _need_gpu(mesh_out, mesh_in)
        if metric not in METRIC_ID:
            raise ValueError(f"unknown metric {metric!r}")
        if mesh_out.dim() != mesh_in.dim() or mesh_out.dim() not in (2, 3):
            raise RuntimeError(f"mesh shapes {tuple(mesh_out.shape)} / {tuple(mesh_in.shape)} not supported")
        if mesh_out.shape[-1] != mesh_in.shape[-1]:
            raise RuntimeError("mesh_out and mesh_in must have the same number of coordinates")
        batched = mesh_out.dim() == 3
        if batched and metric != "euclid":
            raise RuntimeError("periodic metrics are defined for batch-free meshes only (pit.py:186-258)")
        if batched and mesh_out.shape[0] != mesh_in.shape[0]:
            raise RuntimeError("batched meshes must share the batch size")
        self.metric = metric
        self.metric_id = METRIC_ID[metric]
        self.mesh_out = mesh_out.detach().contiguous()
        self.mesh_in = self.mesh_out if (self_attn and mesh_in is mesh_out) else mesh_in.detach().contiguous()
        self.mesh_batch = mesh_out.shape[0] if batched else 1
        self.n_out, self.n_in, self.sdim = mesh_out.shape[-2], mesh_in.shape[-2], mesh_out.shape[-1]
        if not 1 <= self.sdim <= 3:
            raise RuntimeError("space_dim must be 1, 2 or 3")
        self.period = mesh_period(metric, self.mesh_in) if period is None else float(period)
        self.rank_k, self.rank_w = quantile_rank(locality, self.n_in)
        self.masked = bool(locality < 1.0)
        self.self_attn = bool(self_attn)
        self.stats = None
        self.nbr_idx = self.nbr_cnt = self.rev_ptr = self.rev_row = None
        self.nbr_cap = 0
        self._complete = None
        self._union = None
        self._slab = None
        self._fold = None
        cap = 0
        if self.masked and SPARSE_MASKED:
            want = self.rank_k + 2
            cap = ((want + max(16, want // 4) + 15) // 16) * 16        # k+2 keys plus room for ties
            if cap * 3 > self.n_in:
                cap = 0                                                 # lists not much shorter than the row: dense
        if self.masked or not self.self_attn:
            self.stats = torch.empty((3, self.mesh_batch, self.n_out), device=mesh_out.device, dtype=torch.float32)
        if cap:
            self._build_lists(cap, self._wants_reverse_lists(cap))      # selection + lists in one pass
        elif self.stats is not None:
            rc = _lib.lib().pit_select_fwd(self.mesh_out.data_ptr(), self.mesh_in.data_ptr(), self.mesh_batch,
                                           self.n_out, self.n_in, self.sdim, self.metric_id, self.period,
                                           self.rank_k, 1 if self.masked else 0, self.stats.data_ptr(),
                                           _lib.stream_ptr())
            _lib.check(rc, "pit_select_fwd")

    def _union_key(self, cap):
        return (self.metric, self.n_out, self.n_in, cap, self.mesh_batch > 1, self.sdim)

    def _wants_reverse_lists(self, cap: int) -> bool:
        """The transposed lists (key -> rows) serve d(values) of the candidate-list kernels only.  Plans of meshes shared by the
        batch are cached: built once, with them.  Per-sample plans are rebuilt EVERY step (train_naca.py:62-65): built without,
        and the backward that needs them (d(values) requested and not supplied by the union tiles) builds them on demand
        (ensure_reverse_lists) - an encoder whose inputs need no gradient never pays for them (NACA: 33 us, Elasticity 45 us per
        step), nor does a union-tile layer (NACA decoder: 93 us)."""
        return self.mesh_batch == 1

    def ensure_reverse_lists(self) -> None:
        if self.nbr_idx is None or self.rev_ptr is not None:
            return
        dev = self.mesh_out.device
        rev_ptr = torch.empty((self.mesh_batch, self.n_in + 1), device=dev, dtype=torch.int32)
        rev_row = torch.empty((self.mesh_batch, self.n_out * self.nbr_cap), device=dev, dtype=torch.int32)
        work = torch.empty((2 * self.mesh_batch * self.n_in,), device=dev, dtype=torch.int32)
        rc = _lib.lib().pit_lists_transpose(self.nbr_idx.data_ptr(), self.nbr_cnt.data_ptr(), self.mesh_batch, self.n_out,
                                            self.n_in, self.nbr_cap, rev_ptr.data_ptr(), rev_row.data_ptr(), work.data_ptr(),
                                            _lib.stream_ptr())
        if rc == -4:                                    # PIT_ERR_UNSUPPORTED (rows longer than 4096 keys): the whole plan again
            self._build_lists(self.nbr_cap, True)
            return
        _lib.check(rc, "pit_lists_transpose")
        self.rev_ptr, self.rev_row = rev_ptr, rev_row

    def union_tiles(self) -> bool:
        """Round 4: do the masked-layer kernels take the UNION-TILE form (PIT_ATT_UNION) for this plan?  They contract 16
        consecutive rows against the union of their candidate keys: fast when neighbouring rows share their keys (grids,
        body-fitted meshes), slow for incoherent orderings (random clouds) - never wrong.  Decided ONCE per kind of plan
        (metric, sizes, capacity, batched or not) from 32 sampled tiles of the first plan of that kind built outside a
        stream capture (the probe synchronises); plans built under capture before any probe keep the per-row kernels."""
        if self._union is not None:
            return self._union
        # (meshes shared by the batch: the per-row kernels already stream batch x dim wide rows at the HBM rate - measured)
        if self.nbr_idx is None or UNION_TILES == "0" or self.n_in > 4096 or self.n_out < 16 or self.nbr_cap > 64 \
                or (self.mesh_batch == 1 and UNION_TILES != "1"):
            self._union = False
            return False
        if UNION_TILES == "1":
            self._union = True
            return True
        key = self._union_key(self.nbr_cap)
        hit = _UNION_DECISIONS.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                return False                             # (not cached: a later eager plan of this kind may still probe)
            # 32 tiles of 16 consecutive rows of the first sample: sizes of the unions of their candidate lists
            cap, tiles = self.nbr_cap, min(32, self.n_out // 16)
            first = (torch.linspace(0, self.n_out // 16 - 1, tiles).long() * 16).to(self.nbr_idx.device)
            rows = (first[:, None] + torch.arange(16, device=first.device)[None, :]).reshape(-1)
            idx = self.nbr_idx.view(-1, cap)[rows].long()
            cnt = self.nbr_cnt[rows].long()
            keys = torch.where(torch.arange(cap, device=first.device)[None, :] < cnt[:, None], idx, -1).view(tiles, 16 * cap)
            srt = torch.sort(keys, dim=1).values
            union = (srt[:, 1:] != srt[:, :-1]).sum(1) + 1 - (srt[:, 0] < 0).long()
            mean_u, mean_k, over = (float(v) for v in torch.stack(
                [union.float().mean(), cnt.float().mean(), (cnt > cap).float().max()]).tolist())
            # MFMA work ~ union size, the per-row kernels' ~ list length; one chunk of the kernel holds 64 union keys
            hit = over == 0.0 and mean_u <= min(64.0, 4.0 * mean_k)
            _UNION_DECISIONS[key] = hit
        self._union = hit
        return hit

    def slab_plan(self):
        """Round 5: the static per-slab plan of this (fixed) mesh pair for the fused encoder- / decoder-side launches
        (csrc/pit_edge.hip; include/pit_hip.h: pit_slab_plan) as (struct, largest union of a 16-row slab's candidate keys), or
        None: per-sample meshes, no candidate lists, a list that overflowed its capacity.  Built once per plan - the meshes are
        batch-free and cached - with one host read of two ints; never under stream capture (a plan first seen there keeps the
        per-layer kernels)."""
        if self._slab is not None:
            return self._slab or None
        if self.nbr_idx is None or self.mesh_batch != 1 or not self.masked or self.n_in > 16384 or self.nbr_cap > 64:
            self._slab = False
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        dev = self.mesh_out.device
        built = self._build_slab_plan(16)
        if built is None:
            self._slab = False
            return None
        self._slab = built
        return self._slab

    def _build_slab_plan(self, rows: int):
        """pit_slab_plan_build for slabs of `rows` rows: (struct, largest union, keep-alive tensors, longest list) or None when a
        candidate list overflowed its capacity (one host read of three ints)."""
        dev = self.mesh_out.device
        n_slabs = (self.n_out + rows - 1) // rows
        m = torch.empty((n_slabs * rows, self.nbr_cap), device=dev, dtype=torch.float32)
        slot = torch.empty((n_slabs * rows, self.nbr_cap), device=dev, dtype=torch.int16)
        keys = torch.empty((n_slabs, SLAB_UNION_MAX), device=dev, dtype=torch.int32)
        nkeys = torch.empty((n_slabs,), device=dev, dtype=torch.int32)
        report = torch.zeros((3,), device=dev, dtype=torch.int32)
        rc = _lib.lib().pit_slab_plan_build(self.mesh_out.data_ptr(), self.mesh_in.data_ptr(), self.n_out, self.n_in, self.sdim,
                                            self.metric_id, self.period, self.nbr_idx.data_ptr(), self.nbr_cnt.data_ptr(),
                                            self.nbr_cap, rows, m.data_ptr(), slot.data_ptr(), keys.data_ptr(), nkeys.data_ptr(),
                                            report.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "pit_slab_plan_build")
        max_union, overflowed, max_count = report.tolist()
        if overflowed:
            return None
        sp = _lib.SlabPlan(self.n_out, self.n_in, self.nbr_cap, n_slabs, SLAB_UNION_MAX, self.stats.data_ptr(), self.rank_w,
                           self.nbr_idx.data_ptr(), self.nbr_cnt.data_ptr(), m.data_ptr(), slot.data_ptr(), keys.data_ptr(),
                           nkeys.data_ptr(), rows)
        return (sp, int(max_union), (m, slot, keys, nkeys), max(1, int(max_count)))

    def fold_plan(self):
        """Round 6: the slab plan of the folded decoder (csrc/pit_fold.hip) - the TALLEST slabs (256, 128, 64 rows) whose candidate
        keys' union still fits a tile (64 keys): the taller, the fewer adds d(values) costs.  None: per-sample meshes, no lists,
        an overflowed list, or unions beyond 64 keys even at 64 rows (incoherently ordered meshes).  Built once, never under
        stream capture."""
        if self._fold is not None:
            return self._fold or None
        if self.nbr_idx is None or self.mesh_batch != 1 or not self.masked or self.n_in > 16384 or self.nbr_cap > 64 \
                or self.n_out < 64:
            self._fold = False
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        for rows in FOLD_SLAB_ROWS:
            built = self._build_slab_plan(rows)
            if built is None:
                break
            if built[1] <= SLAB_UNION_MAX:
                # key -> the (slab, slot) entries that hold it (CSR, in slab order): the backward's tiles are summed per key in this
                # fixed order instead of being added to memory with atomics
                sp, mu, keep, mc = built
                keys, nkeys = keep[2], keep[3]
                n_slabs = keys.shape[0]
                slot = torch.arange(SLAB_UNION_MAX, device=keys.device, dtype=torch.int64)[None, :].expand(n_slabs, -1)
                valid = slot < nkeys[:, None].long()
                ent = (torch.arange(n_slabs, device=keys.device, dtype=torch.int64)[:, None] * SLAB_UNION_MAX + slot)[valid]
                key = keys.long()[valid]
                order = torch.sort(key, stable=True).indices
                rev_ent = ent[order].to(torch.int32).contiguous()
                rev_ptr = torch.zeros((self.n_in + 1,), device=keys.device, dtype=torch.int64)
                rev_ptr[1:] = torch.cumsum(torch.bincount(key, minlength=self.n_in), 0)
                rev_ptr = rev_ptr.to(torch.int32).contiguous()
                self._fold = (sp, mu, keep + (rev_ptr, rev_ent), mc)
                return self._fold
        self._fold = False
        return None

    def lists_complete(self) -> int:
        """1 if no row's candidate list overflowed its capacity (checked once, for batch-free meshes
        whose plan is cached; never during stream capture - the check synchronises), else 0."""
        if self.nbr_cnt is None or self.mesh_batch != 1:
            return 0
        if self._complete is None:
            if torch.cuda.is_current_stream_capturing():
                return 0
            self._complete = int(bool((self.nbr_cnt <= self.nbr_cap).all().item()))
        return self._complete

    def _build_lists(self, cap: int, reverse: bool = True) -> None:
        """Order statistics, candidate lists (row -> keys) and - with `reverse` - their transpose (key -> rows) for the
        sparse kernels, one pass over the rows (pit_plan_fwd)."""
        dev = self.mesh_out.device
        rows = self.mesh_batch * self.n_out
        L = _lib.lib()
        self.nbr_cap = cap
        self.nbr_idx = torch.empty((rows, cap), device=dev, dtype=torch.int32)
        self.nbr_cnt = torch.empty((rows,), device=dev, dtype=torch.int32)
        work = None
        if reverse:
            self.rev_ptr = torch.empty((self.mesh_batch, self.n_in + 1), device=dev, dtype=torch.int32)
            self.rev_row = torch.empty((self.mesh_batch, self.n_out * cap), device=dev, dtype=torch.int32)
            work = torch.empty((2 * self.mesh_batch * self.n_in,), device=dev, dtype=torch.int32)
        rc = L.pit_plan_fwd(self.mesh_out.data_ptr(), self.mesh_in.data_ptr(), self.mesh_batch, self.n_out,
                            self.n_in, self.sdim, self.metric_id, self.period, self.rank_k, self.stats.data_ptr(), cap,
                            self.nbr_idx.data_ptr(), self.nbr_cnt.data_ptr(), _lib.ptr(self.rev_ptr),
                            _lib.ptr(self.rev_row), _lib.ptr(work), int(PLAN_FLAGS), _lib.stream_ptr())
        _lib.check(rc, "pit_plan_fwd")


def _row_view(t: torch.Tensor) -> torch.Tensor:
    """(b, L, D) tensor with unit channel stride (copy only if it is not already so)."""
    if t.dim() != 3:
        raise RuntimeError(f"expected a (batch, points, channels) tensor, got {tuple(t.shape)}")
    if t.stride(2) != 1 or t.stride(1) < t.shape[2]:
        t = t.contiguous()
    return t


UNION_ATT = os.environ.get("PIT_UNION_ATT", "1") != "0"
# bf16 mode: dense self-attention of hid 128 / 256 on csrc/pit_satt.hip.  "auto": where it measured faster than the fp32-era kernels
# with rounded operands - layers of >= 256 points (per layer, with the weight tiles and the operands written by the neighbouring MLP
# chains: Elasticity 972 x 256 x 2 heads ~105 us against 191, NACA 728 x 128 x 1 ~55 against 70, Vorticity 256 x 256 x 2 ~36 against
# 47 - DESIGN.md section 4 round 6; shorter layers: not measured); "1": every supported shape; "0": never
SATT = os.environ.get("PIT_SATT", "auto")
SATT_TILES = os.environ.get("PIT_SATT_TILES", "1") != "0"      # the forward keeps its weights as bf16 tiles for the backward
SATT_FUSE_PREP = os.environ.get("PIT_SATT_FUSE_PREP", "1") != "0"      # the MLP chains either side write the bf16 operands (no prep launches)
SATT_DW_RIDER = os.environ.get("PIT_SATT_DW_RIDER", "1") != "0"        # the consuming MLP's dW / db reductions ride in the layer's backward launch
WIDE_DW_RIDER = os.environ.get("PIT_WIDE_DW_RIDER", "1") != "0"        # the same for dense layers on posatt_bwd_pair_wide_kernel (fp32 mode, large regime)


def _satt_pays(n_pts: int, n_head: int, d: int) -> bool:
    if SATT == "auto":
        return n_pts >= 256
    return SATT not in ("0", "", False)


def _union_att_ok(plan: "MeshPlan", n_head: int, d: int, b: int, values: torch.Tensor) -> bool:
    if torch.are_deterministic_algorithms_enabled():       # (d(values) leaves these kernels as fp32 atomic adds)
        return False
    if not (plan.mesh_batch == 1 and plan.masked and plan.nbr_idx is not None and not plan.self_attn and n_head in (1, 2)
            and d % 64 == 0 and d >= UNION_ATT_MIN_DIM):
        return False
    if values.stride(1) % 4 or values.stride(0) % 4 or values.data_ptr() % 16:
        return False
    if not _lib.lib().pit_union_att_supported(int(n_head), int(d), int(b), int(plan.n_out)):
        return False
    sp = plan.slab_plan()
    return sp is not None and sp[1] <= SLAB_UNION_MAX


UNION_ATT_MIN_DIM = 128            # (width 64 models take the fused decoder launch; a layer that cannot keeps the candidate-list kernels)


class _PosAtt(torch.autograd.Function):
    """dist2att + convolution (+ the self-attention concat) as one op."""

    @staticmethod
    def forward(ctx, values, head, plan: MeshPlan, n_head: int, concat: bool, head_is_scale: bool,
                head_param=None, out_slot=None, coord_dims: int = 0, scale_in=None, out_bf16: bool = False, link=None):
        _need_gpu(values, head)
        ctx.math = _math_code()
        out_bf16 = bool(out_bf16 and not concat and plan.nbr_idx is not None and ctx.math == MATH_MODES["bf16"])
        ctx.coord_dims = int(coord_dims)
        # masked layer over a coherently ordered mesh: the union-tile kernels (decided per kind of plan, MeshPlan.union_tiles)
        ctx.union = ATT_UNION if (plan.masked and plan.nbr_idx is not None and not coord_dims and n_head <= 2
                                  and values.shape[-1] % 8 == 0 and plan.union_tiles()) else 0
        # (the concat buffer arrives in a one-element list, not as a tensor argument: a tensor that is both an
        # input and the returned output would be re-materialised by autograd with a full copy)
        out_buf = out_slot[0].detach() if out_slot else None
        values = _row_view(values)
        b, j, d = values.shape
        d += int(coord_dims)                      # coordinate channels come from mesh_in inside the kernel (pit_hip.h)
        if j != plan.n_in:
            raise RuntimeError(f"inputs have {j} points but mesh_in has {plan.n_in}")
        if plan.mesh_batch not in (1, b):
            raise RuntimeError("mesh batch and input batch differ")
        head = head.detach().reshape(-1).contiguous()
        if head.numel() != n_head:
            raise RuntimeError("lmda must hold one value per head")
        # round 5: masked cross attention on a batch-free mesh pair with a slab plan and a width that is a multiple of 64 - the
        # union-tile contraction of csrc/pit_edge.hip (weights once per call, d_out read once in the backward): Vorticity / Cylinder
        ctx.uatt = None
        if UNION_ATT and not concat and not coord_dims and out_buf is None and _union_att_ok(plan, n_head, d, b, values):
            w = _new_decoder_weights(plan, head, scale_in, n_head, head_is_scale, True)      # (Q too: 2 KB per slab)
            _launch_decoder_weights(w)
            out = torch.empty((b, plan.n_out, n_head * d), device=values.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
            sp, max_union = plan.slab_plan()[0], plan.slab_plan()[1]
            rc = _lib.lib().pit_union_att_fwd(ctypes.byref(sp), values.data_ptr(), values.stride(1), values.stride(0), b, n_head, d,
                                              w.pw.data_ptr(), out.data_ptr(), out.stride(1), out.stride(0), 1 if out_bf16 else 0,
                                              max_union, _lib.stream_ptr())
            _lib.check(rc, "pit_union_att_fwd")
            ctx.uatt = w
            ctx.plan, ctx.n_head, ctx.concat, ctx.head_is_scale = plan, n_head, concat, head_is_scale
            ctx.head_param = head_param
            ctx.union = 0
            ctx.save_for_backward(values, head, w.scale, w.scale)
            return out
        # round 6, bf16 mode: the processor's dense self-attention (locality 1.0) at hid 128 / 256 on bf16 MFMA with the values rounded
        # once per layer (csrc/pit_satt.hip)
        ctx.satt = None
        if _satt_pays(int(plan.n_in), int(n_head), int(d)) and concat and not coord_dims and plan.self_attn and not plan.masked and ctx.math == MATH_MODES["bf16"] \
                and values.dtype == torch.float32 and values.stride(1) % 4 == 0 and values.stride(0) % 4 == 0 and values.data_ptr() % 16 == 0 \
                and _lib.lib().pit_satt_supported(int(plan.n_in), int(n_head), int(d), int(b), int(plan.mesh_batch)):
            k_head, k_is_scale = (scale_in, True) if scale_in is not None else (head, head_is_scale)
            out = out_buf if out_buf is not None else torch.empty((b, plan.n_out, (n_head + 1) * d), device=values.device, dtype=torch.float32)
            # (the producing MLP chain may already have written bf16(values): link["x16"] - then no prep launch)
            x16 = link.get("x16") if (link is not None and out_buf is not None) else None
            x16_ready = x16 is not None and tuple(x16.shape) == (b * j, d) and x16.dtype == torch.bfloat16 and x16.is_contiguous()
            if not x16_ready:
                x16 = torch.empty((b, j, d), device=values.device, dtype=torch.bfloat16)
            rowstat = torch.empty((plan.mesh_batch, n_head, plan.n_out, 4), device=values.device, dtype=torch.float32)
            scale = torch.empty((n_head,), device=values.device, dtype=torch.float32)
            # the forward's rounded weights, kept as MFMA A-fragment tiles for the backward's d(values) (2 MB per sample and head at
            # Elasticity's 972 points)
            et = torch.empty((plan.mesh_batch, n_head, int(_lib.lib().pit_satt_tiles_elems(int(plan.n_in)))), device=values.device,
                             dtype=torch.bfloat16) if SATT_TILES else None
            rc = _lib.lib().pit_satt_fwd(plan.mesh_in.data_ptr(), plan.mesh_batch, plan.n_in, plan.sdim, plan.metric_id, plan.period,
                                         values.data_ptr(), values.stride(1), values.stride(0), b, d, k_head.data_ptr(), n_head,
                                         1 if k_is_scale else 0, x16.data_ptr(), out.data_ptr(), out.stride(1), out.stride(0), d,
                                         1 if out_buf is None else 0, rowstat.data_ptr(), scale.data_ptr(), _lib.ptr(et),
                                         1 if x16_ready else 0, _lib.stream_ptr())
            _lib.check(rc, "pit_satt_fwd")
            ctx.satt = x16
            ctx.satt_tiles = et
            # what the consuming MLP chain's backward needs to write this layer's G16 beside its d_x (posatt_apply hangs it on `out`)
            ctx.satt_link = link
            if link is not None:
                link.update(rowstat=rowstat, n_head=n_head, pts=int(j), mesh_batch=int(plan.mesh_batch), dim=int(d), batch=int(b), g16=None,
                            dx_ptr=None)
            ctx.union = 0
            ctx.plan, ctx.n_head, ctx.concat, ctx.head_is_scale = plan, n_head, concat, head_is_scale
            ctx.head_param = head_param
            ctx.save_for_backward(values, head, rowstat, scale)
            return out
        width = (n_head + (1 if concat else 0)) * d
        copy_inputs = 1 if concat else 0
        if out_buf is not None:
            # the producer (mlp_apply(..., concat_heads=H)) already wrote `values` into columns [0, d) of this
            # buffer: the kernel only adds the head columns - no copy of the inputs (torch.cat of pit.py:44)
            out, copy_inputs = out_buf, 0
        else:
            out = torch.empty((b, plan.n_out, width), device=values.device,
                              dtype=torch.bfloat16 if out_bf16 else torch.float32)
        rowstat = torch.empty((plan.mesh_batch, n_head, plan.n_out, 4), device=values.device, dtype=torch.float32)
        scale = torch.empty((n_head,), device=values.device, dtype=torch.float32)
        # route 'host': `head` is lmda (autograd's input, the chain rule of the backward) but the kernels are
        # handed the scale c the host evaluated for it (scale_in); the backward gets it back through `scale`
        k_head, k_is_scale = (scale_in, True) if scale_in is not None else (head, head_is_scale)
        wjob = getattr(_STEP, "fwd_job", None)             # the processor's weights riding in this launch (early_block_weights)
        if wjob is not None:
            _STEP.fwd_job = None
        rc = _lib.lib().pit_posatt_fwd_job(
            plan.mesh_out.data_ptr(), plan.mesh_in.data_ptr(), plan.mesh_batch, plan.n_out, plan.n_in, plan.sdim,
            plan.metric_id, plan.period,
            values.data_ptr(), b, d, values.stride(1), values.stride(0),
            k_head.data_ptr(), n_head, 1 if k_is_scale else 0,
            _lib.ptr(plan.stats), plan.rank_w, 1 if plan.masked else 0, 1 if plan.self_attn else 0,
            out.data_ptr(), out.stride(1), out.stride(0), d if concat else 0, copy_inputs,
            rowstat.data_ptr(), scale.data_ptr(),
            _lib.ptr(plan.nbr_idx), _lib.ptr(plan.nbr_cnt), plan.nbr_cap, ctx.coord_dims,
            ctx.math | (IO_OUT_BF16 if out_bf16 else 0) | ctx.union, _lib.stream_ptr(),
            ctypes.cast(ctypes.pointer(wjob.job), ctypes.c_void_p) if wjob is not None else None)
        _lib.check(rc, "pit_posatt_fwd")
        ctx.plan, ctx.n_head, ctx.concat, ctx.head_is_scale = plan, n_head, concat, head_is_scale
        ctx.head_param = head_param
        ctx.save_for_backward(values, head, rowstat, scale)
        return out

    @staticmethod
    def backward(ctx, d_out):
        values, head, rowstat, scale = ctx.saved_tensors
        plan, n_head, concat = ctx.plan, ctx.n_head, ctx.concat
        b, j, dv = values.shape
        d = dv + ctx.coord_dims
        d_out = _row_view(d_out)
        _need_gpu_bf16_ok(d_out)
        io = IO_DOUT_BF16 if d_out.dtype == torch.bfloat16 else 0
        need_v, need_h = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        union = ctx.union
        if union:                                       # (the kernels' 16-byte row pieces: csrc/pit_posatt.hip union_ok / aligned_rows)
            k = 8 if io else 4
            if (values.data_ptr() | d_out.data_ptr()) % 16 or values.stride(1) % 4 or values.stride(0) % 4 \
                    or d_out.stride(1) % k or d_out.stride(0) % k:
                union = 0
        # d(values): from the union tiles (fp32 atomic adds: sums that differ in the last bits from run to run - not under
        # torch.use_deterministic_algorithms, nor with PIT_UNION_DV=lists), else from the transposed lists, built on demand
        if need_v and not (union and UNION_DV != "lists" and not torch.are_deterministic_algorithms_enabled()):
            plan.ensure_reverse_lists()
        d_values = torch.empty((b, j, dv), device=values.device, dtype=torch.float32) if need_v else None
        slot = _grad_slot(ctx.head_param) if need_h else None
        if slot is not None:
            d_head, acc_head = slot, 1                  # accumulate into lmda.grad in place
        else:
            d_head = torch.empty((n_head,), device=values.device, dtype=torch.float32) if need_h else None
            acc_head = 0
        defer = DEFER_HEAD_FINISH and slot is not None
        work = _layer_workspace(slot, n_head) if defer else _dscale_workspace(values.device, n_head)
        if defer:
            acc_head |= 2                               # PIT_HEAD_DEFER: finished by _flush_head_finishes
            _defer_head_begin(work)                     # (clears what an aborted pass left, before the kernel adds)

        # an MLP's postponed weight-gradient reductions ride along - unless this is a candidate-list layer and the job is
        # LARGE (the decoder MLP's: inside that launch it costs more than a launch of its own, measured 34.7 vs 16.1 +
        # 15.7 us at Darcy b=8); left pending, the fused processor spreads it over its block launches (_Processor.backward)
        # or the next MLP backward / the end of the pass runs it
        rider = None if (plan.nbr_idx is not None and _dw_pending_rows(values.device) >= BIG_RIDER_ROWS) \
            else _dw_take(values.device)
        if getattr(ctx, "satt", None) is not None:       # dense self-attention on bf16 MFMA (csrc/pit_satt.hip)
            job = None
            if rider is not None:
                if SATT_DW_RIDER:
                    job = rider                          # (carried by the backward launch below; `rider[1]` keeps its tensors alive)
                else:
                    _dw_run(rider)
            if d_out.dtype != torch.float32 or d_out.stride(1) % 4 or d_out.stride(0) % 4 or d_out.data_ptr() % 16:
                d_out = d_out.float().contiguous()
            lk = getattr(ctx, "satt_link", None)
            g16 = lk.get("g16") if lk is not None else None
            g16_ready = g16 is not None and lk.get("dx_ptr") == d_out.data_ptr() and tuple(g16.shape) == (b, n_head, j, dv)
            if lk is not None:
                lk["g16"] = None                         # (consumed)
            if not g16_ready:
                g16 = torch.empty((b, n_head, j, dv), device=values.device, dtype=torch.bfloat16)
            rc = _lib.lib().pit_satt_bwd(plan.mesh_in.data_ptr(), plan.mesh_batch, plan.n_in, plan.sdim, plan.metric_id, plan.period,
                                         b, dv, scale.data_ptr(), n_head, rowstat.data_ptr(), ctx.satt.data_ptr(), g16.data_ptr(),
                                         d_out.data_ptr(), d_out.stride(1), d_out.stride(0), dv,
                                         _lib.ptr(d_values), d_values.stride(1) if d_values is not None else 0,
                                         d_values.stride(0) if d_values is not None else 0, 1,
                                         work.data_ptr() if need_h else None, _lib.ptr(getattr(ctx, "satt_tiles", None)),
                                         1 if g16_ready else 0, ctypes.byref(job[0]) if job is not None else None, _lib.stream_ptr())
            _lib.check(rc, "pit_satt_bwd")
            if need_h:
                flags = (1 if slot is not None else 0) | (4 if ctx.head_is_scale else 0)
                if defer:
                    _defer_head_finish(work, d_head, head, scale, n_head, 1 | (4 if ctx.head_is_scale else 0))
                else:
                    _finish_heads_now(work, d_head, head, scale, n_head, flags)
            return d_values, (None if slot is not None else d_head), None, None, None, None, None, None, None, None, None, None
        if ctx.uatt is not None:                         # the union-tile backward: d_out read once, d(values) added from the tiles
            if rider is not None:
                _dw_run(rider)
            w = ctx.uatt
            if d_values is not None:
                d_values.zero_()
            sp, max_union = plan.slab_plan()[0], plan.slab_plan()[1]
            rc = _lib.lib().pit_union_att_bwd(ctypes.byref(sp), values.data_ptr(), values.stride(1), values.stride(0), b, n_head, dv,
                                              w.pw.data_ptr(), w.qw.data_ptr(), d_out.data_ptr(), d_out.stride(1), d_out.stride(0),
                                              1 if io else 0, _lib.ptr(d_values), d_values.stride(1) if d_values is not None else 0,
                                              d_values.stride(0) if d_values is not None else 0, work.data_ptr() if need_h else None,
                                              max_union, _lib.stream_ptr())
            _lib.check(rc, "pit_union_att_bwd")
            flags = (1 if slot is not None else 0) | (4 if ctx.head_is_scale else 0)
            if need_h:
                if defer:
                    _defer_head_finish(work, d_head, head, w.scale, n_head, 1 | (4 if ctx.head_is_scale else 0))
                else:
                    _finish_heads_now(work, d_head, head, w.scale, n_head, flags)
            return d_values, (None if slot is not None else d_head), None, None, None, None, None, None, None, None, None, None

        def launch(dv, dh, stream_ptr, job=None):
            rc = _lib.lib().pit_posatt_bwd(
                plan.mesh_out.data_ptr(), plan.mesh_in.data_ptr(), plan.mesh_batch, plan.n_out, plan.n_in,
                plan.sdim, plan.metric_id, plan.period,
                values.data_ptr(), b, d, values.stride(1), values.stride(0),
                head.data_ptr(), n_head, 1 if ctx.head_is_scale else 0, scale.data_ptr(),
                rowstat.data_ptr(), 1 if plan.masked else 0,
                d_out.data_ptr(), d_out.stride(1), d_out.stride(0), d if concat else 0,
                _lib.ptr(dv), dv.stride(1) if dv is not None else 0, dv.stride(0) if dv is not None else 0,
                1 if concat else 0,
                _lib.ptr(dh), acc_head, work.data_ptr(),
                _lib.ptr(plan.nbr_idx), _lib.ptr(plan.nbr_cnt), plan.nbr_cap, plan.lists_complete(),
                _lib.ptr(plan.rev_ptr), _lib.ptr(plan.rev_row),
                ctypes.cast(ctypes.pointer(job[0]), ctypes.c_void_p) if job is not None else None,
                ctx.coord_dims, ctx.math | io | union, stream_ptr)
            _lib.check(rc, "pit_posatt_bwd")

        launch(d_values, d_head, _lib.stream_ptr(), rider)
        if defer:
            _defer_head_finish(work, d_head, head, scale, n_head, 1 | (4 if ctx.head_is_scale else 0))
        return d_values, (None if slot is not None else d_head), None, None, None, None, None, None, None, None, None, None


# Where the head scale c = tan(0.25*pi*(1-1e-7)*(1+sin(lmda))) (pit.py:48) is evaluated.
#   "device" (default): inside the kernels, through fp64 with the reference's fp32 roundings - no host
#       sync, hipGraph-capturable.  ATen-CPU evaluates sin/tan with MKL VML (high-accuracy mode, NOT
#       correctly rounded, kernels chosen by the host CPU), so this c differs from the reference's by
#       1 ulp or more for ~2 % of lmda; on regular grids that can move a tie shell across the quantile
#       threshold (DESIGN.md section 2 quantifies it).
#   "host": the reference's own op sequence on the HOST CPU (torch.sin / torch.tan on a CPU copy of
#       lmda, i.e. bit-for-bit what pit.py:48 computes on that machine), injected into the kernels as
#       the scale and CACHED per lmda version (host_head_scale): one device->host copy per layer when
#       lmda changed, none while it is frozen - evaluation of a checkpoint and forward+backward with
#       frozen lmda are sync-free and hipGraph-capturable after one eager call.  d(lmda) is the kernels'
#       closed-form chain rule (1+c^2)*K*cos(lmda) in fp64 with that exact c.  A captured step that
#       UPDATES lmda (optimizer inside the graph) is refused: use 'device' there.
HEAD_SCALE_ROUTES = ("device", "host")
_ROUTE = threading.local()
_DEFAULT_ROUTE = os.environ.get("PIT_HEAD_SCALE_ROUTE", "device")
if _DEFAULT_ROUTE not in HEAD_SCALE_ROUTES:
    raise ValueError(f"PIT_HEAD_SCALE_ROUTE must be one of {HEAD_SCALE_ROUTES}, got {_DEFAULT_ROUTE!r}")


def set_head_scale_route(route: str) -> None:
    if route not in HEAD_SCALE_ROUTES:
        raise ValueError(f"head-scale route must be one of {HEAD_SCALE_ROUTES}, got {route!r}")
    _ROUTE.route = route


def get_head_scale_route() -> str:
    """Per-thread route; the process default is 'device' unless PIT_HEAD_SCALE_ROUTE=host is set in the environment
    (exact reproduction of a reference checkpoint by an unmodified script)."""
    return getattr(_ROUTE, "route", _DEFAULT_ROUTE)


class head_scale_route:
    """`with ops.head_scale_route('host'): ...` - scoped set_head_scale_route."""

    def __init__(self, route: str):
        self.route, self.prev = route, None

    def __enter__(self):
        self.prev = get_head_scale_route()
        set_head_scale_route(self.route)
        return self

    def __exit__(self, *exc):
        set_head_scale_route(self.prev)
        return False


# Anything that rewrites parameters behind autograd's back (raw-pointer writers: ddp.FlatAdam, a replayed
# hipGraph that contains an optimizer step) bumps this epoch through ``parameters_changed()``; together with the
# tensor's own version counter (bumped by every torch in-place op: torch.optim, load_state_dict, copy_) it keys the
# cached host-evaluated head scales below.
_PARAM_EPOCH = [0]
_HOSTC_CAPTURE = {"used": False, "log": []}      # scales consumed under the running stream capture: (lmda, version)
HOST_SCALE_EVALUATIONS = [0]                     # device->host round trips taken by host_head_scale (tests read it)


def parameters_changed() -> None:
    """Tell the operators that parameter VALUES were rewritten without a torch in-place op (fused optimizer
    through raw pointers, graph replay of a captured optimizer step): cached head scales are dropped.  Inside a
    stream capture that already consumed a cached scale this is an error - replays would change lmda while the
    graph keeps feeding the kernels the scale of the lmda it was captured with."""
    if _capturing() and _HOSTC_CAPTURE["used"]:
        raise RuntimeError("head-scale route 'host' is exact for a FROZEN lmda only: this capture consumed a "
                           "host-evaluated scale and now updates the parameters inside the same graph; capture "
                           "training steps that include the optimizer with the 'device' route")
    _PARAM_EPOCH[0] += 1


def host_head_scale(lmda: torch.Tensor) -> torch.Tensor:
    """c(lmda) (n_head floats on lmda's device) by the reference's own op sequence on the HOST CPU - bit-for-bit
    what pit.py:48 evaluates on this machine - cached on the tensor per (version counter, storage, epoch):
    one device->host copy when lmda changed, none while it is frozen (evaluation, fwd+bwd benchmarks), so a step
    with frozen lmda is sync-free and can be captured into a hipGraph after one eager call.  A miss under capture
    raises: the copy would synchronise."""
    key = (lmda._version, lmda.data_ptr(), _PARAM_EPOCH[0], lmda.device.index)
    ent = getattr(lmda, "_pit_host_c", None)
    if ent is None or ent[0] != key:
        if _capturing():
            raise RuntimeError("head-scale route 'host' cannot be captured into a hipGraph for an lmda whose current "
                               "value has no host-evaluated scale yet (that needs a device->host copy): it is exact "
                               "for a FROZEN lmda only - run the step once eagerly with this route first; if an "
                               "optimizer updates lmda every step, capture with the 'device' route")
        HOST_SCALE_EVALUATIONS[0] += 1
        host = lmda.detach().reshape(-1).cpu()
        c = torch.tan(0.25 * math.pi * (1 - 1e-7) * (1.0 + torch.sin(host)))
        ent = (key, c.to(lmda.device))
        lmda._pit_host_c = ent
    if _capturing():
        _HOSTC_CAPTURE["used"] = True
        _HOSTC_CAPTURE["log"].append((lmda, lmda._version))
        _pin(ent[1])
    else:
        _HOSTC_CAPTURE["used"] = False
        _HOSTC_CAPTURE["log"].clear()
    return ent[1]


def assert_frozen_since_capture() -> None:
    """After a capture (engine.TrainStep.capture): every lmda whose host-evaluated scale went into the graph must
    not have been modified by the captured work (a torch optimizer inside the graph bumps its version)."""
    log, _HOSTC_CAPTURE["log"] = _HOSTC_CAPTURE["log"], []
    _HOSTC_CAPTURE["used"] = False
    for lmda, version in log:
        if lmda._version != version:
            raise RuntimeError("head-scale route 'host' is exact for a FROZEN lmda only, but the captured step "
                               "modified lmda (an optimizer inside the graph): capture it with the 'device' route")


def _concat_buffer_of(values: torch.Tensor, n_out: int, n_head: int):
    """The concat buffer a producer attached to ``values`` (mlp_apply(..., concat_heads=H)), if ``values`` really
    is its first-columns view with the layout this layer needs; consumed once (a second consumer gets a copy)."""
    buf = getattr(values, "_pit_concat", None)
    if buf is None:
        return None
    values._pit_concat = None
    b, j, d = values.shape
    w = (1 + n_head) * d
    ok = (buf.dim() == 3 and tuple(buf.shape) == (b, n_out, w) and j == n_out and buf.is_contiguous()
          and values.data_ptr() == buf.data_ptr() and tuple(values.stride()) == (n_out * w, w, 1))
    return buf if ok else None


def tag_coords(func: torch.Tensor, mesh_in: torch.Tensor) -> torch.Tensor:
    """``func`` (b, J, C) as the non-coordinate channels of the encoder input ``cat((tile(mesh_in), func), -1)``
    that the fixed-mesh task forwards build (train_darcy.py:51-55): returns an alias of ``func`` carrying the mesh,
    so that a cross-attention layer on candidate lists reads the coordinate channels from ``mesh_in`` itself instead
    of a materialised concat (``materialize_coords`` builds the concat for every other consumer)."""
    out = func.view_as(func)                     # a fresh tensor object: never tag the caller's own tensor
    out._pit_coords = mesh_in
    return out


def materialize_coords(func: torch.Tensor) -> torch.Tensor:
    """The explicit concat for a tensor tagged by ``tag_coords`` (identity for any other tensor)."""
    mesh = getattr(func, "_pit_coords", None)
    if mesh is None:
        return func
    return torch.cat((mesh.reshape(1, -1, mesh.shape[-1]).expand(func.shape[0], -1, -1), func), -1)


@torch.compiler.disable
def posatt_apply(values: torch.Tensor, lmda: torch.Tensor, plan: MeshPlan, n_head: int, concat: bool,
                 head_is_scale: bool = False, coord_dims: int = 0, out_bf16: bool = False) -> torch.Tensor:
    """out[b,n,h*D+d] = sum_j softmax_j(-c_h m[n,j] | quantile mask)[n,j] * values[b,j,d]
    (pit.py:46-57); with ``concat`` the inputs are prepended (pit.py:44).  ``lmda`` is the
    (H,1,1) parameter, or the scale c itself when ``head_is_scale`` (tests inject it).
    ``coord_dims`` > 0: the first coord_dims value channels are the key coordinates (plan.mesh_in), ``values`` holds
    the others (candidate-list layers only, see pit_hip.h).
    Opaque to torch.compile (dynamo runs it eagerly: raw pointers cross a ctypes boundary)."""
    out_buf = _concat_buffer_of(values, plan.n_out, n_head) if concat else None
    slot = [out_buf] if out_buf is not None else None
    param = lmda if isinstance(lmda, torch.nn.Parameter) else None
    c = host_head_scale(lmda) if (not head_is_scale and get_head_scale_route() == "host") else None
    # the hand-offs between a dense self-attention layer on csrc/pit_satt.hip and the MLP chains either side of it (bf16 mode):
    # bf16(values) from the producing chain, and - on the way back - G16 from the consuming chain's backward
    link = {"x16": getattr(values, "_pit_x16", None)} if concat else None
    out = _PosAtt.apply(values, lmda.reshape(-1), plan, n_head, concat, head_is_scale, param, slot, coord_dims, c,
                        out_bf16, link)
    if link is not None and link.get("rowstat") is not None:
        out._pit_satt = link
    elif concat and plan.self_attn and not plan.masked and plan.nbr_idx is None and not coord_dims:
        out._pit_dense_att = True                # (the MLP behind it may postpone its weight-gradient reductions for this layer's backward)
    return out


# ---- one-launch MLP chains of the bf16 math mode (csrc/pit_chain.hip): hid 128 / 256 on a few thousand rows ----------------------
# The chains read their weights as bf16: copies are kept per weight tensor and re-formed when the weight changed (version counter /
# parameters_changed() epoch) - all requested copies in ONE launch.  Inside a stream capture nothing can be known about the
# weights at replay time (an optimizer between replays, or inside the graph): the cast is then always part of the graph, once per
# forward pass (pit.processor requests every block's weights at its entry: prepare_chain_weights).
CHAIN_MLP = os.environ.get("PIT_CHAIN_MLP", "1") != "0"


def chain_mlp_supported(rows: int, n0: int, n1: int, n2: int, out_gelu: bool) -> bool:
    return bool(CHAIN_MLP and out_gelu and get_math_mode() == "bf16"
                and _lib.lib().pit_mlp_chain_supported(int(rows), int(n0), int(n1), int(n2)))


def bf16_weights(tensors, new_pass: bool = False):
    """bf16 copies (RNE) of contiguous fp32 weight tensors.  The copy lives ON the tensor object it was made from (``_pit_bf16``:
    buffer, version counter, parameters_changed() epoch, address) - a key built from the address alone would hand a freed
    weight's copy to whatever tensor the allocator puts there next.  See the section comment for the capture rule."""
    cap = _capturing()
    if new_pass:
        _STEP.cap_fresh = set()
    fresh = getattr(_STEP, "cap_fresh", None) if cap else None
    out, need = [], []
    for t in tensors:
        ent = getattr(t, "_pit_bf16", None)
        if ent is None or ent[0].shape != t.shape or ent[0].device != t.device:
            ent = [torch.empty(t.shape, device=t.device, dtype=torch.bfloat16), -1, -1, 0]
            t._pit_bf16 = ent
        if cap:
            _pin(ent[0], t)
            stale = fresh is None or id(t) not in fresh
        else:
            stale = ent[1] != t._version or ent[2] != _PARAM_EPOCH[0] or ent[3] != t.data_ptr()
        if stale:
            need.append((t, ent))
        out.append(ent[0])
    if need:
        n = len(need)
        src = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _e in need])
        dst = (ctypes.c_void_p * n)(*[e[0].data_ptr() for _t, e in need])
        cnt = (ctypes.c_long * n)(*[t.numel() for t, _e in need])
        rc = _lib.lib().pit_cast_bf16_multi(n, src, dst, cnt, _lib.stream_ptr())
        _lib.check(rc, "pit_cast_bf16_multi")
        for t, ent in need:
            if cap:                       # recorded, not executed: the copy in memory is NOT this version
                ent[1] = ent[2] = -1
                if fresh is None:
                    fresh = _STEP.cap_fresh = set()
                fresh.add(id(t))
            else:
                ent[1], ent[2], ent[3] = t._version, _PARAM_EPOCH[0], t.data_ptr()
    return out


def _chain_weight_ok(w) -> bool:
    return torch.is_tensor(w) and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.data_ptr() % 16 == 0 \
        and w.numel() % 4 == 0


def prepare_chain_weights(mlps, rows: int) -> None:
    """pit.processor's entry: the bf16 copies of every block MLP's weights that will take the chain launches, in one launch."""
    ws = []
    for w1, w2 in mlps:
        n1, n0 = w1.shape
        if w2.shape == (n1, n1) and chain_mlp_supported(rows, n0, n1, n1, True) and _chain_weight_ok(w1) and _chain_weight_ok(w2):
            ws += [w1, w2]
    if ws:
        bf16_weights(ws[:32], new_pass=True)


class _Mlp(torch.autograd.Function):
    """kaiming_mlp forward/backward, optionally with the trailing gelu of pit.py:111,121."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, out_gelu: bool, concat_heads: int = 0, satt_link=None, y16_slot=None, after_dense_att: bool = False):
        _need_gpu(w1, b1, w2, b2)
        _need_gpu_bf16_ok(x)
        shape = x.shape
        n0 = shape[-1]
        n1, n2 = w1.shape[0], w2.shape[0]
        x16 = x.dtype == torch.bfloat16
        if x16 and (out_gelu or concat_heads > 0 or not mlp_bf16_io_supported(x.numel() // n0, n0, n1, n2)):
            x, x16 = x.float(), False              # (a bf16-stored input this shape's kernels do not read: widen once)
        x2 = x.reshape(-1, n0)
        if x2.stride(1) != 1 or x2.stride(0) < n0:
            x2 = x2.contiguous()
        rows = x2.shape[0]
        if w1.shape[1] != n0 or w2.shape[1] != n1:
            raise RuntimeError(f"mlp shapes do not chain: x[...,{n0}], w1{tuple(w1.shape)}, w2{tuple(w2.shape)}")
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        dev = x.device
        save_dt = torch.bfloat16 if x16 else torch.float32      # decoder tail in bf16 storage: Z1 / H saved as bf16 too
        z1 = torch.empty((rows, n1), device=dev, dtype=save_dt)
        h = torch.empty((rows, n1), device=dev, dtype=save_dt)
        z2 = torch.empty((rows, n2), device=dev, dtype=torch.float32) if out_gelu else None
        buf = None
        if concat_heads > 0:             # y goes straight into columns [0, n2) of the next self-attention's concat buffer
            buf = torch.empty((rows, (1 + concat_heads) * n2), device=dev, dtype=torch.float32)
            y = buf[:, :n2]
        else:
            y = torch.empty((rows, n2), device=dev, dtype=torch.float32)
        ctx.math = _math_code() | ((IO_X_BF16 | IO_SAVE_BF16 | IO_DX_BF16) if x16 else 0)
        ctx.x16 = x16
        ctx.chain = None
        if not x16 and chain_mlp_supported(rows, n0, n1, n2, out_gelu) and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0 \
                and _chain_weight_ok(w1) and _chain_weight_ok(w2):
            # bf16 mode, hid 128 / 256, a few thousand rows: GEMM1 + gelu + GEMM2 + gelu in ONE launch (csrc/pit_chain.hip)
            w1b, w2b = bf16_weights([w1, w2])
            # (feeding a self-attention layer: bf16(y) beside y, so that pit_satt_fwd needs no prep launch)
            y16 = torch.empty((rows, n2), device=dev, dtype=torch.bfloat16) if (y16_slot is not None and buf is not None and SATT_FUSE_PREP) else None
            rc = _lib.lib().pit_mlp_chain_fwd(x2.data_ptr(), x2.stride(0), rows, n0, n1, w1b.data_ptr(), b1c.data_ptr(),
                                              w2b.data_ptr(), b2c.data_ptr(), z1.data_ptr(), h.data_ptr(), z2.data_ptr(),
                                              y.data_ptr(), y.stride(0), _lib.ptr(y16), _lib.stream_ptr())
            _lib.check(rc, "pit_mlp_chain_fwd")
            ctx.chain = (w1b, w2b)
            if y16 is not None:
                y16_slot.append(y16)
        else:
            rc = _lib.lib().pit_mlp_fwd(x2.data_ptr(), x2.stride(0), rows, n0, n1, n2, w1c.data_ptr(), b1c.data_ptr(),
                                        w2c.data_ptr(), b2c.data_ptr(), 1 if out_gelu else 0, z1.data_ptr(),
                                        h.data_ptr(), _lib.ptr(z2), y.data_ptr(), y.stride(0), ctx.math & ~IO_DX_BF16,
                                        _lib.stream_ptr())
            _lib.check(rc, "pit_mlp_fwd")
        ctx.out_gelu, ctx.dims, ctx.in_shape = out_gelu, (rows, n0, n1, n2), shape
        ctx.satt_link = satt_link if ctx.chain is not None else None
        ctx.after_dense_att = bool(after_dense_att)
        ctx.params = (w1, b1, w2, b2)
        ctx.save_for_backward(x2, w1c, w2c, z1, h, z2 if out_gelu else z1)
        out = y.reshape(*shape[:-1], n2)             # (a view: constant row stride)
        if buf is None:
            return out
        buf = buf.reshape(*shape[:-1], (1 + concat_heads) * n2)
        ctx.mark_non_differentiable(buf)
        ctx.set_materialize_grads(False)             # no zero-filled "gradient" tensor for the buffer output
        return out, buf

    @staticmethod
    def backward(ctx, d_y, _d_buf=None):
        x2, w1, w2, z1, h, z2 = ctx.saved_tensors
        rows, n0, n1, n2 = ctx.dims
        if d_y is None:                              # (only with set_materialize_grads(False): output unused)
            d_y = torch.zeros((rows, n2), device=x2.device, dtype=torch.float32)
        dev = x2.device
        d_y2 = d_y.reshape(rows, n2)
        if d_y2.stride(1) != 1 or d_y2.stride(0) < n2:
            d_y2 = d_y2.contiguous()
        need_x = ctx.needs_input_grad[0]
        d_x = torch.empty((rows, n0), device=dev, dtype=torch.bfloat16 if ctx.x16 else torch.float32) if need_x else None
        slots = [_grad_slot(p) if isinstance(p, torch.nn.Parameter) else None for p in ctx.params]
        inplace = all(s is not None for s in slots) and all(ctx.needs_input_grad[1:5])
        if inplace:
            d_w1, d_b1, d_w2, d_b2 = slots
        else:
            d_w1 = torch.empty((n1, n0), device=dev, dtype=torch.float32)
            d_b1 = torch.empty((n1,), device=dev, dtype=torch.float32)
            d_w2 = torch.empty((n2, n1), device=dev, dtype=torch.float32)
            d_b2 = torch.empty((n2,), device=dev, dtype=torch.float32)
        scratch = torch.empty((rows * (n1 + n2),), device=dev, dtype=torch.float32)
        L = _lib.lib()
        z2p = z2.data_ptr() if ctx.out_gelu else 0
        og = 1 if ctx.out_gelu else 0
        if ctx.chain is not None and d_y2.stride(0) % 4 == 0 and d_y2.data_ptr() % 16 == 0:
            # the data path (dZ2, dZ1, d_x) in ONE launch on the forward's bf16 weight copies, then both weight-gradient reductions
            w1b, w2b = ctx.chain
            # x is a self-attention layer's concat buffer (csrc/pit_satt.hip): its backward's G16 = bf16(d_x_h / rowsum_h) is written here,
            # beside d_x - that layer then needs no prep launch (it checks that the d_out it receives IS this d_x)
            lk, g16 = ctx.satt_link, None
            if (lk is not None and SATT_FUSE_PREP and d_x is not None and lk.get("rowstat") is not None and lk["dim"] == n1
                    and (1 + lk["n_head"]) * n1 == n0 and lk["batch"] * lk["pts"] == rows):
                g16 = torch.empty((lk["batch"], lk["n_head"], lk["pts"], n1), device=dev, dtype=torch.bfloat16)
            rc = L.pit_mlp_chain_bwd(rows, n0, n1, w1b.data_ptr(), w2b.data_ptr(), z1.data_ptr(), z2p, d_y2.data_ptr(),
                                     d_y2.stride(0), _lib.ptr(d_x), n0, scratch.data_ptr(), _lib.ptr(g16),
                                     lk["rowstat"].data_ptr() if g16 is not None else None, lk["pts"] if g16 is not None else 0,
                                     lk["mesh_batch"] if g16 is not None else 0, _lib.stream_ptr())
            _lib.check(rc, "pit_mlp_chain_bwd")
            if g16 is not None:
                lk["g16"], lk["dx_ptr"] = g16, d_x.data_ptr()
            if lk is not None and inplace and SATT_DW_RIDER and lk.get("rowstat") is not None:
                # the weight-gradient reductions depend on this launch only: they ride in the attention layer's ONE backward launch
                # (pit_satt_bwd's rider) instead of standing between the two
                st = _lib.MlpParamsJob(x2.data_ptr(), x2.stride(0), rows, n0, n1, n2, h.data_ptr(), og, d_y2.data_ptr(),
                                       d_y2.stride(0), d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(),
                                       1, scratch.data_ptr(), ctx.math)
                _dw_defer(st, (x2, h, d_y2, scratch, d_w1, d_b1, d_w2, d_b2), dev)
            else:
                rc = L.pit_mlp_bwd_params(x2.data_ptr(), x2.stride(0), rows, n0, n1, n2, h.data_ptr(), og, d_y2.data_ptr(),
                                          d_y2.stride(0), d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(),
                                          1 if inplace else 0, scratch.data_ptr(), ctx.math, _lib.stream_ptr())
                _lib.check(rc, "pit_mlp_bwd_params")
        elif MLP_PARAMS_RIDER and inplace and _dw_deferrable(rows, n0, n1, n2, og, d_y2.stride(0)):
            # dZ2, dZ1 and d_x now; the weight-gradient reductions ride along with the next attention backward
            rc = L.pit_mlp_bwd_data(rows, n0, n1, n2, w1.data_ptr(), w2.data_ptr(), z1.data_ptr(), z2p, og,
                                    d_y2.data_ptr(), d_y2.stride(0), _lib.ptr(d_x), n0, scratch.data_ptr(),
                                    ctx.math, _lib.stream_ptr())
            _lib.check(rc, "pit_mlp_bwd_data")
            # (the postponed reductions of a small-regime MLP contract in fp32 in every math mode, like its fused data-path
            # kernels: as LDS-staged fp32 tiles they ride in the block launches at a third of the cost of the register-direct
            # bf16 form - bf16 mode at batch 8 was 5 % slower than fp32 for this alone)
            st = _lib.MlpParamsJob(x2.data_ptr(), x2.stride(0), rows, n0, n1, n2, h.data_ptr(), og, d_y2.data_ptr(),
                                   d_y2.stride(0), d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(),
                                   1, scratch.data_ptr(), 0)
            _dw_defer(st, (x2, h, d_y2, scratch, d_w1, d_b1, d_w2, d_b2), dev)
        elif (WIDE_DW_RIDER and inplace and ctx.after_dense_att and not ctx.x16 and rows >= 1024 and n1 >= 128 and n1 % 64 == 0
              and n2 % 64 == 0 and n0 % 64 == 0 and (not og or d_y2.stride(0) == n2)):
            # x is the output of a dense self-attention layer (pit.py:116-121): the same split in the large regime - the reductions
            # (gemm_rr_kernel: 20-44 us between this MLP's backward and the attention's, which does not depend on them) ride in the
            # attention layer's merged backward launch as tiles (pit_posatt_bwd's rider -> posatt_bwd_pair_wide_kernel); a layer
            # that takes other kernels runs them as the launch they were.  (hid 128 / 256: at hid 64 the fused call's data path is the
            # slab kernel of csrc/pit_mlp_slab.hip, which pit_mlp_bwd_data does not take - Darcy b=256 159 k -> 151 k samples/s)
            rc = L.pit_mlp_bwd_data(rows, n0, n1, n2, w1.data_ptr(), w2.data_ptr(), z1.data_ptr(), z2p, og,
                                    d_y2.data_ptr(), d_y2.stride(0), _lib.ptr(d_x), n0, scratch.data_ptr(),
                                    ctx.math, _lib.stream_ptr())
            _lib.check(rc, "pit_mlp_bwd_data")
            st = _lib.MlpParamsJob(x2.data_ptr(), x2.stride(0), rows, n0, n1, n2, h.data_ptr(), og, d_y2.data_ptr(),
                                   d_y2.stride(0), d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(),
                                   1, scratch.data_ptr(), ctx.math)
            _dw_defer(st, (x2, h, d_y2, scratch, d_w1, d_b1, d_w2, d_b2), dev)
        else:
            # one call: dZ1, then dX and both weight-gradient reductions (merged into one launch when small)
            # (small regime: fp32 in every math mode, like the postponed form above - the two must agree)
            small = not ctx.x16 and _dw_deferrable(rows, n0, n1, n2, og, d_y2.stride(0))
            rc = L.pit_mlp_bwd(x2.data_ptr(), x2.stride(0), rows, n0, n1, n2, w1.data_ptr(), w2.data_ptr(),
                               z1.data_ptr(), h.data_ptr(), z2p, og, d_y2.data_ptr(), d_y2.stride(0),
                               _lib.ptr(d_x), n0, d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(),
                               1 if inplace else 0, scratch.data_ptr(), 0 if small else ctx.math, _lib.stream_ptr())
            _lib.check(rc, "pit_mlp_bwd")
        dx = d_x.reshape(ctx.in_shape) if need_x else None
        if inplace:
            return dx, None, None, None, None, None, None, None, None, None
        return dx, d_w1, d_b1, d_w2, d_b2, None, None, None, None, None


@torch.compiler.disable
def mlp_apply(x, w1, b1, w2, b2, out_gelu: bool = False, concat_heads: int = 0) -> torch.Tensor:
    """kaiming_mlp forward (pit.py:21-26) (+ trailing gelu).  ``concat_heads=H`` (the result feeds a
    self-attention layer with H heads: pit.py:116-121) makes the kernels write the result directly into the
    first columns of that layer's (b, L, (1+H)*n2) concat buffer; the returned tensor is that strided view and
    carries the buffer (``_pit_concat``), so the attention kernel skips copying its inputs (pit.py:44)."""
    link = getattr(x, "_pit_satt", None)         # x is the output of a dense self-attention layer on csrc/pit_satt.hip
    dense = bool(getattr(x, "_pit_dense_att", False))      # ... or of one on the fp32-era kernels (posatt_apply)
    if concat_heads <= 0 or x.dim() != 3:
        return _Mlp.apply(x, w1, b1, w2, b2, out_gelu, 0, link, None, dense)
    y16_slot = []
    y, buf = _Mlp.apply(x, w1, b1, w2, b2, out_gelu, int(concat_heads), link, y16_slot, dense)
    y._pit_concat = buf
    if y16_slot:
        y._pit_x16 = y16_slot[0]
    return y


# ---------------------------------------------------------------------------------------------------------------
# Fused processor (pit.py:114-122 on batch-free meshes, small regime): csrc/pit_block.hip
BLOCK_FUSION = os.environ.get("PIT_BLOCK_FUSION", "1") != "0"
BLOCK_MAX_LAYERS = 16              # MAX_LAYERS of csrc/pit_block_dev.h: blocks whose weights one pit_block_weights launch forms


def block_fusion_supported(n_pts: int, n_head: int, dim: int, batch: int) -> bool:
    return BLOCK_FUSION and bool(_lib.lib().pit_block_supported(int(n_pts), int(n_head), int(dim), int(batch)))


# Round 4: large-regime self-attention of batch-free models on PRECOMPUTED weights (pit_posatt_pre_fwd / _bwd): the weights
# of all processor blocks from ONE pit_block_weights launch per step, the attention launches read them instead of re-forming
# exp(-c m) in every workgroup.  PIT_PRE_WEIGHTS=0: the recompute-from-coordinates kernels.
PRE_WEIGHTS = os.environ.get("PIT_PRE_WEIGHTS", "1") != "0"


def pre_weights_supported(n_pts: int, n_head: int, dim: int, batch: int) -> bool:
    return PRE_WEIGHTS and bool(_lib.lib().pit_posatt_pre_supported(int(n_pts), int(n_head), int(dim), int(batch)))


def block_weights(plan: "MeshPlan", lmdas, n_head: int, need_q: bool = True):
    """Softmax weights of len(lmdas) unmasked self-attention layers on one batch-free mesh (pit_block_weights, in launches of
    at most BLOCK_MAX_LAYERS layers): (E, Q or None, inv, rowstat, scale), each with a leading layer axis.  Route 'host':
    the host-evaluated head scales are what the kernel gets."""
    n, L, dev = len(lmdas), plan.n_in, plan.mesh_in.device
    scales = [host_head_scale(p) for p in lmdas] if get_head_scale_route() == "host" else None
    heads = [t.detach().reshape(-1).contiguous() for t in lmdas]
    kheads = scales if scales is not None else heads
    E = torch.empty((n, n_head, L, L), device=dev, dtype=torch.float32)
    Q = torch.empty((n, n_head, L, L), device=dev, dtype=torch.float32) if need_q else None
    inv = torch.empty((n, n_head, L), device=dev, dtype=torch.float32)
    rowstat = torch.empty((n, n_head, L, 4), device=dev, dtype=torch.float32)
    scale = torch.empty((n, n_head), device=dev, dtype=torch.float32)
    for l0 in range(0, n, BLOCK_MAX_LAYERS):
        m = min(BLOCK_MAX_LAYERS, n - l0)
        hp = (ctypes.c_void_p * m)(*[t.data_ptr() for t in kheads[l0:l0 + m]])
        rc = _lib.lib().pit_block_weights(plan.mesh_in.data_ptr(), L, plan.sdim, plan.metric_id, plan.period, m, hp,
                                          1 if scales is not None else 0, n_head, E[l0].data_ptr(),
                                          Q[l0].data_ptr() if Q is not None else None, inv[l0].data_ptr(),
                                          rowstat[l0].data_ptr(), scale[l0].data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "pit_block_weights")
    return E, Q, inv, rowstat, scale, heads


class _PosAttPre(torch.autograd.Function):
    """posatt.forward (pit.py:37-44) of a batch-free self-attention layer with locality 1.0 on precomputed weights:
    out = cat((values, conv), -1).  Tensor inputs: values, lmda (flat); e / q / rowstat / scale are this layer's slices of
    block_weights' outputs (functions of the mesh and lmda: d(lmda) is delivered through the saved q)."""

    @staticmethod
    def forward(ctx, values, head, e, q, rowstat, scale, n_head: int, head_param, out_slot):
        _need_gpu(values, head)
        b, L, D = values.shape
        W = (1 + n_head) * D
        if values.stride(2) != 1:
            values = values.contiguous()
        given = out_slot[0] if out_slot else None
        out = given if given is not None else torch.empty((b, L, W), device=values.device, dtype=torch.float32)
        ctx.math = _math_code()
        rc = _lib.lib().pit_posatt_pre_fwd(e.data_ptr(), rowstat.data_ptr(), L, n_head, D, b, values.data_ptr(), values.stride(1),
                                           values.stride(0), out.data_ptr(), W, L * W, D, 0 if given is not None else 1,
                                           ctx.math, _lib.stream_ptr())
        _lib.check(rc, "pit_posatt_pre_fwd")
        ctx.n_head, ctx.head_param = n_head, head_param
        ctx.save_for_backward(values, head, e, q, rowstat, scale)
        return out

    @staticmethod
    def backward(ctx, d_out):
        values, head, e, q, rowstat, scale = ctx.saved_tensors
        n_head = ctx.n_head
        b, L, D = values.shape
        d_out = _row_view(d_out)
        if d_out.dtype != torch.float32:
            d_out = d_out.float()
        need_v, need_h = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dev = values.device
        d_values = torch.empty((b, L, D), device=dev, dtype=torch.float32)
        slot = _grad_slot(ctx.head_param) if need_h else None
        defer = DEFER_HEAD_FINISH and slot is not None
        work = None
        if need_h:
            if q is None:
                raise RuntimeError("d(lmda) needs the d(scale) weights: block_weights(..., need_q=True)")
            if defer:
                work = _layer_workspace(slot, n_head)
                _defer_head_begin(work)
            else:
                work = torch.zeros(n_head * 1024, device=dev, dtype=torch.float64)
        rider = _dw_take(dev)                       # (a postponed MLP job: these launches carry nothing - run it now)
        if rider is not None:
            _dw_run(rider)
        rc = _lib.lib().pit_posatt_pre_bwd(e.data_ptr(), _lib.ptr(q), rowstat.data_ptr(), L, n_head, D, b, values.data_ptr(),
                                           values.stride(1), values.stride(0), d_out.data_ptr(), d_out.stride(1), d_out.stride(0),
                                           D, d_values.data_ptr(), D, L * D, 1, _lib.ptr(work), ctx.math, _lib.stream_ptr())
        _lib.check(rc, "pit_posatt_pre_bwd")
        d_head = None
        if need_h:
            if defer:
                _defer_head_finish(work, slot, head, scale, n_head, 1)
            else:
                d_head = slot if slot is not None else torch.empty((n_head,), device=dev, dtype=torch.float32)
                one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
                _lib.check(_lib.lib().pit_posatt_dhead_finish(1, one(work), one(d_head), one(head), one(scale), (ctypes.c_int * 1)(n_head),
                                                              (ctypes.c_int * 1)(1 if slot is not None else 0), None, _lib.stream_ptr()),
                           "pit_posatt_dhead_finish")
                if slot is not None:
                    d_head = None
        return (d_values if need_v else None), d_head, None, None, None, None, None, None, None


@torch.compiler.disable
def posatt_pre_apply(values: torch.Tensor, lmda: torch.Tensor, weights, layer: int, n_head: int) -> torch.Tensor:
    """cat((values, conv), -1) of processor block ``layer`` from ``weights`` = block_weights(...)."""
    E, Q, _inv, rowstat, scale, _heads = weights
    out_buf = _concat_buffer_of(values, values.shape[1], n_head)
    slot = [out_buf] if out_buf is not None else None
    param = lmda if isinstance(lmda, torch.nn.Parameter) else None
    return _PosAttPre.apply(values, lmda.reshape(-1), E[layer], Q[layer] if Q is not None else None, rowstat[layer], scale[layer],
                            n_head, param, slot)


# The fused processor's softmax weights (pit_block_weights) depend on the latent mesh and the lmda's only - not on the data -
# so their launch need not sit in the step's chain between the encoder and the first block.  PIT_EARLY_WEIGHTS:
#   "rider" (default)  extra workgroups of the down-projection's launch form them (pit_posatt_fwd_job: one launch less in the
#                      chain); a down-projection that is not one of the small candidate-list launches gets them as a launch
#                      of their own right after it - the order of round 3, one call earlier.
#   (a side stream under the down-projection - a parallel branch of the step graph - was measured slower twice, rounds 3 and 4:
#   Darcy b=8 0.194 -> 0.222 ms/step; a straight chain of kernels is the fastest graph this runtime replays; removed in round 5)
#   "0"                formed by _Processor.forward itself (round 3).
EARLY_WEIGHTS = os.environ.get("PIT_EARLY_WEIGHTS", "rider")


class EarlyWeights:
    """Weights of all blocks (pit_block_weights) requested before the processor runs; join() before anything reads them."""
    __slots__ = ("key", "E", "Q", "inv", "rowstat", "scale", "job", "keep")

    def join(self) -> None:
        if getattr(_STEP, "fwd_job", None) is self:        # no attention launch took the job: a launch of its own, now
            _STEP.fwd_job = None
            j = self.job
            rc = _lib.lib().pit_block_weights(j.mesh, j.n_pts, j.space_dim, j.metric, j.period, j.n_layers, j.heads,
                                              j.head_is_scale, j.n_head, j.e, j.q, j.inv, j.rowstat, j.scale_out,
                                              _lib.stream_ptr())
            _lib.check(rc, "pit_block_weights")


def drop_forward_job(early) -> None:
    """Disarm ``early`` if no launch took it (pit.encoder's finally: a forward that raised leaves nothing pending on the thread)."""
    if early is not None and getattr(_STEP, "fwd_job", None) is early:
        _STEP.fwd_job = None
    _STEP.dec_job = None                         # (a decoder-weights request nobody carried: the decoder forms them itself)


def _weights_key(plan: MeshPlan, lmdas, scales, n_head: int):
    return (id(plan), n_head, _PARAM_EPOCH[0], scales is not None,
            tuple((p._version, p.data_ptr()) for p in lmdas), tuple(t.data_ptr() for t in scales) if scales is not None else None)


def _weights_buffers(plan: MeshPlan, n: int, n_head: int, need_q: bool):
    L, dev = plan.n_in, plan.mesh_in.device
    E = torch.empty((n, n_head, L, L), device=dev, dtype=torch.float32)
    # (Q, the d(scale) weights, is only read by the backward: not formed under no_grad / in eval)
    Q = torch.empty((n, n_head, L, L), device=dev, dtype=torch.float32) if need_q else None
    inv = torch.empty((n, n_head, L), device=dev, dtype=torch.float32)
    rowstat = torch.empty((n, n_head, L, 4), device=dev, dtype=torch.float32)
    scale = torch.empty((n, n_head), device=dev, dtype=torch.float32)
    return E, Q, inv, rowstat, scale


def _launch_block_weights(plan: MeshPlan, kheads, is_scale: bool, n_head: int, need_q: bool, stream_ptr):
    n = len(kheads)
    E, Q, inv, rowstat, scale = _weights_buffers(plan, n, n_head, need_q)
    hp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in kheads])
    rc = _lib.lib().pit_block_weights(plan.mesh_in.data_ptr(), plan.n_in, plan.sdim, plan.metric_id, plan.period, n, hp,
                                      1 if is_scale else 0, n_head, E.data_ptr(), _lib.ptr(Q), inv.data_ptr(),
                                      rowstat.data_ptr(), scale.data_ptr(), stream_ptr)
    _lib.check(rc, "pit_block_weights")
    return E, Q, inv, rowstat, scale


def early_block_weights(plan: MeshPlan, lmdas, n_head: int, need_q: bool):
    """Request pit_block_weights for processor_apply(x, plan, n_head, lmdas, ...) BEFORE the layer in front of the processor
    runs (modes: EARLY_WEIGHTS above).  The caller join()s the returned handle on the same stream before its function ends
    (nothing is left pending or unjoined) and hands it to processor_apply, which uses it when lmdas / scales are still the ones
    it was formed from."""
    if EARLY_WEIGHTS != "rider" or not plan.mesh_in.is_cuda:
        return None
    scales = [host_head_scale(p) for p in lmdas] if get_head_scale_route() == "host" else None
    heads = [t.detach().reshape(-1).contiguous() for t in lmdas]
    kheads = scales if scales is not None else heads
    ew = EarlyWeights()
    ew.key = _weights_key(plan, lmdas, scales, n_head)
    ew.job = ew.keep = None
    n = len(kheads)
    ew.E, ew.Q, ew.inv, ew.rowstat, ew.scale = _weights_buffers(plan, n, n_head, need_q)
    hp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in kheads])
    ew.keep = (hp, kheads, plan)
    ew.job = _lib.BlockWeightsJob(plan.mesh_in.data_ptr(), plan.n_in, plan.sdim, plan.metric_id, plan.period, n,
                                  ctypes.cast(hp, ctypes.c_void_p), 1 if scales is not None else 0, n_head,
                                  ew.E.data_ptr(), _lib.ptr(ew.Q), ew.inv.data_ptr(), ew.rowstat.data_ptr(), ew.scale.data_ptr())
    _STEP.fwd_job = ew                       # the next attention forward of this thread carries it (_PosAtt.forward)
    return ew


class _Processor(torch.autograd.Function):
    """n_blocks x [posatt.forward -> kaiming_mlp -> gelu] (pit.py:114-122) on one batch-free latent mesh as ONE
    autograd node: the softmax weights of all blocks come from one launch (pit_block_weights: they depend on the mesh
    and the lmda's only), every block's forward is one launch (pit_block_fwd) and its backward chain one launch
    (pit_block_bwd: d(values) of block i + the data path of block i-1's MLP backward, with block i's d(scale) and its
    MLP's weight-gradient reductions riding along).  Tensor inputs: x, lmda_0..n-1, then (w1, b1, w2, b2) per block."""

    @staticmethod
    def forward(ctx, x, plan: MeshPlan, n_head: int, scales, params, early, *tensors):
        n = len(tensors) // 5
        lmdas, mlps = tensors[:n], [tensors[n + 4 * i:n + 4 * i + 4] for i in range(n)]
        _need_gpu(x, *tensors)
        b, L, D = x.shape
        H, W, rows = n_head, (1 + n_head) * D, b * L
        dev = x.device
        L_ = _lib.lib()
        # the fused blocks contract in exact fp32 in EVERY math mode: the small regime is latency-bound (an MFMA form 16x as fast
        # changes nothing) and fp32 products are inside any bf16 tolerance - so the bf16 mode never loses to fp32 at the scripts'
        # batch 8 (round 3: 31.8 k vs 40.1 k samples/s because bf16 mode fell back to one launch per layer)
        ctx.math = 0
        heads = [t.detach().reshape(-1).contiguous() for t in lmdas]
        kheads = list(scales) if scales is not None else heads         # route 'host': the host-evaluated c is what the kernels get
        need_q = any(ctx.needs_input_grad)
        if early is not None and early.key == _weights_key(plan, params[0], scales, H) and (early.Q is not None or not need_q):
            E, Q, inv, rowstat, scale = early.E, early.Q, early.inv, early.rowstat, early.scale   # (formed under the encoder)
        else:
            E, Q, inv, rowstat, scale = _launch_block_weights(plan, kheads, scales is not None, H, need_q, _lib.stream_ptr())
        buf0 = _concat_buffer_of(x, L, H)
        if buf0 is None:                       # the producer did not write into a concat buffer: one copy
            buf0 = torch.empty((b, L, W), device=dev, dtype=torch.float32)
            buf0[:, :, :D].copy_(x.detach())
        else:
            buf0 = buf0.detach()
        bufs = [buf0] + [torch.empty((b, L, W), device=dev, dtype=torch.float32) for _ in range(n - 1)]
        out = torch.empty((b, L, D), device=dev, dtype=torch.float32)
        z1 = torch.empty((n, rows, D), device=dev, dtype=torch.float32)
        hh = torch.empty((n, rows, D), device=dev, dtype=torch.float32)
        z2 = torch.empty((n, rows, D), device=dev, dtype=torch.float32)
        wts = [tuple(t.detach().contiguous() for t in m) for m in mlps]
        for i in range(n):
            w1, b1, w2, b2 = wts[i]
            if tuple(w1.shape) != (D, W) or tuple(w2.shape) != (D, D):
                raise RuntimeError(f"fused processor: block {i} MLP is {tuple(w1.shape)} / {tuple(w2.shape)}, expected "
                                   f"({D}, {W}) / ({D}, {D})")
        for i in range(n):
            y, ldy = (bufs[i + 1], W) if i + 1 < n else (out, D)
            w1, b1, w2, b2 = wts[i]
            rc = L_.pit_block_fwd(E[i].data_ptr(), inv[i].data_ptr(), L, H, D, b, bufs[i].data_ptr(), w1.data_ptr(),
                                  b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 1, z1[i].data_ptr(), hh[i].data_ptr(),
                                  z2[i].data_ptr(), y.data_ptr(), ldy, ctx.math, _lib.stream_ptr())
            _lib.check(rc, "pit_block_fwd")
        ctx.n, ctx.H, ctx.dims, ctx.plan = n, H, (b, L, D), plan
        ctx.params = params                    # (lmda parameters, (w1, b1, w2, b2) parameters) for the in-place gradient slots
        ctx.hook = _processor_hook()           # the caller's per-thread step state
        ctx.keep = (bufs, wts, heads, E, Q, inv, scale, z1, hh, z2)
        return out

    @staticmethod
    def backward(ctx, d_out):
        n, H = ctx.n, ctx.H
        b, L, D = ctx.dims
        W, rows = (1 + H) * D, b * L
        bufs, wts, heads, E, Q, inv, scale, z1, hh, z2 = ctx.keep
        lm_params, mlp_params = ctx.params
        dev = d_out.device
        L_ = _lib.lib()
        d_out = d_out.contiguous()
        dxc = [torch.empty((b, L, W), device=dev, dtype=torch.float32) for _ in range(n)]
        scratch = [torch.empty((rows * 2 * D,), device=dev, dtype=torch.float32) for _ in range(n)]
        dx = torch.empty((b, L, D), device=dev, dtype=torch.float32)
        # gradient destinations: the parameters' own .grad slots (in place, nothing returned) or fresh tensors
        lm_slots = [_grad_slot(p) if isinstance(p, torch.nn.Parameter) else None for p in lm_params]
        w_slots, w_grads = [], []
        for i in range(n):
            sl = [_grad_slot(p) if isinstance(p, torch.nn.Parameter) else None for p in mlp_params[i]]
            if all(s is not None for s in sl):
                w_slots.append(sl)
                w_grads.append(None)
            else:
                g = [torch.zeros_like(t) for t in wts[i]]
                w_slots.append(g)
                w_grads.append(g)
        defer = [DEFER_HEAD_FINISH and s is not None for s in lm_slots]
        work = []
        for i in range(n):
            if defer[i]:
                ws = _layer_workspace(lm_slots[i], H)
                _defer_head_begin(ws)
            else:
                ws = torch.zeros(H * 1024, device=dev, dtype=torch.float64)
            work.append(ws)
        # a large job the pass has postponed (the decoder MLP's weight gradients): one row slice per block launch
        extra = _dw_take(dev)
        slices = []
        if extra is not None:
            st = extra[0]
            if not st.out_gelu and st.accumulate and (st.math_mode & 0xff) == 0:
                # a two-bucket step reduces the postponed job's gradients right after block `complete_by`: every
                # slice must ride in a launch up to that one
                slices = _dw_slices(extra, n if ctx.hook is None else max(1, n - ctx.hook[1]))
            else:
                _dw_run(extra)
        # top of the chain: the last block's MLP backward (data path) from d_out
        w1, _, w2, _ = wts[n - 1]
        rc = L_.pit_mlp_bwd_data(rows, W, D, D, w1.data_ptr(), w2.data_ptr(), z1[n - 1].data_ptr(), z2[n - 1].data_ptr(), 1,
                                 d_out.data_ptr(), D, dxc[n - 1].data_ptr(), W, scratch[n - 1].data_ptr(), ctx.math,
                                 _lib.stream_ptr())
        _lib.check(rc, "pit_mlp_bwd_data")
        for i in range(n - 1, -1, -1):
            dw1, db1, dw2, db2 = w_slots[i]
            job = _lib.MlpParamsJob(bufs[i].data_ptr(), W, rows, W, D, D, hh[i].data_ptr(), 1, scratch[i].data_ptr(), D,
                                    dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), 1,
                                    scratch[i].data_ptr(), ctx.math)
            if i > 0:
                pw1, _, pw2, _ = wts[i - 1]
                prev = (pw1.data_ptr(), pw2.data_ptr(), z1[i - 1].data_ptr(), z2[i - 1].data_ptr(), 1, W,
                        dxc[i - 1].data_ptr(), W, scratch[i - 1].data_ptr(), None, 0)
            else:
                prev = (None, None, None, None, 0, 0, None, 0, None, dx.data_ptr(), D)
            k = n - 1 - i                              # launch order
            job2 = ctypes.cast(ctypes.pointer(slices[k]), ctypes.c_void_p) if k < len(slices) else None
            rc = L_.pit_block_bwd(E[i].data_ptr(), inv[i].data_ptr(), Q[i].data_ptr(), L, H, D, b, dxc[i].data_ptr(),
                                  bufs[i].data_ptr(), work[i].data_ptr(), *prev,
                                  ctypes.cast(ctypes.pointer(job), ctypes.c_void_p), job2, ctx.math,
                                  _lib.stream_ptr())
            _lib.check(rc, "pit_block_bwd")
            if ctx.hook is not None:
                ctx.hook[0](i)                              # (block i's weight gradients are now enqueued)
        # d(lmda): deferred layers are finished by the pass's one finishing launch; the others here, in one launch
        d_heads = [None] * n
        now = [i for i in range(n) if not defer[i]]
        for i in range(n):
            if defer[i]:
                _defer_head_finish(work[i], lm_slots[i], heads[i], scale[i], H, 1)
        if now:
            m = len(now)
            for i in now:
                d_heads[i] = lm_slots[i] if lm_slots[i] is not None else torch.empty((H,), device=dev, dtype=torch.float32)
            ws = (ctypes.c_void_p * m)(*[work[i].data_ptr() for i in now])
            dh = (ctypes.c_void_p * m)(*[d_heads[i].data_ptr() for i in now])
            hd = (ctypes.c_void_p * m)(*[heads[i].data_ptr() for i in now])
            sc = (ctypes.c_void_p * m)(*[scale[i].data_ptr() for i in now])
            nh = (ctypes.c_int * m)(*[H] * m)
            fl = (ctypes.c_int * m)(*[1 if lm_slots[i] is not None else 0 for i in now])
            _lib.check(L_.pit_posatt_dhead_finish(m, ws, dh, hd, sc, nh, fl, None, _lib.stream_ptr()), "pit_posatt_dhead_finish")
        grads = [dx, None, None, None, None, None]
        for i in range(n):
            g = None if lm_slots[i] is not None else d_heads[i]
            grads.append(g)
        for i in range(n):
            grads.extend(w_grads[i] if w_grads[i] is not None else [None, None, None, None])
        return tuple(grads)


# engine.TrainStep(all_reduce_buckets=2) sets step_state(processor_hook=(callback, complete_by)): the callback is called
# with the block index after each block's backward launch; everything the pass postponed before the processor (the
# decoder MLP's weight gradients) must be enqueued by the launch of block `complete_by`, after which the step all-reduces
# those gradients on its second stream.  Per thread, carried by the autograd node (see _STEP).


@torch.compiler.disable
def processor_apply(x: torch.Tensor, plan: MeshPlan, n_head: int, lmdas, mlps, early=None) -> torch.Tensor:
    """The whole processor (pit.py:114-122) on a batch-free mesh through the fused block kernels.  ``lmdas``: the
    blocks' lmda parameters; ``mlps``: per block (w1, b1, w2, b2); ``early``: early_block_weights(...) of the same plan and
    lmdas, already joined.  The caller checked block_fusion_supported()."""
    scales = None
    if get_head_scale_route() == "host":
        scales = [host_head_scale(p) for p in lmdas]
    flat = [p.reshape(-1) for p in lmdas] + [t for m in mlps for t in m]
    return _Processor.apply(x, plan, n_head, scales, (tuple(lmdas), tuple(tuple(m) for m in mlps)), early, *flat)


# ---------------------------------------------------------------------------------------------------------------
# Round 5: the encoder side (pit.py:108-112) and the decoder side (pit.py:124-127) of a small-regime model on batch-free
# meshes as ONE launch per direction each (csrc/pit_edge.hip), on the static slab plan of the mesh pair.
EDGE_FUSION = os.environ.get("PIT_EDGE_FUSION", "1") != "0"


def edge_fusion_supported(plan: MeshPlan, n_head: int, dim: int, batch: int, needs_union: bool) -> bool:
    """The fused encoder- / decoder-side launch covers this layer: a masked cross-attention on a batch-free mesh pair with
    complete candidate lists (and, for the decoder, unions of at most 64 keys per 16-row slab), 1-2 heads, hidden width 32 / 64,
    in the latency regime."""
    if not EDGE_FUSION or plan.mesh_batch != 1 or plan.self_attn or not plan.masked or plan.nbr_idx is None:
        return False
    if needs_union and torch.are_deterministic_algorithms_enabled():     # (the decoder's d(values): fp32 atomic adds)
        return False
    if not _lib.lib().pit_edge_supported(int(n_head), int(dim), int(batch), int(plan.n_out)):
        return False
    sp = plan.slab_plan()
    return sp is not None and (not needs_union or sp[1] <= SLAB_UNION_MAX)


def _finish_heads_now(work, d_head, head, scale, n_head: int, flags: int) -> None:
    """d(lmda) of ONE layer from freshly loaded accumulators, now (the gradient is returned to autograd, not accumulated in place)."""
    one = lambda t: (ctypes.c_void_p * 1)(_lib.ptr(t))
    rc = _lib.lib().pit_posatt_dhead_finish(1, one(work), one(d_head), one(head), one(scale), (ctypes.c_int * 1)(n_head),
                                            (ctypes.c_int * 1)(flags), None, _lib.stream_ptr())
    _lib.check(rc, "pit_posatt_dhead_finish")


def _mlp_grad_slots(params, shapes, needs, device):
    """([d_w1, d_b1, d_w2, d_b2], in place?) for an MLP's weight gradients: the parameters' own .grad (the kernels accumulate) when
    every one of them opted in, else fresh ZEROED tensors returned to autograd."""
    slots = [_grad_slot(p) if isinstance(p, torch.nn.Parameter) else None for p in params]
    if all(s is not None for s in slots) and all(needs):
        return slots, True
    return [torch.zeros(tuple(sh), device=device, dtype=torch.float32) for sh in shapes], False


class LossSpec:
    """The RelLp loss a training step will apply to the model's prediction (engine.TrainStep sets it in step_state): the fused
    decoder forward accumulates the loss's partial sums in its epilogue, the decoder backward forms d(pred) from them - the step
    then has no loss launch.  ``true`` (b, npts, out_dim) contiguous, ``scale`` / ``shift`` (npts, out_dim) or None, p in {1, 2}."""
    __slots__ = ("true", "scale", "shift", "p", "out_dim", "seed", "partials", "pred", "token", "value")

    def __init__(self, true, scale, shift, out_dim: int, p: int, seed=None):
        b = true.size(0)
        self.true = true.reshape(b, -1, out_dim).contiguous()
        npts = self.true.shape[1]
        self.scale = scale.reshape(npts, out_dim).contiguous() if scale is not None else None
        self.shift = shift.reshape(npts, out_dim).contiguous() if shift is not None else None
        self.p, self.out_dim, self.seed = int(p), int(out_dim), seed
        self.partials = self.pred = self.token = self.value = None

    def fits(self, batch: int, npts: int, n2: int) -> bool:
        return self.p in (1, 2) and n2 == self.out_dim and tuple(self.true.shape) == (batch, npts, n2) and not self.true.requires_grad


class DecoderWeights:
    """The up-projection's softmax weights of one step (pit_decoder_weights): P / Q tiles per 16-row slab and the head scales c.
    They depend on (mesh pair, lmda) only: formed once per step - by extra workgroups of the encoder-side launch when pit.encoder
    could request them (early_decoder_weights), else by a launch of their own in front of the decoder launch."""
    __slots__ = ("key", "pw", "qw", "scale", "job", "keep", "w1f")


def _dec_weights_key(plan: MeshPlan, lmda, scale_in, n_head: int, head_is_scale: bool, w1=None):
    return (id(plan), n_head, _PARAM_EPOCH[0], lmda._version, lmda.data_ptr(), bool(head_is_scale),
            scale_in.data_ptr() if scale_in is not None else None,
            (w1.data_ptr(), w1._version, tuple(w1.shape)) if w1 is not None else None)


def _new_decoder_weights(plan: MeshPlan, lmda, scale_in, n_head: int, head_is_scale: bool, need_q: bool, w1=None) -> DecoderWeights:
    sp, max_union, _t, max_count = plan.slab_plan()
    um = 32 if max_union <= 32 else (48 if max_union <= 48 else 64)
    dev = plan.mesh_out.device
    w = DecoderWeights()
    w.key = _dec_weights_key(plan, lmda, scale_in, n_head, head_is_scale, w1)
    # the decoder MLP's W1 in MFMA-fragment order (pit_hip.h: w1f): formed with the tiles, once per step
    w1c = w1.detach() if (w1 is not None and w1.is_contiguous() and w1.dtype == torch.float32 and w1.dim() == 2
                          and w1.shape[0] % 16 == 0 and w1.shape[1] == n_head * w1.shape[0] and w1.data_ptr() % 16 == 0) else None
    w.w1f = torch.empty_like(w1c) if w1c is not None else None
    w.pw = torch.empty((sp.n_slabs, n_head, 16, um), device=dev, dtype=torch.float32)
    w.qw = torch.empty((sp.n_slabs, n_head, 16, um), device=dev, dtype=torch.float32) if need_q else None
    w.scale = torch.empty((n_head,), device=dev, dtype=torch.float32)
    head = scale_in if scale_in is not None else lmda.detach().reshape(-1).contiguous()
    w.keep = (head, plan, sp, w1c)
    w.job = _lib.DecoderWeightsJob(ctypes.cast(ctypes.pointer(sp), ctypes.c_void_p), head.data_ptr(),
                                   1 if (scale_in is not None or head_is_scale) else 0, n_head, max_union, max_count,
                                   w.pw.data_ptr(), _lib.ptr(w.qw), w.scale.data_ptr(),
                                   _lib.ptr(w1c), _lib.ptr(w.w1f), w1c.shape[0] if w1c is not None else 0)
    return w


def _launch_decoder_weights(w: DecoderWeights) -> None:
    j = w.job
    rc = _lib.lib().pit_decoder_weights(j.plan, j.head, j.head_is_scale, j.n_head, j.max_union, j.max_count, j.pw, j.qw,
                                        j.scale_out, j.w1, j.w1f, j.dim, _lib.stream_ptr())
    _lib.check(rc, "pit_decoder_weights")


def early_decoder_weights(plan: MeshPlan, lmda, n_head: int, need_q: bool, w1=None) -> None:
    """Request the decoder's weights BEFORE the encoder-side launch of the same forward (pit.encoder): encoder_apply carries the job
    in its launch; decoder_apply uses the result when plan / lmda / route are still the ones it was formed from.  A job no launch
    took is dropped (drop_forward_job) - the decoder then forms its weights itself."""
    scale_in = host_head_scale(lmda) if get_head_scale_route() == "host" else None
    _STEP.dec_job = _new_decoder_weights(plan, lmda, scale_in, n_head, False, need_q, w1)
    _STEP.dec_ready = None


class _Decoder(torch.autograd.Function):
    """pit.decoder (pit.py:124-127): posatt_cross_* (mesh_ltt -> mesh_out) followed by the thin-output kaiming_mlp `de`."""

    @staticmethod
    def forward(ctx, values, head, plan: MeshPlan, n_head: int, head_is_scale: bool, head_param, weights, params, loss, w1, b1, w2, b2):
        _need_gpu(values, head, w1, b1, w2, b2)
        values = _row_view(values)
        if values.stride(1) % 4 or values.stride(0) % 4 or values.data_ptr() % 16:
            values = values.contiguous()
        b, j, d = values.shape
        n2 = w2.shape[0]
        dev = values.device
        sp = plan.slab_plan()[0]
        head = head.detach().reshape(-1).contiguous()
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        rows = b * plan.n_out
        need = weights.qw is not None          # (decoder_apply asked for the d(scale) tiles exactly when a backward can follow)
        y = torch.empty((b, plan.n_out, n2), device=dev, dtype=torch.float32)
        x = z1 = h = dvals = None
        if need:
            x = torch.empty((rows, n_head * d), device=dev, dtype=torch.float32)
            z1 = torch.empty((rows, d), device=dev, dtype=torch.float32)
            h = torch.empty((rows, d), device=dev, dtype=torch.float32)
            dvals = torch.empty((b, j, d), device=dev, dtype=torch.float32)       # zeroed by the launch, added to by the backward
        lt = ls = lh = lpart = None
        lp = 0
        if loss is not None and need:
            lt, ls, lh, lp = loss.true, loss.scale, loss.shift, loss.p
            lpart = loss.partials = torch.empty((b, n2, sp.n_slabs, 2), device=dev, dtype=torch.float64)
            loss.value = torch.empty((), device=dev, dtype=torch.float32)
        # (the fragment-order copy of W1 belongs to the tensor the weights were formed from)
        w1f = weights.w1f if (weights.w1f is not None and weights.keep[3].data_ptr() == w1c.data_ptr()) else None
        rc = _lib.lib().pit_decoder_fwd(ctypes.byref(sp), values.data_ptr(), values.stride(1), values.stride(0), b, n_head, d,
                                        weights.pw.data_ptr(), w1c.data_ptr(), _lib.ptr(w1f), b1c.data_ptr(), w2c.data_ptr(),
                                        b2c.data_ptr(), n2, _lib.ptr(x), _lib.ptr(z1), _lib.ptr(h), y.data_ptr(),
                                        _lib.ptr(dvals), dvals.numel() if dvals is not None else 0,
                                        _lib.ptr(lt), _lib.ptr(ls), _lib.ptr(lh), lp, _lib.ptr(lpart), plan.slab_plan()[1],
                                        _lib.stream_ptr())
        _lib.check(rc, "pit_decoder_fwd")
        ctx.plan, ctx.n_head, ctx.head_is_scale, ctx.head_param, ctx.params = plan, n_head, head_is_scale, head_param, params
        ctx.loss = loss if lpart is not None else None
        if ctx.loss is not None:
            loss.pred = y
        ctx.dvals_clean = True
        ctx.keep = (values, head, w1c, w2c, x, z1, h, weights, dvals)
        return y

    @staticmethod
    def backward(ctx, d_y):
        values, head, w1, w2, x, z1, h, weights, dvals = ctx.keep
        scale = weights.scale
        plan, n_head = ctx.plan, ctx.n_head
        b, j, d = values.shape
        n2, rows = w2.shape[0], b * plan.n_out
        dev = values.device
        sp = plan.slab_plan()[0]
        loss = ctx.loss
        inside = loss is not None and loss.token is not None and d_y.data_ptr() == loss.token.data_ptr()
        if loss is not None and loss.token is not None and not inside:
            raise RuntimeError("the fused loss's gradient token reached pit.decoder's backward as a copy: the prediction must reach "
                               "the loss through views only (engine.TrainStep)")
        if not ctx.dvals_clean:                         # a second backward through the same node (retain_graph)
            dvals.zero_()
        ctx.dvals_clean = False
        dz1 = torch.empty((rows, d), device=dev, dtype=torch.float32)
        if inside:
            d_pred = torch.empty((rows, n2), device=dev, dtype=torch.float32)
            dyp, ld = None, n2
        else:
            d_pred = d_y.reshape(rows, n2)
            if d_pred.stride(1) != 1 or d_pred.stride(0) < n2:
                d_pred = d_pred.contiguous()
            dyp, ld = d_pred, d_pred.stride(0)
        need_h = ctx.needs_input_grad[1]
        slot = _grad_slot(ctx.head_param) if need_h else None
        defer = DEFER_HEAD_FINISH and slot is not None
        work = None
        if need_h:
            work = _layer_workspace(slot, n_head) if defer else torch.zeros(n_head * 1024, device=dev, dtype=torch.float64)
            if defer:
                _defer_head_begin(work)
        L = _lib.lib()
        rc = L.pit_decoder_bwd(ctypes.byref(sp), values.data_ptr(), values.stride(1), values.stride(0), b, n_head, d,
                               weights.pw.data_ptr(), weights.qw.data_ptr(), w1.data_ptr(), w2.data_ptr(), n2, z1.data_ptr(),
                               _lib.ptr(dyp), ld, dz1.data_ptr(), dvals.data_ptr(), dvals.stride(0), _lib.ptr(work),
                               loss.pred.data_ptr() if inside else None, _lib.ptr(loss.true) if inside else None,
                               _lib.ptr(loss.scale) if inside else None, _lib.ptr(loss.shift) if inside else None,
                               _lib.ptr(loss.seed) if inside else None, loss.p if inside else 0,
                               _lib.ptr(loss.partials) if inside else None, d_pred.data_ptr() if inside else None,
                               loss.value.data_ptr() if inside else None, None, plan.slab_plan()[1], _lib.stream_ptr())
        _lib.check(rc, "pit_decoder_bwd")
        d_head = None
        if need_h:
            flags = 1 | (4 if ctx.head_is_scale else 0)
            if defer:
                _defer_head_finish(work, slot, head, scale, n_head, flags)
            else:
                d_head = slot if slot is not None else torch.empty((n_head,), device=dev, dtype=torch.float32)
                _finish_heads_now(work, d_head, head, scale, n_head, flags if slot is not None else flags & ~1)
                if slot is not None:
                    d_head = None
        # weight gradients: the reductions over all rows are postponed and carried by a later launch of the pass (_dw_defer) in the
        # in-place mode, performed now otherwise
        grads, inplace = _mlp_grad_slots(ctx.params, (w1.shape, (d,), w2.shape, (n2,)), ctx.needs_input_grad[9:13], dev)
        d_w1, d_b1, d_w2, d_b2 = grads
        st = _lib.MlpParamsJob(x.data_ptr(), x.stride(0), rows, n_head * d, d, n2, h.data_ptr(), 0, d_pred.data_ptr(),
                               d_pred.stride(0), d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(), 1,
                               dz1.data_ptr(), 0)
        keep = (x, h, d_pred, dz1, d_w1, d_b1, d_w2, d_b2)
        if inplace and MLP_PARAMS_RIDER and _dw_deferrable(rows, n_head * d, d, n2, 0, d_pred.stride(0)):
            _dw_defer(st, keep, dev)
        else:
            _dw_run((st, keep, torch.cuda.current_stream(dev)))
        dv = dvals if ctx.needs_input_grad[0] else None
        w = (None, None, None, None) if inplace else (d_w1, d_b1, d_w2, d_b2)
        return (dv, d_head, None, None, None, None, None, None, None) + w


@torch.compiler.disable
def decoder_apply(values: torch.Tensor, lmda: torch.Tensor, plan: MeshPlan, n_head: int, mlp, head_is_scale: bool = False) -> torch.Tensor:
    """pit.decoder (pit.py:124-127) as one launch per direction: de(posatt_cross(mesh_out, mesh_ltt, values)) with
    ``mlp`` = de's (w1, b1, w2, b2).  The caller checked edge_fusion_supported(plan, ..., needs_union=True)."""
    param = lmda if isinstance(lmda, torch.nn.Parameter) else None
    c = host_head_scale(lmda) if (not head_is_scale and get_head_scale_route() == "host") else None
    need_q = torch.is_grad_enabled() and (values.requires_grad or lmda.requires_grad or any(t.requires_grad for t in mlp))
    # the step's weights: formed under the encoder-side launch when pit.encoder requested them for exactly this plan / lmda / route
    weights = getattr(_STEP, "dec_ready", None)
    _STEP.dec_ready = None
    if weights is None or weights.key != _dec_weights_key(plan, lmda, c, n_head, head_is_scale, mlp[0]) or (need_q and weights.qw is None):
        weights = _new_decoder_weights(plan, lmda, c, n_head, head_is_scale, need_q, mlp[0])
        _launch_decoder_weights(weights)
    loss = getattr(_STEP, "loss", None)
    if loss is not None:
        _STEP.loss = None                        # (one decoder per step takes it)
        if not loss.fits(values.shape[0], plan.n_out, mlp[2].shape[0]):
            loss = None
        else:
            _STEP.loss_issued = loss
    return _Decoder.apply(values, lmda.reshape(-1), plan, n_head, head_is_scale, param, weights, tuple(mlp), loss, *mlp)


# ------------------------------------------------------------------------------------------------ folded decoder (round 6)
# pit.decoder = de(up(values)) with de.mlp1 folded into the values (csrc/pit_fold.hip):
#     vw = values @ W'^T  (W' = W1's memory as an (H*hid, hid) matrix: head-interleaved columns)      _Linear, n_in rows
#     z  = sum_h P_h vw_h                                                                              _FoldAtt (batch-free meshes)
#          or posatt_cross on vw for one head (any mesh kind: the candidate-list / union-tile kernels)  _PosAtt
#     y  = gelu(z + b1) @ W2^T + b2                                                                     _ThinTail
# The (batch, n_out, H*hid) tensor of pit.py:125 and the three GEMMs on its rows are gone; nothing of size n_out x hid is saved
# except z itself.  PIT_FOLD_DECODER=0: the round-5 path (attention output materialised, kaiming_mlp kernels).
FOLD_DECODER = os.environ.get("PIT_FOLD_DECODER", "1") != "0"
# d(values) of the fold attention WITHOUT atomics (per-slab tiles + a fixed-order reduction: the same bits on every run): "auto" - when
# the tiles take at most 64 MB (Vorticity b=20: 42 MB, no slower than the atomic adds; Darcy b=256: 126 MB, 1.3 % slower) or under
# torch.use_deterministic_algorithms; "1" always; "0" never (fp32 atomic adds into a zeroed buffer)
FOLD_TILES = os.environ.get("PIT_FOLD_TILES", "auto")
FOLD_TILES_MAX_BYTES = 64 << 20


def _fold_tiles(nbytes: int) -> bool:
    if FOLD_TILES in ("0", False):
        return False
    return FOLD_TILES in ("1", True) or nbytes <= FOLD_TILES_MAX_BYTES or torch.are_deterministic_algorithms_enabled()
# hid 32 / 64 models on batch-free meshes: rows (batch x output points) from which pit.decoder prefers the folded decoder to the
# fused one-launch-per-direction decoder of csrc/pit_edge.hip (which recomputes nothing but pays 16-row slabs: one gather of the
# union's value rows and one pass of atomic adds per 16 rows)
FOLD_EDGE_ROWS = int(os.environ.get("PIT_FOLD_EDGE_ROWS", "300000"))      # Darcy: batch 128 0.974 vs 0.979 ms, batch 256 1.789 vs 1.711
_ZERO_BIAS = {}


def _zero_bias(n: int, device) -> torch.Tensor:
    key = (device.index, n)
    z = _ZERO_BIAS.get(key)
    if z is None:
        z = _ZERO_BIAS[key] = torch.zeros((n,), device=device, dtype=torch.float32)
        _pin(z)
    return z


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """(b, L, D) tensor whose (b*L) rows are uniformly strided with unit channel stride (copy only if they are not)."""
    t = _row_view(t)
    if t.stride(0) != t.shape[1] * t.stride(1) or t.stride(1) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


class _Linear(torch.autograd.Function):
    """y = x @ W'^T without bias, W' = ``w``'s memory read as a (w.numel() / d, d) row-major matrix (d = x's width): for the fold,
    ``w`` is de.mlp1.weight (hid, H*hid) and W' its (H*hid, hid) view - row n*H + h of W' is W1[n, h*hid:(h+1)*hid]."""

    @staticmethod
    def forward(ctx, x, w, w_param):
        _need_gpu(x, w)
        x = _rows2d(x)
        b, j, d = x.shape
        wc = w.detach()
        if not wc.is_contiguous() or wc.data_ptr() % 16:
            wc = wc.contiguous()
        n_out = wc.numel() // d
        y = torch.empty((b, j, n_out), device=x.device, dtype=torch.float32)
        ctx.math = _math_code()
        rc = _lib.lib().pit_linear_fwd(x.data_ptr(), x.stride(1), b * j, d, n_out, wc.data_ptr(), _zero_bias(n_out, x.device).data_ptr(),
                                       y.data_ptr(), n_out, ctx.math, _lib.stream_ptr())
        _lib.check(rc, "pit_linear_fwd")
        ctx.w_param = w_param
        ctx.save_for_backward(x, wc)
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, wc = ctx.saved_tensors
        b, j, d = x.shape
        n_out = wc.numel() // d
        d_y = d_y.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_x = torch.empty((b, j, d), device=x.device, dtype=torch.float32) if need_x else None
        slot = _grad_slot(ctx.w_param) if (need_w and ctx.w_param is not None and ctx.w_param.data_ptr() == wc.data_ptr()) else None
        d_w = slot if slot is not None else (torch.empty_like(wc) if need_w else None)
        rc = _lib.lib().pit_linear_bwd(x.data_ptr(), x.stride(1), b * j, d, n_out, wc.data_ptr(), d_y.data_ptr(), n_out,
                                       _lib.ptr(d_x), d, _lib.ptr(d_w), 1 if slot is not None else 0, ctx.math, _lib.stream_ptr())
        _lib.check(rc, "pit_linear_bwd")
        return d_x, (None if slot is not None else d_w), None


class FoldWeights:
    """The softmax weights of one step on a fold plan (pit_fold_weights): pw / qw (n_slabs*H, rows, um), the head scales c."""
    __slots__ = ("pw", "qw", "scale", "keep", "pw16", "qw16")


def _new_fold_weights(plan: MeshPlan, head, scale_in, n_head: int, head_is_scale: bool, need_q: bool) -> FoldWeights:
    sp, max_union, _t, max_count = plan.fold_plan()
    um = 32 if max_union <= 32 else (48 if max_union <= 48 else 64)
    dev = plan.mesh_out.device
    w = FoldWeights()
    w.pw = torch.empty((sp.n_slabs * n_head, sp.rows, um), device=dev, dtype=torch.float32)
    w.qw = torch.empty((sp.n_slabs * n_head, sp.rows, um), device=dev, dtype=torch.float32) if need_q else None
    w.scale = torch.empty((n_head,), device=dev, dtype=torch.float32)
    k_head = scale_in if scale_in is not None else head
    w.keep = (k_head, plan, sp)
    # bf16 math mode: the fold launches read the tiles as bf16 (every workgroup of every sample and column chunk used to round the
    # same fp32 tiles on their way into LDS; now half the bytes per pass and nothing to round)
    bf = _math_code() == MATH_MODES["bf16"]
    w.pw16 = torch.empty((sp.n_slabs * n_head, sp.rows, um), device=dev, dtype=torch.bfloat16) if bf else None
    w.qw16 = torch.empty((sp.n_slabs * n_head, sp.rows, um), device=dev, dtype=torch.bfloat16) if (bf and need_q) else None
    rc = _lib.lib().pit_fold_weights(ctypes.byref(sp), k_head.data_ptr(), 1 if (scale_in is not None or head_is_scale) else 0, n_head,
                                     max_union, max_count, w.pw.data_ptr(), _lib.ptr(w.qw), w.scale.data_ptr(), _lib.ptr(w.pw16),
                                     _lib.ptr(w.qw16), _lib.stream_ptr())
    _lib.check(rc, "pit_fold_weights")
    return w


def fold_att_supported(plan: MeshPlan, n_head: int, dim: int, batch: int) -> bool:
    """The fold attention launches cover this layer: a masked cross attention on a batch-free mesh pair with complete candidate
    lists whose unions fit a tile for slabs of at least 64 rows, 1-2 heads, a width that is a multiple of 64."""
    if plan.mesh_batch != 1 or plan.self_attn or not plan.masked or plan.nbr_idx is None:
        return False
    if torch.are_deterministic_algorithms_enabled() and FOLD_TILES in ("0", False):         # (d(values) as fp32 atomic adds)
        return False
    if not _lib.lib().pit_fold_supported(int(n_head), int(dim), int(batch), int(plan.n_out), int(plan.n_in)):
        return False
    return plan.fold_plan() is not None


class _FoldAtt(torch.autograd.Function):
    """z[b, n, c] = sum_h sum_j P_h[n, j] vw[b, j, c*H + h] on a batch-free mesh pair (pit_fold_att_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, vw, head, plan: MeshPlan, n_head: int, head_is_scale: bool, head_param, scale_in, out_bf16: bool,
                need_q: bool = True):
        _need_gpu(vw, head)
        vw = _rows2d(vw)
        b, j, hd = vw.shape
        d = hd // n_head
        if j != plan.n_in:
            raise RuntimeError(f"inputs have {j} points but mesh_in has {plan.n_in}")
        head = head.detach().reshape(-1).contiguous()
        ctx.math = _math_code()
        out_bf16 = bool(out_bf16 and ctx.math == MATH_MODES["bf16"])
        w = _new_fold_weights(plan, head, scale_in, n_head, head_is_scale, need_q)
        sp, max_union = plan.fold_plan()[0], plan.fold_plan()[1]
        z = torch.empty((b, plan.n_out, d), device=vw.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        rc = _lib.lib().pit_fold_att_fwd(ctypes.byref(sp), vw.data_ptr(), vw.stride(1), vw.stride(0), b, n_head, d,
                                         (w.pw16 if w.pw16 is not None else w.pw).data_ptr(),
                                         z.data_ptr(), z.stride(1), z.stride(0), max_union,
                                         ctx.math | (IO_OUT_BF16 if out_bf16 else 0), _lib.stream_ptr())
        _lib.check(rc, "pit_fold_att_fwd")
        ctx.plan, ctx.n_head, ctx.head_is_scale, ctx.head_param, ctx.w = plan, n_head, head_is_scale, head_param, w
        ctx.save_for_backward(vw, head)
        return z

    @staticmethod
    def backward(ctx, dz):
        vw, head = ctx.saved_tensors
        plan, n_head, w = ctx.plan, ctx.n_head, ctx.w
        b, j, hd = vw.shape
        d = hd // n_head
        dz = _row_view(dz)
        _need_gpu_bf16_ok(dz)
        io = IO_DOUT_BF16 if dz.dtype == torch.bfloat16 else 0
        if dz.stride(1) % 4 or dz.stride(0) % 4 or dz.data_ptr() % 16:
            dz = dz.contiguous()
        need_v, need_h = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_h and w.qw is None:
            raise RuntimeError("fold attention: d(lmda) requested but the forward ran without grad mode")
        # d(values): per-slab sums into tiles + a fixed-order reduction per key (no atomics, nothing to zero: FOLD_TILES), or fp32
        # atomic adds into a zeroed buffer
        fp = plan.fold_plan()
        tiles = rev_ptr = rev_ent = None
        if need_v and _fold_tiles(4 * b * fp[0].n_slabs * SLAB_UNION_MAX * hd):
            tiles = torch.empty((b, fp[0].n_slabs, SLAB_UNION_MAX, hd), device=vw.device, dtype=torch.float32)
            rev_ptr, rev_ent = fp[2][4], fp[2][5]
            d_vw = torch.empty((b, j, hd), device=vw.device, dtype=torch.float32)
        else:
            d_vw = torch.zeros((b, j, hd), device=vw.device, dtype=torch.float32) if need_v else None
        slot = _grad_slot(ctx.head_param) if need_h else None
        defer = DEFER_HEAD_FINISH and slot is not None
        work = None
        if need_h:
            work = _layer_workspace(slot, n_head) if defer else torch.zeros(n_head * 1024, device=vw.device, dtype=torch.float64)
            if defer:
                _defer_head_begin(work)
        rider = _dw_take(vw.device)                  # (a postponed weight-gradient job: nothing here carries it - run it now)
        if rider is not None:
            _dw_run(rider)
        sp, max_union = plan.fold_plan()[0], plan.fold_plan()[1]
        pw = w.pw16 if w.pw16 is not None else w.pw    # (bf16 math mode: the bf16 tiles)
        qw = (w.qw16 if w.qw16 is not None else pw) if w.pw16 is not None else (w.qw if w.qw is not None else w.pw)      # (read only for d(scale))
        rc = _lib.lib().pit_fold_att_bwd(ctypes.byref(sp), vw.data_ptr(), vw.stride(1), vw.stride(0), b, n_head, d, pw.data_ptr(),
                                         qw.data_ptr(), dz.data_ptr(), dz.stride(1), dz.stride(0),
                                         _lib.ptr(d_vw), hd, j * hd, _lib.ptr(work), _lib.ptr(tiles), _lib.ptr(rev_ptr),
                                         _lib.ptr(rev_ent), max_union, ctx.math | io, _lib.stream_ptr())
        _lib.check(rc, "pit_fold_att_bwd")
        d_head = None
        if need_h:
            flags = 1 | (4 if ctx.head_is_scale else 0)
            if defer:
                _defer_head_finish(work, slot, head, w.scale, n_head, flags)
            else:
                d_head = slot if slot is not None else torch.empty((n_head,), device=vw.device, dtype=torch.float32)
                _finish_heads_now(work, d_head, head, w.scale, n_head, flags if slot is not None else flags & ~1)
                if slot is not None:
                    d_head = None
        return d_vw, d_head, None, None, None, None, None, None, None


_TAIL_WS = {}       # (device index, stream) -> the thin tail's slotted partial sums (self-cleaning)


def _tail_scratch(device) -> torch.Tensor:
    key = _ws_key(device)
    ws = _TAIL_WS.get(key)
    if ws is None:
        ws = _TAIL_WS[key] = torch.zeros((_lib.lib().pit_thin_tail_scratch_floats(),), device=device, dtype=torch.float32)
        _pin(ws)
    return ws


class _ThinTail(torch.autograd.Function):
    """y = gelu(z + b1) @ W2^T + b2 for out_dim <= 4 (pit_thin_tail_fwd / _bwd): z fp32 or bf16, nothing but z is saved."""

    @staticmethod
    def forward(ctx, z, b1, w2, b2, params):
        _need_gpu_bf16_ok(z)
        _need_gpu(b1, w2, b2)
        z = _row_view(z)
        if z.stride(0) != z.shape[1] * z.stride(1) or z.stride(1) % 4 or z.data_ptr() % 16:
            z = z.contiguous()
        b, n, d = z.shape
        n2 = w2.shape[0]
        b1c, w2c, b2c = (t.detach().contiguous() for t in (b1, w2, b2))
        y = torch.empty((b, n, n2), device=z.device, dtype=torch.float32)
        ctx.math = _math_code() | (IO_X_BF16 if z.dtype == torch.bfloat16 else 0)
        rc = _lib.lib().pit_thin_tail_fwd(z.data_ptr(), z.stride(1), b * n, d, n2, b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(),
                                          y.data_ptr(), n2, ctx.math, _lib.stream_ptr())
        _lib.check(rc, "pit_thin_tail_fwd")
        ctx.params = params
        ctx.save_for_backward(z, b1c, w2c)
        return y

    @staticmethod
    def backward(ctx, d_y):
        z, b1c, w2c = ctx.saved_tensors
        b, n, d = z.shape
        n2 = w2c.shape[0]
        d_y = d_y.reshape(b * n, n2)
        if d_y.stride(1) != 1 or d_y.stride(0) < n2:
            d_y = d_y.contiguous()
        dz = torch.empty((b, n, d), device=z.device, dtype=z.dtype)
        slots = [_grad_slot(q) if isinstance(q, torch.nn.Parameter) else None for q in ctx.params]
        inplace = all(t is not None for t in slots) and all(ctx.needs_input_grad[1:4])
        if inplace:
            d_b1, d_w2, d_b2 = slots
        else:
            d_b1, d_w2, d_b2 = (torch.zeros_like(t) for t in (b1c, w2c, b1c[:n2]))
        rc = _lib.lib().pit_thin_tail_bwd(z.data_ptr(), z.stride(1), b * n, d, n2, b1c.data_ptr(), w2c.data_ptr(), d_y.data_ptr(),
                                          d_y.stride(0), dz.data_ptr(), d, d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(),
                                          _tail_scratch(z.device).data_ptr(), ctx.math, _lib.stream_ptr())
        _lib.check(rc, "pit_thin_tail_bwd")
        if inplace:
            return dz, None, None, None, None
        return dz, d_b1, d_w2, d_b2, None


@torch.compiler.disable
def fold_decoder_apply(values: torch.Tensor, lmda: torch.Tensor, plan: MeshPlan, n_head: int, mlp, use_fold_att: bool) -> torch.Tensor:
    """pit.decoder (pit.py:124-127) with de.mlp1 folded into the values (see the section comment); ``mlp`` = de's
    (w1, b1, w2, b2).  ``use_fold_att``: the fold attention launches (batch-free meshes, fold_att_supported) - else one head on
    the candidate-list / union-tile kernels of posatt_apply."""
    w1, b1, w2, b2 = mlp
    vw = _Linear.apply(values, w1, w1 if isinstance(w1, torch.nn.Parameter) else None)
    bf16 = get_math_mode() == "bf16" and BF16_STORAGE
    if use_fold_att:
        param = lmda if isinstance(lmda, torch.nn.Parameter) else None
        c = host_head_scale(lmda) if get_head_scale_route() == "host" else None
        need_q = torch.is_grad_enabled() and lmda.requires_grad        # (Q = P (m - mbar): read by d(scale) only)
        z = _FoldAtt.apply(vw, lmda.reshape(-1), plan, n_head, False, param, c, bf16, need_q)
    else:
        z = posatt_apply(vw, lmda, plan, n_head, concat=False, out_bf16=bf16)
    return _ThinTail.apply(z, b1, w2, b2, (b1, w2, b2))


class _Encoder(torch.autograd.Function):
    """pit.encoder (pit.py:108-112): posatt_cross_* (mesh_in -> mesh_ltt) + kaiming_mlp `en_layer` + gelu."""

    @staticmethod
    def forward(ctx, values, head, plan: MeshPlan, n_head: int, head_is_scale: bool, head_param, scale_in, params, coord_dims: int,
                concat_heads: int, wjob, clear, w1, b1, w2, b2, djob=None, need=True):
        _need_gpu(values, head, w1, b1, w2, b2)
        values = _row_view(values)
        b, j, dv = values.shape
        d = w1.shape[0]
        kd = int(coord_dims)
        k0 = n_head * (kd + dv)
        dev = values.device
        sp = plan.slab_plan()[0]
        head = head.detach().reshape(-1).contiguous()
        k_head, k_is_scale = (scale_in, True) if scale_in is not None else (head, head_is_scale)
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        rows = b * plan.n_out
        buf = None
        if concat_heads > 0:
            buf = torch.empty((rows, (1 + concat_heads) * d), device=dev, dtype=torch.float32)
            y = buf[:, :d]
        else:
            y = torch.empty((rows, d), device=dev, dtype=torch.float32)
        x = z1 = h = z2 = rowstat = scale = None
        if need:
            x = torch.empty((rows, k0), device=dev, dtype=torch.float32)
            z1 = torch.empty((rows, d), device=dev, dtype=torch.float32)
            h = torch.empty((rows, d), device=dev, dtype=torch.float32)
            z2 = torch.empty((rows, d), device=dev, dtype=torch.float32)
            rowstat = torch.empty((n_head, plan.n_out, 4), device=dev, dtype=torch.float32)
            scale = torch.empty((n_head,), device=dev, dtype=torch.float32)
        rc = _lib.lib().pit_encoder_fwd(ctypes.byref(sp), plan.mesh_in.data_ptr(), plan.sdim, kd, values.data_ptr(), values.stride(1),
                                        values.stride(0), dv, b, n_head, d, k_head.data_ptr(), 1 if k_is_scale else 0,
                                        w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(), _lib.ptr(x), _lib.ptr(z1),
                                        _lib.ptr(h), _lib.ptr(z2), y.data_ptr(), y.stride(0), _lib.ptr(rowstat), _lib.ptr(scale),
                                        _lib.ptr(clear), clear.numel() if clear is not None else 0,
                                        ctypes.cast(ctypes.pointer(wjob.job), ctypes.c_void_p) if wjob is not None else None,
                                        ctypes.cast(ctypes.pointer(djob.job), ctypes.c_void_p) if djob is not None else None,
                                        _lib.stream_ptr())
        _lib.check(rc, "pit_encoder_fwd")
        ctx.plan, ctx.n_head, ctx.head_is_scale, ctx.head_param, ctx.params, ctx.kd = plan, n_head, head_is_scale, head_param, params, kd
        ctx.keep = (values, head, w1c, w2c, x, z1, h, z2, rowstat, scale)
        out = y.reshape(b, plan.n_out, d)
        if buf is None:
            return out
        buf = buf.reshape(b, plan.n_out, (1 + concat_heads) * d)
        ctx.mark_non_differentiable(buf)
        ctx.set_materialize_grads(False)
        return out, buf

    @staticmethod
    def backward(ctx, d_y, _d_buf=None):
        values, head, w1, w2, x, z1, h, z2, rowstat, scale = ctx.keep
        plan, n_head, kd = ctx.plan, ctx.n_head, ctx.kd
        b, j, dv = values.shape
        d, rows = w1.shape[0], b * plan.n_out
        dev = values.device
        sp = plan.slab_plan()[0]
        if d_y is None:
            d_y = torch.zeros((rows, d), device=dev, dtype=torch.float32)
        d_y2 = d_y.reshape(rows, d)
        if d_y2.stride(1) != 1 or d_y2.stride(0) < d:
            d_y2 = d_y2.contiguous()
        scratch = torch.empty((rows * 2 * d,), device=dev, dtype=torch.float32)
        need_h = ctx.needs_input_grad[1]
        slot = _grad_slot(ctx.head_param) if need_h else None
        defer = DEFER_HEAD_FINISH and slot is not None
        # (the launch always reduces d(scale): accumulators of its own when lmda needs no gradient)
        work = _layer_workspace(slot, n_head) if defer else torch.zeros(n_head * 1024, device=dev, dtype=torch.float64)
        if defer:
            _defer_head_begin(work)
        rc = _lib.lib().pit_encoder_bwd(ctypes.byref(sp), plan.mesh_in.data_ptr(), plan.sdim, kd, values.data_ptr(), values.stride(1),
                                        values.stride(0), dv, b, n_head, d, scale.data_ptr(), rowstat.data_ptr(), w1.data_ptr(),
                                        w2.data_ptr(), z1.data_ptr(), z2.data_ptr(), d_y2.data_ptr(), d_y2.stride(0),
                                        scratch.data_ptr(), _lib.ptr(work), _lib.stream_ptr())
        _lib.check(rc, "pit_encoder_bwd")
        d_head = None
        if need_h:
            flags = 1 | (4 if ctx.head_is_scale else 0)
            if defer:
                _defer_head_finish(work, slot, head, scale, n_head, flags)
            else:
                d_head = slot if slot is not None else torch.empty((n_head,), device=dev, dtype=torch.float32)
                _finish_heads_now(work, d_head, head, scale, n_head, flags if slot is not None else flags & ~1)
                if slot is not None:
                    d_head = None
        grads, inplace = _mlp_grad_slots(ctx.params, (w1.shape, (d,), w2.shape, (d,)), ctx.needs_input_grad[12:16], dev)
        d_w1, d_b1, d_w2, d_b2 = grads
        k0 = n_head * (kd + dv)
        st = _lib.MlpParamsJob(x.data_ptr(), x.stride(0), rows, k0, d, d, h.data_ptr(), 1, d_y2.data_ptr(), d_y2.stride(0),
                               d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(), 1, scratch.data_ptr(), 0)
        keep = (x, h, d_y2, scratch, d_w1, d_b1, d_w2, d_b2)
        if inplace and MLP_PARAMS_RIDER and _dw_deferrable(rows, k0, d, d, 1, d_y2.stride(0)):
            _dw_defer(st, keep, dev)
        else:
            _dw_run((st, keep, torch.cuda.current_stream(dev)))
        w = (None, None, None, None) if inplace else (d_w1, d_b1, d_w2, d_b2)
        return (None, d_head, None, None, None, None, None, None, None, None, None, None) + w + (None, None)


@torch.compiler.disable
def encoder_apply(values: torch.Tensor, lmda: torch.Tensor, plan: MeshPlan, n_head: int, mlp, concat_heads: int = 0,
                  head_is_scale: bool = False, early=None) -> torch.Tensor:
    """pit.encoder (pit.py:108-112) as one launch per direction: gelu(en_layer(posatt_cross(mesh_ltt, mesh_in, values))); ``values``
    may be tagged by tag_coords (the coordinate channels then come from the mesh).  ``early``: early_block_weights(...) whose job
    this launch carries.  The inputs get no gradient (the caller checked that they need none)."""
    param = lmda if isinstance(lmda, torch.nn.Parameter) else None
    c = host_head_scale(lmda) if (not head_is_scale and get_head_scale_route() == "host") else None
    coords = getattr(values, "_pit_coords", None)
    kd = plan.sdim if coords is not None else 0
    wjob = None
    if early is not None and getattr(_STEP, "fwd_job", None) is early:
        _STEP.fwd_job = None
        wjob = early
    clear = getattr(_STEP, "clear", None)
    if clear is not None:
        _STEP.clear = None                       # (zeroed by this launch: the step's loss launch need not)
    djob = getattr(_STEP, "dec_job", None)       # the decoder's weights of this forward ride in this launch (early_decoder_weights)
    _STEP.dec_job = None
    res = _Encoder.apply(values.detach(), lmda.reshape(-1), plan, n_head, head_is_scale, param, c, tuple(mlp), kd, int(concat_heads),
                         wjob, clear, *mlp, djob,
                         torch.is_grad_enabled() and (lmda.requires_grad or any(t.requires_grad for t in mlp)))
    _STEP.dec_ready = djob
    if concat_heads <= 0:
        return res
    y, buf = res
    y._pit_concat = buf
    return y


class _RelLpLoss(torch.autograd.Function):
    """RelLpNorm (utils.py:80-98), optionally fused with the per-pixel affine
    de-normalisation of the prediction (utils.py:25-34).

    With ``unit_seed`` (the tensor of ones a training step seeds its backward pass with) the forward
    launch also writes the gradients for d loss = 1; backward returns them without a launch when
    it is indeed handed that seed, and falls back to the general backward kernel otherwise.
    ``clear`` is a tensor the same launch zeroes (the step's flat gradient accumulators)."""

    @staticmethod
    def forward(ctx, pred, true, out_dim: int, p: int, scale, shift, unit_seed=None, clear=None):
        _need_gpu(pred, true, scale, shift)
        b = true.size(0)
        t = true.reshape(b, -1, out_dim).contiguous()
        q = pred.reshape(b, -1, out_dim).contiguous()
        if t.shape != q.shape:
            raise RuntimeError(f"true {tuple(true.shape)} and pred {tuple(pred.shape)} do not match")
        npts = t.shape[1]
        sc = scale.reshape(npts, out_dim).contiguous() if scale is not None else None
        sh = shift.reshape(npts, out_dim).contiguous() if shift is not None else None
        norms = torch.empty((b, out_dim, 2), device=t.device, dtype=torch.float32)
        loss = torch.empty((), device=t.device, dtype=torch.float32)
        wkey = _ws_key(t.device) + (b * out_dim,)             # PIT_REL_LP_WS_FLOATS(batch, nch)
        ws = _LOSS_WS.get(wkey)
        if ws is None:
            ws = _LOSS_WS[wkey] = torch.zeros(4 + 5 * b * out_dim, device=t.device, dtype=torch.float32)
        if _capturing():
            _pin(ws)
        unit_p = unit_t = None
        if unit_seed is not None or clear is not None:
            if unit_seed is not None:
                unit_p = torch.empty_like(q) if pred.requires_grad else None
                unit_t = torch.empty_like(t) if true.requires_grad else None
            if clear is not None and (not clear.is_contiguous() or clear.dtype != torch.float32):
                raise RuntimeError("clear must be a contiguous fp32 tensor")
            rc = _lib.lib().pit_rel_lp_loss_fwd_grad(t.data_ptr(), q.data_ptr(), _lib.ptr(sc), _lib.ptr(sh), b, npts,
                                                     out_dim, int(p), norms.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                                     _lib.ptr(unit_p), _lib.ptr(unit_t), _lib.ptr(clear),
                                                     clear.numel() if clear is not None else 0, _lib.stream_ptr())
            _lib.check(rc, "pit_rel_lp_loss_fwd_grad")
        else:
            rc = _lib.lib().pit_rel_lp_loss_fwd(t.data_ptr(), q.data_ptr(), _lib.ptr(sc), _lib.ptr(sh), b, npts,
                                                out_dim, int(p), norms.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                                _lib.stream_ptr())
            _lib.check(rc, "pit_rel_lp_loss_fwd")
        ctx.meta = (b, npts, out_dim, int(p), pred.shape, true.shape)
        ctx.save_for_backward(t, q, norms, sc if sc is not None else norms, sh if sh is not None else norms)
        ctx.affine = sc is not None
        ctx.unit = (unit_seed.data_ptr() if unit_seed is not None else 0, unit_p, unit_t)
        return loss

    @staticmethod
    def backward(ctx, g):
        t, q, norms, sc, sh = ctx.saved_tensors
        b, npts, out_dim, p, shape, tshape = ctx.meta
        need_p, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_p or need_t):
            return None, None, None, None, None, None, None, None
        seed_ptr, unit_p, unit_t = ctx.unit
        if seed_ptr and g.data_ptr() == seed_ptr and (unit_p is not None or not need_p) and \
                (unit_t is not None or not need_t):
            # d loss is the step's tensor of ones: the forward launch already wrote these gradients
            return (unit_p.reshape(shape) if need_p else None), (unit_t.reshape(tshape) if need_t else None), \
                None, None, None, None, None, None
        g = g.contiguous()
        d_pred = torch.empty_like(q) if need_p else None
        d_true = torch.empty_like(t) if need_t else None
        rc = _lib.lib().pit_rel_lp_loss_bwd(t.data_ptr(), q.data_ptr(), sc.data_ptr() if ctx.affine else 0,
                                            sh.data_ptr() if ctx.affine else 0, b, npts, out_dim, p,
                                            norms.data_ptr(), g.data_ptr(), _lib.ptr(d_pred), _lib.ptr(d_true),
                                            _lib.stream_ptr())
        _lib.check(rc, "pit_rel_lp_loss_bwd")
        return (d_pred.reshape(shape) if need_p else None), (d_true.reshape(tshape) if need_t else None), \
            None, None, None, None, None, None


class _FusedLoss(torch.autograd.Function):
    """The loss node of a step whose decoder launches carry the loss (LossSpec): no launch in either direction.  The forward
    returns the scalar the decoder BACKWARD launch will write (a training step reads its loss after the pass); the backward hands
    a token tensor down the graph that pit.decoder's backward recognises by its address - it reaches it through views only.
    THE RETURNED TENSOR IS VALID ONLY AFTER backward() HAS RUN: until then it is uninitialised memory (initialising it would be a
    launch - the step has fourteen), so nothing may be derived from it in the forward (scaling, logging, a finite check).  Only
    engine.TrainStep / ops.step_state(loss=...) open this path, and they read the loss after the pass; every other caller of
    rel_lp_loss gets _RelLpLoss, whose value is computed in the forward."""

    @staticmethod
    def forward(ctx, pred, spec: LossSpec):
        ctx.spec, ctx.pred_shape = spec, pred.shape
        return spec.value

    @staticmethod
    def backward(ctx, g):
        spec = ctx.spec
        spec.seed = g.contiguous()
        spec.token = torch.empty_like(spec.pred)           # never written, never read: its address is the message
        return spec.token.view(ctx.pred_shape), None


def _take_fused_loss(true, pred, out_dim: int, p: int, scale, shift):
    """The LossSpec the running step's decoder accumulated for exactly this loss call, or None."""
    spec = getattr(_STEP, "loss_issued", None)
    if spec is None or spec.pred is None:
        return None
    _STEP.loss_issued = None
    same = lambda a, b_: (a is None and b_ is None) or (a is not None and b_ is not None and a.data_ptr() == b_.data_ptr())
    npts = spec.true.shape[1]
    sc = scale.reshape(npts, out_dim) if scale is not None and scale.numel() == npts * out_dim else scale
    sh = shift.reshape(npts, out_dim) if shift is not None and shift.numel() == npts * out_dim else shift
    ok = (int(p) == spec.p and int(out_dim) == spec.out_dim and pred.requires_grad and pred.is_contiguous()
          and pred.data_ptr() == spec.pred.data_ptr() and pred.numel() == spec.pred.numel()
          and true.numel() == spec.true.numel() and true.is_contiguous() and true.data_ptr() == spec.true.data_ptr()
          and (sc is None or sc.is_contiguous()) and (sh is None or sh.is_contiguous())
          and same(sc, spec.scale) and same(sh, spec.shift))
    return spec if ok else None


@torch.compiler.disable
def rel_lp_loss(true, pred, out_dim: int, p: int, pred_scale=None, pred_shift=None, unit_seed=None,
                clear=None) -> torch.Tensor:
    """sum_b mean_c ||true - pred'||_p / ||true||_p with pred' = pred*pred_scale + pred_shift.
    ``unit_seed`` / ``clear``: see _RelLpLoss (used by engine.TrainStep to save two launches).  Inside a step whose fused decoder
    already accumulated this very loss (step_state(loss=...)), no launch at all: see _FusedLoss."""
    spec = _take_fused_loss(true, pred, out_dim, p, pred_scale, pred_shift)
    if spec is not None:
        if clear is not None:
            clear.zero_()
        return _FusedLoss.apply(pred, spec)
    return _RelLpLoss.apply(pred, true, out_dim, p, pred_scale, pred_shift, unit_seed, clear)


_RELMAX_WS = {}


@torch.compiler.disable
def rel_max_norm(true: torch.Tensor, pred: torch.Tensor, out_dim: int) -> torch.Tensor:
    """RelMaxNorm (utils.py:59-77) on device: sum_b mean_c max|true-pred| / max|true|.  Forward only -
    the scripts use it as an evaluation metric under no_grad (train_burgers.py:80, train_naca.py:132)."""
    _need_gpu(true, pred)
    if torch.is_grad_enabled() and (true.requires_grad or pred.requires_grad):
        raise NotImplementedError("RelMaxNorm on device tensors is forward-only (an evaluation metric in the reference "
                                  "scripts): call it under torch.no_grad() or on detached tensors")
    b = true.size(0)
    t = true.detach().reshape(b, -1, out_dim).contiguous()
    q = pred.detach().reshape(b, -1, out_dim).contiguous()
    if t.shape != q.shape:
        raise RuntimeError(f"true {tuple(true.shape)} and pred {tuple(pred.shape)} do not match")
    wkey = _ws_key(t.device)
    ws = _RELMAX_WS.get(wkey)
    if ws is None:
        ws = _RELMAX_WS[wkey] = torch.zeros(2, device=t.device, dtype=torch.float64)
    if _capturing():
        _pin(ws)
    out = torch.empty((), device=t.device, dtype=torch.float32)
    _lib.check(_lib.lib().pit_rel_max_norm(t.data_ptr(), q.data_ptr(), b, t.shape[1], out_dim, out.data_ptr(),
                                           ws.data_ptr(), _lib.stream_ptr()), "pit_rel_max_norm")
    return out


class _InstanceNorm(torch.autograd.Function):
    """nn.InstanceNorm1d over the point axis on the (batch, points, channels) layout
    (train_vorticity.py:43,56,59), no permutes."""

    @staticmethod
    def forward(ctx, x, eps: float):
        _need_gpu(x)
        x = _row_view(x)
        b, npts, nch = x.shape
        y = torch.empty((b, npts, nch), device=x.device, dtype=torch.float32)
        rstd = torch.empty((b, nch), device=x.device, dtype=torch.float32)
        rc = _lib.lib().pit_instance_norm_fwd(x.data_ptr(), x.stride(1), x.stride(0), b, npts, nch, float(eps),
                                              y.data_ptr(), rstd.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "pit_instance_norm_fwd")
        ctx.save_for_backward(y, rstd)
        return y

    @staticmethod
    def backward(ctx, d_y):
        y, rstd = ctx.saved_tensors
        b, npts, nch = y.shape
        d_y = d_y.contiguous()
        d_x = torch.empty_like(y)
        rc = _lib.lib().pit_instance_norm_bwd(d_y.data_ptr(), y.data_ptr(), rstd.data_ptr(), b, npts, nch,
                                              d_x.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "pit_instance_norm_bwd")
        return d_x, None


@torch.compiler.disable
def instance_norm_points(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """(x - mean over points) / sqrt(var over points + eps) per (sample, channel) of a (b, L, C) tensor."""
    return _InstanceNorm.apply(x, eps)


MATH_MODES = {"fp32": 0, "bf16": 1}      # PIT_MATH_* of include/pit_hip.h
_MATH = threading.local()


def _math_code() -> int:
    return MATH_MODES[getattr(_MATH, "mode", "fp32")]


def set_math_mode(mode: str) -> None:
    """'fp32' (default, the reference's arithmetic) or 'bf16' (bf16 MFMA operands, fp32 accumulate) for
    the attention / MLP contractions (the small-regime fused MLP kernels contract in fp32 in both modes - they
    are latency-bound - and the fused processor blocks run in fp32 mode only: bf16 mode takes the per-layer path).
    In bf16 mode the decoder tail is also STORED as bf16 when its shape allows (BF16_STORAGE, pit.decoder).  Host-side, per Python thread: the mode is an ARGUMENT of every
    C-ABI call (the library keeps no mode of its own), read when an operator's forward runs and reused
    by its backward; a captured hipGraph keeps the mode its launches were captured with."""
    if mode not in MATH_MODES:
        raise ValueError(f"math mode must be one of {sorted(MATH_MODES)}, got {mode!r}")
    _MATH.mode = mode


def get_math_mode() -> str:
    return getattr(_MATH, "mode", "fp32")


class math_mode:
    """`with ops.math_mode('bf16'): ...` - scoped set_math_mode."""

    def __init__(self, mode: str):
        self.mode, self.prev = mode, None

    def __enter__(self):
        self.prev = get_math_mode()
        set_math_mode(self.mode)
        return self

    def __exit__(self, *exc):
        set_math_mode(self.prev)
        return False


def head_scale(lmda: torch.Tensor) -> torch.Tensor:
    """c = tan(0.25*pi*(1-1e-7)*(1+sin(lmda))) on device (pit.py:48)."""
    _need_gpu(lmda)
    flat = lmda.detach().reshape(-1).contiguous()
    out = torch.empty_like(flat)
    _lib.check(_lib.lib().pit_head_scale(flat.data_ptr(), flat.numel(), out.data_ptr(), _lib.stream_ptr()),
               "pit_head_scale")
    return out.reshape(lmda.shape)


def debug_mfma_tile(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """D = A(32x8) @ B(8x32) through the kernels' MFMA fragment maps (layout self-test)."""
    _need_gpu(a, b)
    d = torch.empty((32, 32), device=a.device, dtype=torch.float32)
    _lib.check(_lib.lib().pit_debug_mfma_tile(a.contiguous().data_ptr(), b.contiguous().data_ptr(), d.data_ptr(),
                                              _lib.stream_ptr()), "pit_debug_mfma_tile")
    return d


_ = ctypes  # keep the import explicit: pointers cross the ABI as plain integers
