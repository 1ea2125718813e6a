"""MI355X-native PiT position-attention hot path (gfx950 HIP kernels behind a C ABI).

``position_induced_transformer_amd.pit`` / ``.utils`` mirror the reference's modules;
``.ops`` holds the operator-level API, ``.tasks`` the task forward wrappers and
``.ddp`` the flat-gradient data-parallel helper."""
import os as _os

# ROCm 7.2's hipGraph "packet capture" fast path (pre-built AQL packets) corrupts a replayed graph that
# builds the per-sample selection plans, runs the forward AND the backward of a masked layer in ONE
# graph, once any other device work ran between two replays: the second replay dies with a GPU memory
# fault (reproduced with tools/graph_replay_repro.py; the same graph is correct with the flag off, on
# the eager path, and when the plan is built outside the graph).  The runtime reads its flags at its
# first HIP call, so setting the variable here - before this package touches the device - is enough.
# Measured cost of turning it off: none (Darcy b=8 0.3272 -> 0.3235 ms/step, NACA b=20 2.751 -> 2.741).
if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in _os.environ:
    import torch as _torch
    if _torch.cuda.is_initialized():          # too late: the HIP runtime has already read its flags
        import warnings as _warnings
        _warnings.warn("position_induced_transformer_amd was imported after the GPU was initialised, so it cannot switch "
                       "off ROCm's hipGraph packet capture (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0): hipGraph-captured "
                       "training steps on per-sample meshes (engine.TrainStep with Elasticity/NACA) may fault on "
                       "replay.  Import this package (or set the variable) before the first .cuda() call.")
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"

from . import ops, pit, utils  # noqa: F401

__version__ = "0.1.0"
