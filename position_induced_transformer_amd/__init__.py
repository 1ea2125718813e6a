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
# What is recorded: GRAPH_PACKET_CAPTURE_OFF is True only if the variable was "0" BEFORE the runtime read
# its flags; engine.TrainStep.capture() refuses to capture a step that builds per-sample plans when it is False
# instead of risking the fault.  (The variable is process-wide: it also applies to any other hipGraph user in the
# process; measured cost of the slower path: none at these graph sizes.)
import torch as _torch

_hip_up = _torch.cuda.is_initialized()
if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in _os.environ:
    # set by the user / the harness: in time unless the process exported it after its first HIP call (not knowable here)
    GRAPH_PACKET_CAPTURE_OFF = _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0"
elif _hip_up:                                 # too late: the HIP runtime has already read its flags
    import warnings as _warnings
    _warnings.warn("position_induced_transformer_amd was imported after the GPU was initialised, so it cannot switch "
                   "off ROCm's hipGraph packet capture (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0): engine.TrainStep.capture() "
                   "will refuse steps on per-sample meshes (Elasticity/NACA), which may fault on replay.  Import this "
                   "package (or set the variable) before the first .cuda() call.")
    GRAPH_PACKET_CAPTURE_OFF = False
else:
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    GRAPH_PACKET_CAPTURE_OFF = True

from . import ops  # noqa: F401,E402

__version__ = "0.1.0"


def __getattr__(name):
    # `pit`, `utils`, `tasks`, `ddp`, `engine` load on first use: importing the drop-in module `pit` reproduces the
    # reference's import-time side effects (pit.py:1-11: global reseed, cudnn flags, matmul precision), which must not
    # fire for code that only wants `ops` or `utils`
    if name in ("pit", "utils", "tasks", "ddp", "engine", "build"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
