"""MI355X-native PiT position-attention hot path (gfx950 HIP kernels behind a C ABI).

``position_induced_transformer_amd.pit`` / ``.utils`` mirror the reference's modules;
``.ops`` holds the operator-level API, ``.tasks`` the task forward wrappers and
``.ddp`` the flat-gradient data-parallel helper."""
from . import ops, pit, utils  # noqa: F401

__version__ = "0.1.0"
