"""ctypes binding of csrc/libpit_hip.so (C ABI declared in include/pit_hip.h).

There is no fallback: if the library is missing or a symbol is absent this raises, and
every op in this package goes through it.  torch is imported first so that the HIP
runtime already loaded by PyTorch-ROCm is the one the library binds to."""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL load)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIT_LIB_PATH") or os.path.join(_HERE, "csrc", "libpit_hip.so")   # (override: diagnostic builds, tools/)

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float

class MlpParamsJob(ctypes.Structure):
    """pit_mlp_params_job of include/pit_hip.h (the argument list of pit_mlp_bwd_params)."""
    _fields_ = [("x", _P), ("ldx", _L), ("rows", _I), ("n0", _I), ("n1", _I), ("n2", _I), ("h", _P), ("out_gelu", _I),
                ("d_y", _P), ("ld_dy", _L), ("d_w1", _P), ("d_b1", _P), ("d_w2", _P), ("d_b2", _P),
                ("accumulate", _I), ("scratch", _P), ("math_mode", _I)]


class BlockWeightsJob(ctypes.Structure):
    """pit_block_weights_job of include/pit_hip.h (the argument list of pit_block_weights)."""
    _fields_ = [("mesh", _P), ("n_pts", _I), ("space_dim", _I), ("metric", _I), ("period", _F),
                ("n_layers", _I), ("heads", _P), ("head_is_scale", _I), ("n_head", _I),
                ("e", _P), ("q", _P), ("inv", _P), ("rowstat", _P), ("scale_out", _P)]


class SlabPlan(ctypes.Structure):
    """pit_slab_plan of include/pit_hip.h (the static per-slab plan of a masked cross-attention layer on a fixed mesh pair)."""
    _fields_ = [("n_out", _I), ("n_in", _I), ("cap", _I), ("n_slabs", _I), ("umax", _I), ("stats", _P), ("rank_w", _F),
                ("idx", _P), ("cnt", _P), ("m", _P), ("slot", _P), ("keys", _P), ("nkeys", _P), ("rows", _I)]


class DecoderWeightsJob(ctypes.Structure):
    """pit_decoder_weights_job of include/pit_hip.h (the argument list of pit_decoder_weights)."""
    _fields_ = [("plan", _P), ("head", _P), ("head_is_scale", _I), ("n_head", _I), ("max_union", _I), ("max_count", _I),
                ("pw", _P), ("qw", _P), ("scale_out", _P), ("w1", _P), ("w1f", _P), ("dim", _I)]


# name -> argtypes, mirrors include/pit_hip.h one to one
SIGNATURES = {
    "pit_version": [],
    "pit_error_string": [_I],
    "pit_head_scale": [_P, _I, _P, _P],
    "pit_select_fwd": [_P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P, _P],
    "pit_plan_fwd": [_P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P],
    "pit_lists_transpose": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "pit_neighbors_fwd": [_P, _P, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P, _P, _P, _P],
    "pit_posatt_fwd": [_P, _P, _I, _I, _I, _I, _I, _F,
                       _P, _I, _I, _L, _L,
                       _P, _I, _I,
                       _P, _F, _I, _I,
                       _P, _L, _L, _I, _I,
                       _P, _P, _P, _P, _I, _I, _I, _P],
    "pit_posatt_fwd_job": [_P, _P, _I, _I, _I, _I, _I, _F,
                           _P, _I, _I, _L, _L,
                           _P, _I, _I,
                           _P, _F, _I, _I,
                           _P, _L, _L, _I, _I,
                           _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "pit_posatt_bwd": [_P, _P, _I, _I, _I, _I, _I, _F,
                       _P, _I, _I, _L, _L,
                       _P, _I, _I, _P,
                       _P, _I,
                       _P, _L, _L, _I,
                       _P, _L, _L, _I,
                       _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _P],
    "pit_posatt_dhead_finish": [_I, _P, _P, _P, _P, _P, _P, _P, _P],
    "pit_block_supported": [_I, _I, _I, _I],
    "pit_block_weights": [_P, _I, _I, _I, _F, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "pit_block_fwd": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _P],
    "pit_slab_plan_build": [_P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "pit_fold_supported": [_I, _I, _I, _I, _I],
    "pit_fold_weights": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "pit_fold_att_fwd": [_P, _P, _L, _L, _I, _I, _I, _P, _P, _L, _L, _I, _I, _P],
    "pit_fold_att_bwd": [_P, _P, _L, _L, _I, _I, _I, _P, _P, _P, _L, _L, _P, _L, _L, _P, _P, _P, _P, _I, _I, _P],
    "pit_thin_tail_scratch_floats": [],
    "pit_thin_tail_fwd": [_P, _L, _I, _I, _I, _P, _P, _P, _P, _L, _I, _P],
    "pit_thin_tail_bwd": [_P, _L, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _I, _P],
    "pit_satt_supported": [_I, _I, _I, _I, _I],
    "pit_satt_fwd": [_P, _I, _I, _I, _I, _F, _P, _L, _L, _I, _I, _P, _I, _I, _P, _P, _L, _L, _I, _I, _P, _P, _P, _I, _P],
    "pit_satt_bwd": [_P, _I, _I, _I, _I, _F, _I, _I, _P, _I, _P, _P, _P, _P, _L, _L, _I, _P, _L, _L, _I, _P, _P, _I, _P, _P],
    "pit_satt_tiles_elems": [_I],
    "pit_mlp_chain_supported": [_I, _I, _I, _I],
    "pit_mlp_chain_fwd": [_P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "pit_mlp_chain_bwd": [_I, _I, _I, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _I, _I, _P],
    "pit_cast_bf16_multi": [_I, _P, _P, _P, _P],
    "pit_linear_fwd": [_P, _L, _I, _I, _I, _P, _P, _P, _L, _I, _P],
    "pit_linear_bwd": [_P, _L, _I, _I, _I, _P, _P, _L, _P, _L, _P, _I, _I, _P],
    "pit_edge_supported": [_I, _I, _I, _I],
    "pit_decoder_weights": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "pit_decoder_fwd": [_P, _P, _L, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _L,
                        _P, _P, _P, _I, _P, _I, _P],
    "pit_decoder_bwd": [_P, _P, _L, _L, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _L, _P, _P, _L, _P,
                        _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "pit_union_att_supported": [_I, _I, _I, _I],
    "pit_union_att_fwd": [_P, _P, _L, _L, _I, _I, _I, _P, _P, _L, _L, _I, _I, _P],
    "pit_union_att_bwd": [_P, _P, _L, _L, _I, _I, _I, _P, _P, _P, _L, _L, _I, _P, _L, _L, _P, _I, _P],
    "pit_encoder_fwd": [_P, _P, _I, _I, _P, _L, _L, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P,
                        _P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _P, _P, _P],
    "pit_encoder_bwd": [_P, _P, _I, _I, _P, _L, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P],
    "pit_posatt_pre_supported": [_I, _I, _I, _I],
    "pit_posatt_pre_fwd": [_P, _P, _I, _I, _I, _I, _P, _L, _L, _P, _L, _L, _I, _I, _I, _P],
    "pit_posatt_pre_bwd": [_P, _P, _P, _I, _I, _I, _I, _P, _L, _L, _P, _L, _L, _I, _P, _L, _L, _I, _P, _I, _P],
    "pit_block_bwd": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _L, _P, _P, _L, _P, _P, _I, _P],
    "pit_mlp_bf16_io_supported": [_I, _I, _I, _I, _I],
    "pit_mlp_fwd": [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _P],
    "pit_mlp_bwd": [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _L,
                    _P, _L, _P, _P, _P, _P, _I, _P, _I, _P],
    "pit_mlp_bwd_data": [_I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _L, _P, _L, _P, _I, _P],
    "pit_mlp_bwd_params": [_P, _L, _I, _I, _I, _I, _P, _I, _P, _L, _P, _P, _P, _P, _I, _P, _I, _P],
    "pit_mlp_bwd_params_deferrable": [_I, _I, _I, _I, _I, _L],
    "pit_rel_lp_loss_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "pit_rel_lp_loss_fwd_grad": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P],
    "pit_rel_lp_loss_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "pit_rel_max_norm": [_P, _P, _I, _I, _I, _P, _P, _P],
    "pit_instance_norm_fwd": [_P, _L, _L, _I, _I, _I, _F, _P, _P, _P],
    "pit_instance_norm_bwd": [_P, _P, _P, _I, _I, _I, _P, _P],
    "pit_adam_step": [_P, _P, _P, _P, _L, _P, _F, _F, _I, _F, _F, _F, _F, _I, _P, _P],
    "pit_debug_mfma_tile": [_P, _P, _P, _P],
}

LONG_RETURN = {"pit_satt_tiles_elems"}
ABI_VERSION = 23       # PIT_ABI_VERSION of include/pit_hip.h this binding was written against

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m position_induced_transformer_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the PiT hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError if the ABI lost a symbol
            fn.argtypes = argtypes
            fn.restype = ctypes.c_char_p if name == "pit_error_string" else (_L if name in LONG_RETURN else _I)
        got = handle.pit_version()           # the default library and a PIT_LIB_PATH override alike
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} implements PIT_ABI_VERSION {got}, this package binds version {ABI_VERSION}: "
                               "rebuild it (python -m position_induced_transformer_amd.build)")
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().pit_error_string(code)
        raise RuntimeError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
