"""CPU (no GPU): the C-ABI library loads and exports exactly what include/pit_hip.h declares;
the drop-in modules present the reference's API (names, ctor signatures, state_dict keys,
seed-for-seed initialisation); host-side helpers agree with the oracle."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

import golden_io as gio
import pit_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pit_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pit_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("pit_version", "pit_select_fwd", "pit_posatt_fwd", "pit_posatt_bwd", "pit_mlp_fwd", "pit_mlp_bwd"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from position_induced_transformer_amd import _lib, build
    build.build()                                   # no-op when up to date; hipcc cross-compiles without a GPU
    assert os.path.exists(_lib.LIB_PATH)
    handle = _lib.lib()                             # resolves every name in _lib.SIGNATURES
    for name in declared_functions():
        assert hasattr(handle, name), f"{name} declared in pit_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared_functions()
    assert handle.pit_version() >= 2
    assert b"NULL" in handle.pit_error_string(-1) and handle.pit_error_string(0) == b"ok"


def test_ctypes_signatures_match_header_arity():
    from position_induced_transformer_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(argtypes), (name, len(args), len(argtypes))


def test_ops_fail_loudly_without_gpu_tensors():
    from position_induced_transformer_amd import ops
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.mlp_apply(torch.zeros(3, 4), torch.zeros(5, 4), torch.zeros(5), torch.zeros(2, 5), torch.zeros(2))
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.MeshPlan("euclid", torch.zeros(4, 2), torch.zeros(4, 2), 0.5, False)


# ----------------------------------------------------------------------------- drop-in API
EXPORTS = ["kaiming_mlp", "posatt", "posatt_cross", "pit", "posatt_fixed", "posatt_cross_fixed", "pit_fixed",
           "posatt_periodic1d", "posatt_cross_periodic1d", "pit_periodic1d", "posatt_periodic2d",
           "posatt_cross_periodic2d", "pit_periodic2d", "torch", "nn", "gelu", "np", "pi"]


def test_star_exports_match_reference_names():
    ns = {}
    exec("from position_induced_transformer_amd.pit import *", ns)
    for name in EXPORTS:                       # scripts use torch/np/gelu/pi without importing them
        assert name in ns, name
    ns = {}
    exec("from position_induced_transformer_amd.utils import *", ns)
    for name in ("PixelWiseNormalization", "count_params", "RelMaxNorm", "RelLpNorm", "F", "reduce", "operator", "torch"):
        assert name in ns, name


def test_constructor_signatures():
    from position_induced_transformer_amd import pit as P
    assert list(inspect.signature(P.kaiming_mlp.__init__).parameters)[1:] == ["n_filters0", "n_filters1", "n_filters2"]
    for cls in (P.posatt, P.posatt_cross, P.posatt_fixed, P.posatt_cross_fixed, P.posatt_periodic1d,
                P.posatt_cross_periodic1d, P.posatt_periodic2d, P.posatt_cross_periodic2d):
        assert list(inspect.signature(cls.__init__).parameters)[1:] == ["n_head", "in_dim", "locality"]
    for cls in (P.pit, P.pit_fixed, P.pit_periodic1d, P.pit_periodic2d):
        assert list(inspect.signature(cls.__init__).parameters)[1:] == [
            "space_dim", "in_dim", "out_dim", "hid_dim", "n_head", "n_blocks", "mesh_ltt", "en_loc", "de_loc"]
        assert not hasattr(cls, "forward") or cls.forward is torch.nn.Module.forward   # subclasses supply forward
        for meth in ("encoder", "processor", "decoder"):
            assert callable(getattr(cls, meth))


@pytest.mark.parametrize("tag", ["fixed", "base", "p1d"])
def test_state_dict_and_seeded_init_match_reference(tag):
    """Same parameter names/shapes and - given the same torch seed - the same initial values as
    the reference (including the extra RNG draws of pit_fixed/periodic re-creating layers)."""
    from position_induced_transformer_amd import pit as P
    fx = gio.load("F13_init_parity")
    build = {"fixed": lambda: P.pit_fixed(2, 1, 1, 64, 2, 4, orc.grid_mesh_2d(16), 0.02, 0.02),
             "base": lambda: P.pit(2, 12, 1, 32, 2, 2, None, 0.05, 0.05),
             "p1d": lambda: P.pit_periodic1d(1, 1, 1, 32, 2, 3, orc.line_mesh_1d(64), 0.02, 0.02)}[tag]
    torch.manual_seed(0)
    model = build()
    sd = model.state_dict()
    assert list(sd.keys()) == [str(s) for s in fx[tag + "/names"]]
    assert "mesh_ltt" not in sd                                   # plain attribute, as in the reference
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(int(x) for x in fx[f"{tag}/shape/{k}"]), k
        assert np.array_equal(v.flatten()[:4].numpy(), fx[f"{tag}/head/{k}"]), k
        assert abs(float(v.double().sum()) - float(fx[f"{tag}/sum/{k}"])) <= 1e-9 * max(1.0, abs(float(fx[f"{tag}/sum/{k}"]))), k


def test_param_counts_match_survey():
    from position_induced_transformer_amd import pit as P, utils
    m = P.pit_fixed(2, 1, 1, 64, 2, 4, torch.zeros(16, 16, 2), 0.02, 0.02)
    assert utils.count_params(m) == 78989                         # SURVEY 8(a10) Darcy
    m = P.pit_periodic1d(1, 1, 1, 64, 2, 5, torch.zeros(256, 1), 0.02, 0.02)
    assert utils.count_params(m) == 95375                         # Burgers
    assert m.mesh_ltt.shape == (256, 1)


def test_task_subclass_pattern_of_the_scripts():
    """train_elasticity.py:39 overwrites en_layer after super().__init__; attributes the scripts read exist."""
    from position_induced_transformer_amd import tasks
    m = tasks.pit_elasticity(2, 44, 1, 256, 2, 4, None, 0.02, 0.02)
    assert m.en_layer.mlp1.in_features == 88 and m.mesh_ltt is None
    assert sum(p.numel() for p in m.parameters()) == 1270797      # SURVEY 8(a10) Elasticity
    for attr in ("space_dim", "in_dim", "out_dim", "hid_dim", "n_head", "n_blocks"):
        assert hasattr(m, attr)


# ----------------------------------------------------------------------------- host logic
@pytest.mark.parametrize("q,n", [(0.02, 1849), (0.02, 256), (0.02, 4096), (0.02, 972), (0.02, 1024), (0.02, 728),
                                 (0.02, 120), (1.0, 256), (0.3, 131), (0.05, 100), (0.02, 2), (0.01, 4390)])
def test_quantile_rank_matches_oracle_and_torch(q, n):
    from position_induced_transformer_amd import ops
    k, w = ops.quantile_rank(q, n)
    ko, wo = orc.quantile_rank(q, n)
    assert k == ko and np.float32(w) == np.float32(wo)
    x = torch.from_numpy(gio.synth((7, n), 5)).abs()
    srt = torch.sort(x, dim=-1).values
    a, b = srt[:, k], srt[:, min(k + 1, n - 1)]
    assert torch.equal(orc.lerp_threshold(a, b, w), torch.quantile(x, q, dim=-1))


def test_rel_lp_norm_cpu_path_matches_reference_formula():
    from position_induced_transformer_amd import utils
    t, p = torch.from_numpy(gio.synth((3, 20, 2), 1)), torch.from_numpy(gio.synth((3, 20, 2), 2))
    for order in (1, 2):
        assert torch.equal(utils.RelLpNorm(2, order)(t, p), orc.rel_lp_loss(t, p, 2, order))
    n = utils.PixelWiseNormalization(torch.from_numpy(gio.synth((6, 5, 5, 1), 3)))
    x = torch.from_numpy(gio.synth((2, 5, 5, 1), 4))
    assert torch.allclose(n.denormalize(n.normalize(x)), x, atol=1e-6)
    sc, sh = n.affine()
    assert torch.allclose(x * sc + sh, n.denormalize(x))


def test_reference_checkpoint_loader_strips_compile_prefix(tmp_path):
    """train_darcy.py:150 saves {'model_state': state_dict} of a torch.compile-wrapped model:
    keys start with '_orig_mod.'."""
    from position_induced_transformer_amd import pit as P, utils
    torch.manual_seed(3)
    src = P.pit_fixed(2, 1, 1, 16, 2, 2, torch.zeros(4, 4, 2), 0.1, 0.1)
    ckpt = {"model_state": {"_orig_mod." + k: v.clone() for k, v in src.state_dict().items()}}
    path = tmp_path / "model.pth"
    torch.save(ckpt, path)
    dst = P.pit_fixed(2, 1, 1, 16, 2, 2, torch.zeros(4, 4, 2), 0.1, 0.1)
    res = utils.load_reference_checkpoint(dst, str(path))
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v)
