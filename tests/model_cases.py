"""Model-level parity cases shared by oracle/make_golden.py (generator) and the tests.

Each case fixes a PiT configuration (hyper-parameters cited from the reference's
train scripts), its meshes and its synthetic inputs.  Everything is rebuilt from
integer-hash seeds (tests/golden_io.synth), so only expected outputs are stored.
"""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

import golden_io as gio          # noqa: E402
import pit_oracle as orc         # noqa: E402

CASES = ("F9_model_darcy", "F10_model_cloud", "F11_model_burgers", "F12_model_p2d")


def _t(shape, seed, lo=None, hi=None):
    return torch.from_numpy(gio.synth(shape, seed, lo, hi))


def build_case(name: str) -> dict:
    if name == "F9_model_darcy":          # train_darcy.py:103-111, b=2
        cfg = dict(space_dim=2, in_dim=1, out_dim=1, hid_dim=64, n_head=2, n_blocks=4, en_loc=0.02, de_loc=0.02)
        g43, g16 = orc.grid_mesh_2d(43).reshape(43, 43, 2), orc.grid_mesh_2d(16).reshape(16, 16, 2)
        return dict(cfg=cfg, kind="fixed", metric="euclid", mesh_ltt=g16, mesh_in=g43, mesh_out=g43,
                    func_in=_t((2, 43, 43, 1), 101), target=_t((2, 43, 43, 1), 102), p_norm=2,
                    shapes=orc.param_shapes(2, 1, 1, 64, 2, 4))
    if name == "F11_model_burgers":       # train_burgers.py:64-72, b=2, RelLp p=1
        cfg = dict(space_dim=1, in_dim=1, out_dim=1, hid_dim=64, n_head=2, n_blocks=5, en_loc=0.02, de_loc=0.02)
        l1024, l256 = orc.line_mesh_1d(1024), orc.line_mesh_1d(256)
        return dict(cfg=cfg, kind="fixed", metric="periodic1d", mesh_ltt=l256, mesh_in=l1024, mesh_out=l1024,
                    func_in=_t((2, 1024, 1), 103), target=_t((2, 1024, 1), 104), p_norm=1,
                    shapes=orc.param_shapes(1, 1, 1, 64, 2, 5))
    if name == "F12_model_p2d":           # vorticity-like (train_vorticity.py:98-106) reduced: 32^2 -> 8^2
        cfg = dict(space_dim=2, in_dim=3, out_dim=1, hid_dim=32, n_head=2, n_blocks=2, en_loc=0.05, de_loc=0.05)
        p32 = orc.grid_mesh_2d(32, False).reshape(32, 32, 2)
        p8 = orc.grid_mesh_2d(8, False).reshape(8, 8, 2)
        return dict(cfg=cfg, kind="fixed", metric="periodic2d", mesh_ltt=p8, mesh_in=p32, mesh_out=p32,
                    func_in=_t((2, 32, 32, 3), 105), target=_t((2, 32, 32, 1), 106), p_norm=2,
                    shapes=orc.param_shapes(2, 3, 1, 32, 2, 2))
    if name == "F10_model_cloud":         # elasticity-like (train_elasticity.py:67-75) reduced cloud
        cfg = dict(space_dim=2, in_dim=12, out_dim=1, hid_dim=64, n_head=2, n_blocks=4, en_loc=0.05, de_loc=0.05)
        cloud = _t((2, 243, 2), 107, 0.0, 1.0)
        feats = torch.cat((cloud, _t((2, 243, 10), 108, -1.0, 4.0)), -1)
        return dict(cfg=cfg, kind="cloud", metric="euclid", mesh_ltt=None, mesh_in=cloud, mesh_out=cloud,
                    func_in=feats, target=_t((2, 243, 1), 109), p_norm=2,
                    shapes=orc.param_shapes(2, 12, 1, 64, 2, 4, en_in=2 * 12))
    raise KeyError(name)
