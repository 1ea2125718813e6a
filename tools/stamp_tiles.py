"""Diagnostic: phase timeline of posatt_rows_tiles (one wave of one workgroup) from in-kernel shader-clock stamps.
Needs a library built with -DPIT_STAMPS:  PIT_EXTRA_FLAGS=-DPIT_STAMPS python -m position_induced_transformer_amd.build"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib, ops, tasks
model, _, _ = tasks.make_task("darcy", seed=0)
layer, mesh = model.conv[0], model.mesh_ltt
plan = layer._plan(mesh, mesh, True)
u = torch.randn(256, 256, 64, device="cuda")
with torch.no_grad():
    for _ in range(20):
        ops.posatt_apply(u, layer.lmda, plan, 2, True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L = _lib.lib()
L.pit_debug_read_stamps.argtypes = [ctypes.c_void_p]
assert L.pit_debug_read_stamps(buf) == 0
t = list(buf)
names = {0: "entry", 1: "prologue done (row constants, column setup)", 20: "key loop done", 21: "epilogue done"}
for p in range(4):
    names[2 + 4 * p] = f"pass {p}: previous chunk consumed (barrier)"
    names[3 + 4 * p] = f"pass {p}: key coordinates staged"
    names[4 + 4 * p] = f"pass {p}: this wave's weights done"
    names[5 + 4 * p] = f"pass {p}: all weights in LDS (barrier)"
prev = t[0]
for i in sorted(names):
    if t[i]:
        print(f"{names[i]:50s} +{t[i] - prev:8d} cycles   (t = {t[i] - t[0]:8d})")
        prev = t[i]
