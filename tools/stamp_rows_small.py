"""Diagnostic: phase timeline of posatt_rows_kernel (forward) on the Darcy processor layer at batch 8 (256 latent
points, D = 64, H = 2: 512 value columns).  Needs -DPIT_STAMPS (see tools/stamp_tiles.py)."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib, ops, tasks
model, _, _ = tasks.make_task("darcy", seed=0)
layer, mesh = model.conv[0], model.mesh_ltt
plan = layer._plan(mesh, mesh, True)
u = torch.randn(8, 256, 64, device="cuda")
concat = (sys.argv[1] if len(sys.argv) > 1 else "concat") == "concat"
with torch.no_grad():
    for _ in range(10):
        ops.posatt_apply(u, layer.lmda, plan, 2, concat)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L = _lib.lib()
L.pit_debug_read_stamps.argtypes = [ctypes.c_void_p]
assert L.pit_debug_read_stamps(buf) == 0
t = list(buf)
names = {32: "entry", 33: "prologue done (head scale, row/column constants)", 34: "key loop done", 35: "tiles parked in LDS (2 barriers)",
         37: "row sums summed over the waves", 38: "epilogue trip 1", 36: "reduce-scatter + stores done"}
prev = t[32]
for i in (32, 33, 34, 35, 37, 38, 36):
    if t[i] >= t[32] and t[i] - t[32] < 10**7:
        print(f"{names[i]:52s} +{t[i] - prev:8d} cycles   (t = {t[i] - t[32]:8d})")
        prev = t[i]
