"""Diagnostic: phase timeline of posatt_rows_kernel (forward) on the Elasticity processor layer (972-point
per-sample clouds, D = 256, H = 2, batch 10).  Needs -DPIT_STAMPS (see tools/stamp_tiles.py)."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib, ops
torch.manual_seed(0)
xy = torch.rand(10, 972, 2, device="cuda")
plan = ops.MeshPlan("euclid", xy, xy, 1.0, True)
u = torch.randn(10, 972, 256, device="cuda")
lm = torch.rand(2, device="cuda")
concat = (sys.argv[1] if len(sys.argv) > 1 else "concat") == "concat"
print("self-attention with the input copy (torch.cat of pit.py:44)" if concat else "without the input copy")
with torch.no_grad():
    for _ in range(10):
        ops.posatt_apply(u, lm, plan, 2, concat)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L = _lib.lib()
L.pit_debug_read_stamps.argtypes = [ctypes.c_void_p]
assert L.pit_debug_read_stamps(buf) == 0
t = list(buf)
names = {32: "entry", 33: "prologue done (head scale, row/column constants)", 34: "key loop done", 35: "tiles parked in LDS (2 barriers)", 37: "row sums summed over the waves", 38: "epilogue trip 1 of 4", 39: "epilogue trip 2", 40: "epilogue trip 3", 41: "epilogue trip 4",
         36: "reduce-scatter + stores done"}
prev = t[32]
for i in (32, 33, 34, 35, 37, 38, 39, 40, 41, 36):
    print(f"{names[i]:52s} +{t[i] - prev:8d} cycles   (t = {t[i] - t[32]:8d})")
    prev = t[i]
