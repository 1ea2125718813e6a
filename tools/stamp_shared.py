"""Diagnostic (needs a -DPIT_STAMPS build, PIT_LIB_PATH): steady-state loop of rows_shared_body on the Elasticity
processor layer - cycles per chunk and the share spent waiting at the per-chunk barrier, per wave of one workgroup."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib, ops
torch.manual_seed(0)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 10
xy = torch.rand(b, 972, 2, device="cuda")
plan = ops.MeshPlan("euclid", xy, xy, 1.0, True)
u = torch.randn(b, 972, 256, device="cuda")
lm = torch.rand(2, device="cuda")
with torch.no_grad():
    for _ in range(10):
        ops.posatt_apply(u, lm, plan, 2, True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L = _lib.lib()
L.pit_debug_read_stamps.argtypes = [ctypes.c_void_p]
assert L.pit_debug_read_stamps(buf) == 0
for w in range(4):
    tot, bar, n = buf[48 + 3 * w], buf[49 + 3 * w], buf[50 + 3 * w]
    print(f"batch {b} wave {w}: {n} chunks, {tot} cycles = {tot / max(n, 1):.0f} per chunk (32 MFMAs = 2048 pipe cycles), barrier wait {bar} = {100.0 * bar / max(tot, 1):.1f} %")
