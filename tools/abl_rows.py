"""Diagnostic: ablation timing of the dense attention forward (posatt_rows_kernel) on the Elasticity processor
layer (972-point per-sample clouds, D = 256, H = 2, batch 10).  Variant libraries libpit_hip_abl{1,2,3}.so are
pit_posatt.hip compiled with -DPIT_ABL=n (1: no weight formation, 2: no value loads, 3: neither); results of the
variants are meaningless, only their timing is read.  usage: abl_rows.py [variant]   (no argument: all, in child processes)"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
if len(sys.argv) < 2:
    for v in ("0", "1", "2", "3", "0"):
        subprocess.run([sys.executable, os.path.abspath(__file__), v], check=False)
    sys.exit(0)
import torch
from position_induced_transformer_amd import _lib, ops
v = sys.argv[1]
if v != "0":
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libpit_hip_abl{v}.so")
torch.manual_seed(0)
xy = torch.rand(10, 972, 2, device="cuda")
plan = ops.MeshPlan("euclid", xy, xy, 1.0, True)
u = torch.randn(10, 972, 256, device="cuda")
lm = torch.rand(2, device="cuda")
with torch.no_grad():
    for _ in range(20):
        ops.posatt_apply(u, lm, plan, 2, True)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.posatt_apply(u, lm, plan, 2, True)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
flops = 2.0 * 10 * 2 * 972 * 972 * 256
print(f"variant {v}: {best:7.1f} us per forward launch  ({flops / best / 1e6:6.1f} TF/s)")
