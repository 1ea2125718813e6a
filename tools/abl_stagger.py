"""Diagnostic: effect of staggering co-resident workgroups (PIT_EXP_STAGGER) and of one workgroup per CU
(PIT_EXP_SMEM) on the dense attention forward of the Elasticity processor layer."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
if len(sys.argv) < 2:
    for env in ({}, {"PIT_EXP_SMEM": "90000"}, {"PIT_EXP_STAGGER": "2"}, {"PIT_EXP_STAGGER": "4"}, {"PIT_EXP_STAGGER": "6"},
                {"PIT_EXP_STAGGER": "8"}, {"PIT_EXP_STAGGER": "4", "PIT_EXP_STAGGER_LO": "1", "PIT_EXP_STAGGER_HI": "2"}, {}):
        e = dict(os.environ); e.update(env)
        print(env, end="  ", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "run"], check=False, env=e)
    sys.exit(0)
import torch
from position_induced_transformer_amd import ops
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
print(f"{best:7.1f} us per forward launch")
