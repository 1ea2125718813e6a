"""Diagnostic: dense attention forward time against the number of workgroups (batch of per-sample meshes)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import ops
torch.manual_seed(0)
for b in (4, 8, 10, 12, 16, 20):
    xy = torch.rand(b, 972, 2, device="cuda")
    plan = ops.MeshPlan("euclid", xy, xy, 1.0, True)
    u = torch.randn(b, 972, 256, device="cuda")
    lm = torch.rand(2, device="cuda")
    with torch.no_grad():
        for _ in range(10):
            ops.posatt_apply(u, lm, plan, 2, True)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                ops.posatt_apply(u, lm, plan, 2, True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    flops = 2.0 * b * 2 * 972 * 972 * 256
    print(f"batch {b:3d}: {31 * 2 * 2 * b:5d} workgroups  {best:7.1f} us  {flops / best / 1e6:6.1f} TF/s")
