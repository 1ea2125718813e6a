"""Zero-shot super-resolution forward of train_darcy.py:152-178: the 43x43-trained Darcy model evaluated on the 421x421
grid (J = 177 241 input points), batch 1, no_grad; prints the forward time of repeated calls (plans cached after the first)."""
import os, sys, time
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import tasks
model, sample, meta = tasks.make_task("darcy", seed=0)
s = 421
mesh = tasks.grid_mesh_2d(s, True, torch.device("cuda"))
x = torch.randn(1, s, s, 1, device="cuda")
with torch.no_grad():
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = model(mesh, x, mesh)
        torch.cuda.synchronize(); print(f"zssr 421x421 forward {i}: {(time.perf_counter()-t0)*1e3:.1f} ms", tuple(y.shape), float(y.abs().mean()))
print("finite:", bool(torch.isfinite(y).all()))
