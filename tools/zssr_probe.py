import sys, time
sys.path[:0] = ["/root/repo"]
import torch
from position_induced_transformer_amd import tasks
model, sample, meta = tasks.make_task("darcy", seed=0)
s = 421
mesh = tasks.grid_mesh_2d(s, True, torch.device("cuda"))
x = torch.randn(1, s, s, 1, device="cuda")
with torch.no_grad():
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = model(mesh, x, mesh)
        torch.cuda.synchronize(); print(f"zssr 421x421 forward {i}: {(time.perf_counter()-t0)*1e3:.1f} ms", tuple(y.shape), float(y.abs().mean()))
print("finite:", bool(torch.isfinite(y).all()))
