#!/usr/bin/env python3
"""Soak test (GPU box): N hipGraph replays of the full training step (fwd + loss + bwd + fused Adam),
then checks of everything that is supposed to be self-cleaning or exactly counted:
the Adam step counter, the arrival tickets, the fp64 d(scale) accumulators, the loss workspace."""
import sys, time
sys.path[:0] = ["."]
import torch
from position_induced_transformer_amd import ops, tasks
from position_induced_transformer_amd.ddp import FlatAdam, FlatGradients
from position_induced_transformer_amd.engine import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
task = sys.argv[2] if len(sys.argv) > 2 else "darcy"
model, sample, meta = tasks.make_task(task, seed=0)
batch = sample(meta["batch"])
flat = FlatGradients(model.parameters(), flatten_params=True)
opt = FlatAdam(flat, lr=1e-4, cosine_t_max=n, zero_grads=True)
step = TrainStep(model, batch, meta["out_dim"], meta["p"], optimizer=opt, flat=flat)
step.capture()
torch.cuda.synchronize()
base = int(opt.step_count)
t0 = time.perf_counter()
losses = []
for i in range(n):
    step.replay()
    if i % (n // 10) == 0:
        losses.append(float(step.loss))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{task}: {n} replays in {dt:.2f} s ({dt / n * 1e3:.4f} ms/step), loss {losses[0]:.5f} -> {float(step.loss):.5f}")
assert all(torch.isfinite(torch.tensor(losses))) and torch.isfinite(step.loss)
assert int(opt.step_count) == base + n, (int(opt.step_count), base + n)
assert float(opt.scalars[3].view(torch.int32)) == 0, "Adam arrival ticket not reset"
assert float(flat.flat.abs().max()) == 0.0, "gradients not cleared by the fused Adam"
for ws in ops._LAYER_WS.values():
    assert float(ws.abs().max()) == 0.0, "d(scale) accumulators not drained"
for ws in ops._LOSS_WS.values():
    assert float(ws.abs().max()) == 0.0, "loss workspace not reset"
assert torch.isfinite(flat.flat_params).all()
print("soak ok")
