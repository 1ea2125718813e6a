"""Phase timeline of one workgroup of the fused decoder launches (diagnostic library built by `tools/edge_variants.sh 256`:
PIT_LIB_PATH=_diag/libpit_v256.so python tools/edge_stamps.py).  REFCLK stamps, 10 ns per tick."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from position_induced_transformer_amd import _lib, ops, tasks, utils  # noqa: E402

model, sample, meta = tasks.make_task("darcy")
mi, f, mo, y = sample(8)
loss = utils.RelLpNorm(1, 2)
for _ in range(3):
    model.zero_grad()
    loss(y, model(mi, f, mo)).backward()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
fn = _lib.lib().pit_edge_read_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.cast(buf, ctypes.c_void_p)) == 0
names = {0: ["entry", "loads+gather issued", "weights formed", "union parked", "barrier 1", "X tile", "barrier 2", "GEMM1+gelu", "barrier 3", "thin+loss"],
         1: ["entry", "loads+gather issued", "norms", "weights formed", "dZ1", "union parked", "W1 issued + barrier 1", "dX", "barrier 2", "d(values)+atomics", "d(scale)"]}
for k, title in ((0, "decoder_fwd"), (1, "decoder_bwd")):
    st = [buf[k * 16 + i] for i in range(len(names[k]))]
    print(title, "total", (st[-1] - st[0]) * 10, "ns")
    for i in range(1, len(st)):
        print(f"   {names[k][i]:28s} +{(st[i] - st[i - 1]) * 10:6d} ns")
