"""Diagnostic: per-workgroup records of the gemm_rr_kernel launch inside pit_mlp_bwd_params - entry / exit on the 100 MHz
s_memrealtime clock, HW_ID and XCC_ID of every workgroup: how many workgroups each CU got, lifetimes alone / sharing a CU, per XCD.
Needs a library built with -DPIT_STAMPS (PIT_LIB_OUT=... PIT_EXTRA_FLAGS=-DPIT_STAMPS python -m ...build) in PIT_LIB_PATH.
Usage: stamp_dw.py <shape> <which: 1|2>"""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from position_induced_transformer_amd import _lib
# (rows, n0, n1, n2, out_gelu) of the processor MLPs whose weight-gradient reductions the launch performs
SHAPES = {"darcy64": (16384, 192, 64, 64, 1), "darcy256": (65536, 192, 64, 64, 1), "vort": (5120, 768, 256, 256, 1),
          "naca": (14560, 256, 128, 128, 1), "elast": (9720, 768, 256, 256, 1), "cyl200": (179200, 512, 256, 256, 1)}
name, which = sys.argv[1], int(sys.argv[2])
rows, n0, n1, n2, og = SHAPES[name]
L = _lib.lib()
x, h = torch.randn(rows, n0, device="cuda"), torch.randn(rows, n1, device="cuda")
scratch = torch.randn(rows * (n1 + n2), device="cuda")
dy = torch.randn(rows, n2, device="cuda")
gw1, gb1 = torch.zeros(n1, n0, device="cuda"), torch.zeros(n1, device="cuda")
gw2, gb2 = torch.zeros(n2, n1, device="cuda"), torch.zeros(n2, device="cuda")
L.pit_mlp_debug_read_stamps.argtypes = [ctypes.c_void_p]
if which == 1:      # dW1 only: make the second reduction degenerate by calling with n2 tiny? - simply run both; dW1 is launched LAST
    pass
for it in range(5):
    if it == 4:
        torch.cuda.synchronize()
        L.pit_mlp_debug_reset_stamps()
    rc = L.pit_mlp_bwd_params(x.data_ptr(), n0, rows, n0, n1, n2, h.data_ptr(), og, dy.data_ptr(), n2, gw1.data_ptr(), gb1.data_ptr(),
                              gw2.data_ptr(), gb2.data_ptr(), 1, scratch.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
# per-workgroup records of the dW1 launch: lifetime against placement
n = 1024
rec = (ctypes.c_ulonglong * (4 * n))()
L.pit_mlp_debug_read_wgrec.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.pit_mlp_debug_read_wgrec(rec, n) == 0
import collections
rows_ = [(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3], i) for i in range(n) if rec[4 * i + 1] > rec[4 * i] > 0]
t0 = min(r[0] for r in rows_)
def place(hw, xcc):
    return (int(xcc) & 0xf, (int(hw) >> 13) & 0x7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 0xf)      # XCC, SE, SH, CU
percu = collections.defaultdict(list)
for a, b, hw, xcc, i in rows_:
    percu[place(hw, xcc)].append(((a - t0) * 10, (b - t0) * 10, i))
print(f"{len(rows_)} workgroups on {len(percu)} distinct (XCC, SE, SH, CU); workgroups per CU: "
      f"{dict(collections.Counter(len(v) for v in percu.values()))}")
life = sorted((b - a) * 10 for a, b, _, _, _ in rows_)
print(f"lifetime ns: min {life[0]} median {life[len(life) // 2]} p90 {life[int(len(life) * 0.9)]} max {life[-1]}")
byx = collections.defaultdict(list)
for (x, se, sh, cu), v in percu.items():
    for a, b, i in v:
        byx[x].append(b - a)
for x in sorted(byx):
    v = sorted(byx[x])
    print(f"  XCC {x}: {len(v):3d} workgroups, lifetime median {v[len(v) // 2]} max {v[-1]} ns")
alone = [b - a for v in percu.values() if len(v) == 1 for a, b, _ in v]
shared = [b - a for v in percu.values() if len(v) > 1 for a, b, _ in v]
if alone: print(f"  alone on their CU: {len(alone)} workgroups, mean lifetime {sum(alone) / len(alone):.0f} ns")
if shared: print(f"  sharing a CU: {len(shared)} workgroups, mean lifetime {sum(shared) / len(shared):.0f} ns")
