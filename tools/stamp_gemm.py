"""Diagnostic: per-workgroup placement and lifetime of a gemm_lds_kernel launch inside pit_mlp_fwd (PIT_STAMPS library in
PIT_LIB_PATH).  Usage: stamp_gemm.py rows n0 n1 n2 K_of_the_launch_to_record"""
import collections, ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from position_induced_transformer_amd import _lib
rows, n0, n1, n2, rec_k = [int(a) for a in sys.argv[1:6]]
L = _lib.lib()
x = torch.randn(rows, n0, device="cuda")
w1, b1 = torch.randn(n1, n0, device="cuda") * 0.05, torch.randn(n1, device="cuda")
w2, b2 = torch.randn(n2, n1, device="cuda") * 0.05, torch.randn(n2, device="cuda")
z1, h = torch.empty(rows, n1, device="cuda"), torch.empty(rows, n1, device="cuda")
z2, y = torch.empty(rows, n2, device="cuda"), torch.empty(rows, n2, device="cuda")
assert L.pit_mlp_debug_set_rec_k(rec_k) == 0
for it in range(4):
    rc = L.pit_mlp_fwd(x.data_ptr(), n0, rows, n0, n1, n2, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 1,
                       z1.data_ptr(), h.data_ptr(), z2.data_ptr(), y.data_ptr(), n2, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
n = 4096
rec = (ctypes.c_ulonglong * (4 * n))()
L.pit_mlp_debug_read_wgrec.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.pit_mlp_debug_read_wgrec(rec, n) == 0
rows_ = [(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3], i) for i in range(n) if rec[4 * i + 1] > rec[4 * i] > 0]
t0 = min(r[0] for r in rows_)
place = lambda hw, xcc: (int(xcc) & 0xf, (int(hw) >> 13) & 0x7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 0xf)
percu = collections.defaultdict(list)
for a, b, hw, xcc, i in rows_:
    percu[place(hw, xcc)].append(((a - t0) * 10, (b - t0) * 10, i))
print(f"{len(rows_)} workgroups on {len(percu)} distinct CUs; workgroups per CU: {dict(sorted(collections.Counter(len(v) for v in percu.values()).items()))}")
life = sorted((b - a) * 10 for a, b, _, _, _ in rows_)
ent = sorted((a - t0) * 10 for a, b, _, _, _ in rows_)
print(f"lifetime ns: min {life[0]} median {life[len(life) // 2]} p90 {life[int(len(life) * 0.9)]} max {life[-1]}")
print(f"entry ns after the first: median {ent[len(ent) // 2]} p90 {ent[int(len(ent) * 0.9)]} max {ent[-1]}; last exit {max((b - t0) * 10 for a, b, _, _, _ in rows_)} ns")
