"""Diagnostic (GPU box): pit_mlp_fwd and pit_mlp_bwd_data alone at the large-row shapes (us per call, hipGraph of 20 calls).
Usage: mlp_bench.py [rows n0 n1 n2]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib
from dw_bench import graph_time

shapes = [(65536, 192, 64, 64), (16384, 192, 64, 64), (14560, 256, 128, 128)]
if len(sys.argv) == 5:
    shapes = [tuple(int(a) for a in sys.argv[1:5])]
L = _lib.lib()
for rows, n0, n1, n2 in shapes:
    x = torch.randn(rows, n0, device="cuda")
    w1, b1 = torch.randn(n1, n0, device="cuda") * 0.05, torch.randn(n1, device="cuda")
    w2, b2 = torch.randn(n2, n1, device="cuda") * 0.05, torch.randn(n2, device="cuda")
    z1, h = torch.empty(rows, n1, device="cuda"), torch.empty(rows, n1, device="cuda")
    z2, y = torch.empty(rows, n2, device="cuda"), torch.empty(rows, n2, device="cuda")
    dy, dx = torch.randn(rows, n2, device="cuda"), torch.empty(rows, n0, device="cuda")
    scratch = torch.empty(rows * (n1 + n2), device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream

    def fwd():
        assert L.pit_mlp_fwd(x.data_ptr(), n0, rows, n0, n1, n2, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 1,
                             z1.data_ptr(), h.data_ptr(), z2.data_ptr(), y.data_ptr(), n2, 0, st()) == 0

    def bwd():
        assert L.pit_mlp_bwd_data(rows, n0, n1, n2, w1.data_ptr(), w2.data_ptr(), z1.data_ptr(), z2.data_ptr(), 1, dy.data_ptr(), n2,
                                  dx.data_ptr(), n0, scratch.data_ptr(), 0, st()) == 0
    fwd()
    tf, tb = graph_time(fwd), graph_time(bwd)
    # reference check (fp64)
    xd = x.double()
    z1r = xd @ w1.double().t() + b1.double()
    hr = torch.nn.functional.gelu(z1r)
    z2r = hr @ w2.double().t() + b2.double()
    yr = torch.nn.functional.gelu(z2r)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    fwd(); bwd(); torch.cuda.synchronize()
    g = lambda z: 0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * 3.141592653589793) ** 0.5
    dz2 = dy.double() * g(z2r)
    dz1 = (dz2 @ w2.double()) * g(z1r)
    dxr = dz1 @ w1.double()
    print(f"rows {rows} {n0}->{n1}->{n2}: fwd {tf:7.2f} us, bwd data {tb:7.2f} us | rel err y {rel(y, yr):.1e} z1 {rel(z1, z1r):.1e} "
          f"dx {rel(dx, dxr):.1e} dz1 {rel(scratch[:rows * n1].view(rows, n1), dz1):.1e}", flush=True)
