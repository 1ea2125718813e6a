"""Diagnostic (GPU box): pit_mlp_bwd_params - the two weight-gradient reductions dW1 = dZ1^T X, dW2 = dZ2^T H (+ bias
gradients) of one MLP - alone, at the shapes the task steps run them at.  us per call (hipGraph replay of 20 calls),
achieved TFLOP/s and algorithmic GB/s.  Usage: dw_bench.py [name ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib

MATH = int(os.environ.get("DW_MATH", "0"))      # 0 fp32, 1 bf16 math mode
SHAPES = {   # rows, n0, n1, n2, trailing gelu
    "darcy8": (2048, 192, 64, 64, 1),
    "darcy32": (8192, 192, 64, 64, 1),
    "darcy256": (65536, 192, 64, 64, 1),
    "darcy64": (16384, 192, 64, 64, 1),
    "vort": (5120, 768, 256, 256, 1),
    "naca": (14560, 256, 128, 128, 1),
    "elast": (5120, 384, 128, 128, 1),
    "cyl200": (51200, 768, 256, 256, 1),
}


def graph_time(fn, inner=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
        best = 1e9
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); g.replay(); b.record(s); s.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / inner)
    return best


def main():
    L = _lib.lib()
    names = sys.argv[1:] or list(SHAPES)
    for name in names:
        rows, n0, n1, n2, og = SHAPES[name]
        x, h = torch.randn(rows, n0, device="cuda"), torch.randn(rows, n1, device="cuda")
        scratch = torch.randn(rows * (n1 + n2), device="cuda")
        dy = torch.randn(rows, n2, device="cuda")
        gw1, gb1 = torch.zeros(n1, n0, device="cuda"), torch.zeros(n1, device="cuda")
        gw2, gb2 = torch.zeros(n2, n1, device="cuda"), torch.zeros(n2, device="cuda")

        def call():
            rc = L.pit_mlp_bwd_params(x.data_ptr(), n0, rows, n0, n1, n2, h.data_ptr(), og, dy.data_ptr(), n2, gw1.data_ptr(),
                                      gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), 1, scratch.data_ptr(), MATH,
                                      torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        us = graph_time(call)
        flop = 2.0 * rows * (n0 * n1 + n1 * n2)
        byts = 4.0 * rows * (n0 + 2 * n1 + n2)
        print(f"{name:9s} rows {rows:6d} {n0}->{n1}->{n2}: {us:7.2f} us  {flop / us * 1e-6:6.1f} TFLOP/s  {byts / us * 1e-3:7.1f} GB/s "
              f"(algorithmic)", flush=True)
        # correctness of whatever variant is loaded: against torch (fp64) on the same operands
        gw1.zero_(); gb1.zero_(); gw2.zero_(); gb2.zero_()
        call()
        torch.cuda.synchronize()
        dz1 = scratch[:rows * n1].view(rows, n1).double()
        dz2 = (scratch[rows * n1:].view(rows, n2) if og else dy).double()
        rel = lambda a, b: float((a.double() - b).norm() / b.norm())
        print(f"          rel err dW1 {rel(gw1, dz1.t() @ x.double()):.1e} db1 {rel(gb1, dz1.sum(0)):.1e} "
              f"dW2 {rel(gw2, dz2.t() @ h.double()):.1e} db2 {rel(gb2, dz2.sum(0)):.1e}", flush=True)

if __name__ == "__main__":
    main()
