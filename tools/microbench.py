#!/usr/bin/env python3
"""Kernel microbenchmarks on the GPU box: each op is captured N times into a hipGraph and the
replay is timed with HIP events (removes host launch overhead).  Usage:
    python tools/microbench.py [name-filter]
Env overrides understood by the library: PIT_FORCE_CT, PIT_FORCE_WAVES."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from position_induced_transformer_amd import ops, tasks

def graph_time(fn, reps=20, replays=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)

def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    b = int(os.environ.get("MB_BATCH", "8"))
    model, sample, meta = tasks.make_task("darcy", seed=0)
    mesh_in, func_in, mesh_out, target = sample(b)
    mesh = mesh_in.reshape(-1, 2)
    ltt = model.mesh_ltt
    res = {}
    with torch.no_grad():
        # processor attention fwd (256x256, D=64, H=2, concat)
        layer = model.conv[0]
        plan = layer._plan(ltt, ltt, True)
        u = torch.randn(b, 256, 64, device="cuda")
        c = ops.head_scale(layer.lmda).reshape(-1)
        res["proc_fwd(lmda)"] = graph_time(lambda: ops.posatt_apply(u, layer.lmda, plan, 2, True))
        res["proc_fwd(scale)"] = graph_time(lambda: ops.posatt_apply(u, c, plan, 2, True, True))
        # encoder fwd
        pe = model.down._plan(ltt, mesh, False)
        ue = torch.randn(b, 1849, 3, device="cuda")
        res["enc_fwd"] = graph_time(lambda: ops.posatt_apply(ue, model.down.lmda, pe, 2, False))
        pd = model.up._plan(mesh, ltt, False)
        res["dec_fwd"] = graph_time(lambda: ops.posatt_apply(u, model.up.lmda, pd, 2, False))
        # mlp fwd
        m = model.mlp[0]
        x = torch.randn(b, 256, 192, device="cuda")
        res["mlp_fwd_192_64_64"] = graph_time(lambda: m(x, out_gelu=True))
        res["empty_kernel(head_scale)"] = graph_time(lambda: ops.head_scale(layer.lmda))
    # backward pieces via autograd (captured too)
    def fb(make_in, fn):
        x = make_in().requires_grad_(True)
        def run():
            y = fn(x)
            y.backward(torch.ones_like(y))
            x.grad = None
        return run
    layer = model.conv[0]; plan = layer._plan(ltt, ltt, True)
    res["proc_fwd+bwd"] = graph_time(fb(lambda: torch.randn(b, 256, 64, device="cuda"),
                                        lambda x: ops.posatt_apply(x, layer.lmda, plan, 2, True)))
    pd = model.up._plan(mesh, ltt, False)
    res["dec_fwd+bwd"] = graph_time(fb(lambda: torch.randn(b, 256, 64, device="cuda"),
                                       lambda x: ops.posatt_apply(x, model.up.lmda, pd, 2, False)))
    m = model.mlp[0]
    res["mlp_fwd+bwd_192_64_64"] = graph_time(fb(lambda: torch.randn(b, 256, 192, device="cuda"),
                                                 lambda x: m(x, out_gelu=True)))
    md = model.de
    res["mlp_fwd+bwd_128_64_1(dec)"] = graph_time(fb(lambda: torch.randn(b, 1849, 128, device="cuda"),
                                                     lambda x: md(x)))
    for k, v in res.items():
        if flt in k:
            print(f"{k:32s} {v:8.2f} us")

if __name__ == "__main__":
    main()
