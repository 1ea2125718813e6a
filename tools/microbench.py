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
    # attention backward parts through the raw ABI (processor layer): d(scale) only, d(values) only, both
    from position_induced_transformer_amd import _lib
    L = _lib.lib()
    u = torch.randn(b, 256, 64, device="cuda")
    rowstat = torch.empty(1, 2, 256, 4, device="cuda"); scale = torch.empty(2, device="cuda")
    out = torch.empty(b, 256, 192, device="cuda")
    lm = layer.lmda.detach().reshape(-1).contiguous()
    st = _lib.stream_ptr
    L.pit_posatt_fwd(ltt.data_ptr(), ltt.data_ptr(), 1, 256, 256, 2, 0, 0.0, u.data_ptr(), b, 64, 64, 256 * 64,
                     lm.data_ptr(), 2, 0, _lib.ptr(plan.stats), plan.rank_w, 0, 1, out.data_ptr(), 192, 256 * 192, 64, 1,
                     rowstat.data_ptr(), scale.data_ptr(), None, None, 0, 0, 0, st())
    d_out = torch.randn(b, 256, 192, device="cuda"); d_u = torch.empty_like(u); d_lm = torch.zeros(2, device="cuda")
    work = ops._dscale_workspace(u.device, 2)
    def bwd(dv, dh):
        rc = L.pit_posatt_bwd(ltt.data_ptr(), ltt.data_ptr(), 1, 256, 256, 2, 0, 0.0, u.data_ptr(), b, 64, 64, 256 * 64,
                              lm.data_ptr(), 2, 0, scale.data_ptr(), rowstat.data_ptr(), 0,
                              d_out.data_ptr(), 192, 256 * 192, 64, _lib.ptr(dv), 64, 256 * 64, 1,
                              _lib.ptr(dh), 0, work.data_ptr(), None, None, 0, 0, None, None, 0, 0, st())
        assert rc == 0, rc
    res["proc_bwd_dscale_only(+finish)"] = graph_time(lambda: bwd(None, d_lm))
    res["proc_bwd_dvalues_only"] = graph_time(lambda: bwd(d_u, None))
    res["proc_bwd_both"] = graph_time(lambda: bwd(d_u, d_lm))
    for k, v in res.items():
        if flt in k:
            print(f"{k:32s} {v:8.2f} us")

if __name__ == "__main__":
    main()
