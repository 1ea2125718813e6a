"""Diagnostic (GPU box): the fused processor-block launches in isolation - hipGraph-timed pit_block_weights /
pit_block_fwd / pit_block_bwd at the Darcy shape - and, with a -DPIT_STAMPS library (PIT_LIB_PATH), the phase timeline
of one workgroup of the forward and backward chain kernels.
    python tools/block_bench.py [batch]"""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib, ops, tasks


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


b = int(sys.argv[1]) if len(sys.argv) > 1 and __name__ == "__main__" else 8
model, sample, meta = tasks.make_task("darcy", seed=0)
L_, Lp, H, D, n = _lib.lib(), 256, 2, 64, 4
W, rows = (1 + H) * D, b * 256
plan = model.conv[0]._plan(model.mesh_ltt, model.mesh_ltt, True)
heads = [c.lmda.detach().reshape(-1).contiguous() for c in model.conv]
E = torch.empty(n, H, Lp, Lp, device="cuda"); Q = torch.empty_like(E)
inv = torch.empty(n, H, Lp, device="cuda"); rs = torch.empty(n, H, Lp, 4, device="cuda"); sc = torch.empty(n, H, device="cuda")
hp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in heads])
sp = lambda: torch.cuda.current_stream().cuda_stream


def weights():
    assert L_.pit_block_weights(plan.mesh_in.data_ptr(), Lp, 2, 0, 0.0, n, hp, 0, H, E.data_ptr(), Q.data_ptr(), inv.data_ptr(),
                                rs.data_ptr(), sc.data_ptr(), sp()) == 0


weights()
xc = torch.randn(b, Lp, W, device="cuda"); y = torch.empty(b, Lp, W, device="cuda")
z1 = torch.empty(rows, D, device="cuda"); hh = torch.empty_like(z1); z2 = torch.empty_like(z1)
m = model.mlp[0]
w1, b1, w2, b2 = (t.detach().contiguous() for t in (m.mlp1.weight, m.mlp1.bias, m.mlp2.weight, m.mlp2.bias))


def fwd():
    assert L_.pit_block_fwd(E[0].data_ptr(), inv[0].data_ptr(), Lp, H, D, b, xc.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                            w2.data_ptr(), b2.data_ptr(), 1, z1.data_ptr(), hh.data_ptr(), z2.data_ptr(), y.data_ptr(), W, 0, sp()) == 0


dxc = torch.randn(b, Lp, W, device="cuda"); dxp = torch.empty(b, Lp, W, device="cuda")
scr = torch.empty(rows * 2 * D, device="cuda"); scr_own = torch.randn(rows * 2 * D, device="cuda")
ws = torch.zeros(H * 1024, device="cuda", dtype=torch.float64)
gw1, gb1, gw2, gb2 = (torch.zeros_like(t) for t in (w1, b1, w2, b2))
job = _lib.MlpParamsJob(xc.data_ptr(), W, rows, W, D, D, hh.data_ptr(), 1, scr_own.data_ptr(), D, gw1.data_ptr(), gb1.data_ptr(),
                        gw2.data_ptr(), gb2.data_ptr(), 1, scr_own.data_ptr(), 0)
jp = ctypes.cast(ctypes.pointer(job), ctypes.c_void_p)


def bwd(dscale=True, rider=True):
    assert L_.pit_block_bwd(E[0].data_ptr(), inv[0].data_ptr(), Q[0].data_ptr(), Lp, H, D, b, dxc.data_ptr(), xc.data_ptr(),
                            ws.data_ptr() if dscale else None, w1.data_ptr(), w2.data_ptr(), z1.data_ptr(), z2.data_ptr(), 1, W,
                            dxp.data_ptr(), W, scr.data_ptr(), None, 0, jp if rider else None, None, 0, sp()) == 0


# ---- round 4: the persistent launch (all n blocks) against n block launches
bufs = [torch.randn(b, Lp, W, device="cuda") for _ in range(n)]
wts = [tuple(t.detach().contiguous() for t in (mm.mlp1.weight, mm.mlp1.bias, mm.mlp2.weight, mm.mlp2.bias)) for mm in model.mlp]
z1n = torch.empty(n, rows, D, device="cuda"); hn = torch.empty_like(z1n); z2n = torch.empty_like(z1n)
outn = torch.empty(rows, D, device="cuda")
sync = torch.zeros(ops.LATENT_SYNC_WORDS, device="cuda", dtype=torch.int32)
arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
A_ = [arr(bufs)] + [arr([w[k] for w in wts]) for k in range(4)]


def latent_fwd(flags=0):
    assert L_.pit_latent_fwd(E.data_ptr(), inv.data_ptr(), Lp, H, D, b, n, A_[0], A_[1], A_[2], A_[3], A_[4], z1n.data_ptr(),
                             hn.data_ptr(), z2n.data_ptr(), outn.data_ptr(), D, sync.data_ptr(), flags, 0, sp()) == 0


def blocks_fwd():
    for i in range(n):
        y, ldy = (bufs[i + 1], W) if i + 1 < n else (outn, D)
        assert L_.pit_block_fwd(E[i].data_ptr(), inv[i].data_ptr(), Lp, H, D, b, bufs[i].data_ptr(), wts[i][0].data_ptr(),
                                wts[i][1].data_ptr(), wts[i][2].data_ptr(), wts[i][3].data_ptr(), 1, z1n[i].data_ptr(),
                                hn[i].data_ptr(), z2n[i].data_ptr(), y.data_ptr(), ldy, 0, sp()) == 0


if L_.pit_latent_supported(Lp, H, D, b, n):
    blocks_fwd(); torch.cuda.synchronize(); ref = outn.clone()
    latent_fwd(); torch.cuda.synchronize()
    print(f"latent fwd == block launches: {torch.equal(ref, outn)}; sync[0] = {int(sync[0])}")
    print(f"fast-mode switches after one forward (sync[1]): {int(sync[1])}")
    print(f"batch {b}: {n} block_fwd launches {graph_time(blocks_fwd, reps=5):.2f} us | persistent latent fwd {graph_time(latent_fwd, reps=5):.2f} us "
          f"| spread over XCDs {graph_time(lambda: latent_fwd(1), reps=5):.2f} us")

# ---- the persistent backward (chain + d(scale) of all n blocks, top MLP backward included) against the launches it replaces
dxcs = [torch.empty(b, Lp, W, device="cuda") for _ in range(n)]
scrs = [torch.empty(rows * 2 * D, device="cuda") for _ in range(n)]
wss = [torch.zeros(H * 1024, device="cuda", dtype=torch.float64) for _ in range(n)]
d_out = torch.randn(rows, D, device="cuda"); d_in = torch.empty(rows, D, device="cuda")
B_ = [arr(bufs), arr(dxcs), arr([w[0] for w in wts]), arr([w[2] for w in wts]), arr(scrs), arr(wss)]
gws = [[torch.zeros_like(t) for t in w] for w in wts]
G_ = [arr([g_[k] for g_ in gws]) for k in range(4)]


# a postponed job of the decoder MLP's shape (train_darcy.py: 1849 output points x batch, 128 -> 64 -> 1)
drows = b * 1849
dx_ = torch.randn(drows, 128, device="cuda"); dh_ = torch.randn(drows, 64, device="cuda"); ddy_ = torch.randn(drows, 1, device="cuda")
dscr_ = torch.randn(drows * 65, device="cuda")
dg_ = [torch.zeros(64, 128, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(1, 64, device="cuda"), torch.zeros(1, device="cuda")]
djob = _lib.MlpParamsJob(dx_.data_ptr(), 128, drows, 128, 64, 1, dh_.data_ptr(), 0, ddy_.data_ptr(), 1, dg_[0].data_ptr(), dg_[1].data_ptr(),
                         dg_[2].data_ptr(), dg_[3].data_ptr(), 1, dscr_.data_ptr(), 0)
djp = ctypes.cast(ctypes.pointer(djob), ctypes.c_void_p)


def dec_job_alone():
    assert L_.pit_mlp_bwd_params(dx_.data_ptr(), 128, drows, 128, 64, 1, dh_.data_ptr(), 0, ddy_.data_ptr(), 1, dg_[0].data_ptr(), dg_[1].data_ptr(),
                                 dg_[2].data_ptr(), dg_[3].data_ptr(), 1, dscr_.data_ptr(), 0, sp()) == 0


def latent_bwd(flags=0, riders=True, dec=False):
    assert L_.pit_latent_bwd(E.data_ptr(), inv.data_ptr(), Q.data_ptr(), Lp, H, D, b, n, B_[0], B_[1], B_[2], B_[3], z1n.data_ptr(),
                             z2n.data_ptr(), B_[4], B_[5], hn.data_ptr() if riders else None, G_[0], G_[1], G_[2], G_[3], djp if dec else None,
                             d_out.data_ptr(), D, d_in.data_ptr(), D, sync.data_ptr(), flags, 0, sp()) == 0


def blocks_bwd(riders=False):
    assert L_.pit_mlp_bwd_data(rows, W, D, D, wts[n - 1][0].data_ptr(), wts[n - 1][2].data_ptr(), z1n[n - 1].data_ptr(),
                               z2n[n - 1].data_ptr(), 1, d_out.data_ptr(), D, dxcs[n - 1].data_ptr(), W, scrs[n - 1].data_ptr(), 0, sp()) == 0
    for i in range(n - 1, -1, -1):
        if i > 0:
            prev = (wts[i - 1][0].data_ptr(), wts[i - 1][2].data_ptr(), z1n[i - 1].data_ptr(), z2n[i - 1].data_ptr(), 1, W,
                    dxcs[i - 1].data_ptr(), W, scrs[i - 1].data_ptr(), None, 0)
        else:
            prev = (None, None, None, None, 0, 0, None, 0, None, d_in.data_ptr(), D)
        assert L_.pit_block_bwd(E[i].data_ptr(), inv[i].data_ptr(), Q[i].data_ptr(), Lp, H, D, b, dxcs[i].data_ptr(), bufs[i].data_ptr(),
                                wss[i].data_ptr(), *prev, None, None, 0, sp()) == 0


if L_.pit_latent_supported(Lp, H, D, b, n):
    latent_fwd(); torch.cuda.synchronize()
    blocks_bwd(); torch.cuda.synchronize(); ref_in = d_in.clone(); ref_ws = [w.clone() for w in wss]; [w.zero_() for w in wss]
    latent_bwd(); torch.cuda.synchronize()
    print(f"fast-mode switches so far (sync[1]): {int(sync[1])}")
    print(f"latent bwd d_in == block launches: {torch.equal(ref_in, d_in)}; d(scale) sums rel diff "
          f"{max(float(abs(a.sum() - r.sum()) / (abs(r.sum()) + 1e-300)) for a, r in zip(wss, ref_ws)):.2e}; sync[0] = {int(sync[0])}")
    print(f"batch {b}: top MLP bwd + {n} block_bwd launches (no riders) {graph_time(blocks_bwd, reps=5):.2f} us | persistent latent bwd "
          f"{graph_time(latent_bwd, reps=5):.2f} us (without the weight-gradient tiles {graph_time(lambda: latent_bwd(0, False), reps=5):.2f}) | "
          f"spread over XCDs {graph_time(lambda: latent_bwd(1), reps=5):.2f} us | with the decoder job riding {graph_time(lambda: latent_bwd(0, True, True), reps=5):.2f} us "
          f"(the job as its own launch: {graph_time(dec_job_alone, reps=5):.2f} us)")

if hasattr(L_, "pit_latent_read_stamps") and L_.pit_latent_supported(Lp, H, D, b, n):
    latent_fwd(); torch.cuda.synchronize()
    lb = (ctypes.c_ulonglong * 512)()
    L_.pit_latent_read_stamps.argtypes = [ctypes.c_void_p]
    assert L_.pit_latent_read_stamps(lb) == 0
    t = list(lb)
    pts = ["loop top", "wait done", "contraction issued", "parked", "barrier", "reduced", "barrier", "GEMM1+gelu stored", "barrier",
           "GEMM2+gelu", "barrier", "hand-off stores issued", "posted"]
    for wslot, wname in ((0, "wave 0"), (1, "wave 5")):
        base = t[(wslot * 16 + 0) * 16 + 0]
        print(f"  latent fwd, {wname} of workgroup 5 (cycles since its loop top of block 0):")
        for i in range(n):
            row = [t[(wslot * 16 + i) * 16 + k] - base for k in range(13 if i + 1 < n else 10)]
            print(f"    block {i}: " + " ".join(f"{pts[k].split()[0]}={row[k]}" for k in range(len(row))))

if os.environ.get("BLOCK_DBG"):
    L_.pit_block_set_dbg.argtypes = [ctypes.c_int]
    assert L_.pit_block_set_dbg(int(os.environ["BLOCK_DBG"])) == 0
fwd(); bwd()
print(f"batch {b}: weights {graph_time(weights):.2f} us | block_fwd {graph_time(fwd):.2f} us | block_bwd {graph_time(bwd):.2f} us "
      f"(chain only {graph_time(lambda: bwd(False, False)):.2f}, chain+dscale {graph_time(lambda: bwd(True, False)):.2f}, "
      f"chain+rider {graph_time(lambda: bwd(False, True)):.2f})")
if hasattr(L_, "pit_block_read_stamps"):
    fwd(); bwd(False, False); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    L_.pit_block_read_stamps.argtypes = [ctypes.c_void_p]
    assert L_.pit_block_read_stamps(buf) == 0
    t = list(buf)
    names = {0: "fwd entry", 1: "attention contraction done", 2: "parked + barrier", 3: "reduced, concat tile written",
             4: "barrier", 5: "GEMM1 done", 6: "bias+gelu, Z1/H stored", 7: "barrier", 8: "GEMM2 done", 9: "stored",
             10: "bwd entry", 11: "d(values) contraction done", 12: "parked + barrier", 13: "reduced (+gelu'), dZ2 tile",
             14: "barrier", 15: "phase B (dZ1)", 16: "barrier", 17: "phase C (dX) stored"}
    for lo, hi in ((0, 9), (10, 17)):
        prev = t[lo]
        for i in range(lo, hi + 1):
            print(f"  {names[i]:36s} +{t[i] - prev:7d}   (t = {t[i] - t[lo]:7d})")
            prev = t[i]
