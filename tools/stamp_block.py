"""Diagnostic: per-workgroup records of one block_bwd_kernel launch (the launch of bench.roofline_block_probe: Darcy geometry,
riders included) - entry / exit on the 100 MHz s_memrealtime clock, HW_ID, XCC_ID: lifetimes of chain and rider workgroups, how
many share a CU.  Needs a -DPIT_WGREC library:  tools/variant_build.sh wgrec pit_block.hip -DPIT_WGREC;
PIT_LIB_PATH=_diag/libpit_vwgrec.so python tools/stamp_block.py [batch]"""
import collections, ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from position_induced_transformer_amd import _lib, tasks  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
in_step = len(sys.argv) > 2 and sys.argv[2] == "step"     # the records of the step's LAST block_bwd launch (block 0: no previous MLP)
model, sample, meta = tasks.make_task("darcy", seed=0)
if in_step:
    from position_induced_transformer_amd import utils
    from position_induced_transformer_amd.ddp import FlatGradients
    mesh_in, func_in, mesh_out, target = sample(batch)
    loss_fn = utils.RelLpNorm(meta["out_dim"], meta["p"])
    flat = FlatGradients(model.parameters())
    for _ in range(3):
        flat.zero_()
        loss_fn(target, model(mesh_in, func_in, mesh_out)).backward()
else:
    recs = bench.roofline_block_probe(model, batch, 43 * 43)
    print("block_bwd", recs[0]["us_per_launch"], "us   block_fwd", recs[1]["us_per_launch"], "us (graph replay)")
torch.cuda.synchronize()
n = 1024
rec = (ctypes.c_ulonglong * (4 * n))()
L = _lib.lib()
L.pit_block_read_wgrec.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.pit_block_read_wgrec(rec, n) == 0
rows = [(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3], i) for i in range(n) if rec[4 * i + 1] >= rec[4 * i] > 0]
t0 = min(r[0] for r in rows)
n_chain = 8 * ((batch + 7) // 8) * 16


def place(hw, xcc):
    return (int(xcc) & 0xf, (int(hw) >> 13) & 0x7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 0xf)      # XCC, SE, SH, CU


percu = collections.defaultdict(list)
for a, b, hw, xcc, i in rows:
    kind = "chain" if i < n_chain else ("dscale" if i < 2 * n_chain else "rider")      # (block_bwd's ranges: chain, d(scale) slabs, riders)
    percu[place(hw, xcc)].append((kind, (a - t0) * 10, (b - t0) * 10, i))
print(f"{len(rows)} workgroups on {len(percu)} distinct CUs; per CU: {dict(collections.Counter(len(v) for v in percu.values()))}")
mix = collections.Counter(tuple(sorted(k for k, _, _, _ in v)) for v in percu.values())
print("CU contents:", dict(mix))
for kind in ("chain", "dscale", "rider"):
    ent = sorted(a for v in percu.values() for k, a, b, i in v if k == kind)
    ext = sorted(b for v in percu.values() for k, a, b, i in v if k == kind)
    life = sorted(b - a for v in percu.values() for k, a, b, i in v if k == kind)
    if life:
        print(f"{kind}: {len(life)} workgroups; entry {ent[0]}..{ent[-1]} ns, exit {ext[0]}..{ext[-1]} ns, lifetime min {life[0]} "
              f"median {life[len(life) // 2]} max {life[-1]} ns")
for kind in ("chain", "dscale", "rider"):
    alone = [b - a for v in percu.values() if len(v) == 1 for k, a, b, i in v if k == kind]
    shared = [b - a for v in percu.values() if len(v) > 1 for k, a, b, i in v if k == kind]
    if alone:
        print(f"  {kind} alone on a CU: {len(alone)}, mean lifetime {sum(alone) / len(alone):.0f} ns")
    if shared:
        print(f"  {kind} sharing a CU: {len(shared)}, mean lifetime {sum(shared) / len(shared):.0f} ns")

if hasattr(L, "pit_block_read_stamps"):        # -DPIT_STAMPS as well: shader-clock stamps of wave 0 of workgroup 5 (chain), last launches
    st = (ctypes.c_ulonglong * 32)()
    L.pit_block_read_stamps.argtypes = [ctypes.c_void_p]
    assert L.pit_block_read_stamps(st) == 0
    names = {0: "fwd: own rows requested", 1: "fwd: contraction done", 2: "fwd: parked + barrier", 3: "fwd: reduced", 4: "fwd: barrier",
             5: "fwd: GEMM1", 6: "fwd: gelu + stores", 7: "fwd: barrier", 8: "fwd: GEMM2", 9: "fwd: end",
             10: "bwd: residual requested", 11: "bwd: contraction done", 12: "bwd: parked + barrier", 13: "bwd: reduced", 14: "bwd: barrier",
             15: "bwd: phase B", 16: "bwd: barrier", 17: "bwd: phase C + stores"}
    for base in (0, 10):
        prev = st[base]
        for i in range(base, base + 10):
            if i in names and st[i]:
                print(f"  {names[i]:28s} +{st[i] - prev:6d} ticks of s_memtime  (at {st[i] - st[base]})")
                prev = st[i]

# lifetimes by workgroup-id range (PIT_BLOCK_PRINT=1 prints the ranges of every launch)
ids = sorted((i, (b - a) * 10, (b - t0) * 10) for a, b, hw, xcc, i in rows if i >= 2 * n_chain)
step = 16
for k in range(0, len(ids), step):
    seg = ids[k:k + step]
    print(f"  riders {seg[0][0] - 2 * n_chain:4d}..{seg[-1][0] - 2 * n_chain:4d}: lifetime mean {sum(x[1] for x in seg) / len(seg):6.0f} max {max(x[1] for x in seg):6d} ns, last exit {max(x[2] for x in seg)} ns")
