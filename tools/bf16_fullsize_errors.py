"""Diagnostic (GPU box): per-parameter rel-L2 of the bf16 mode (decoder tail stored as bf16 / as fp32) against the fp32
oracle at the full Vorticity and NACA sizes, batch 2.   python tools/bf16_fullsize_errors.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np, torch
import pit_oracle as orc
from position_induced_transformer_amd import ops, tasks, utils


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-300))


for task in ("vorticity", "naca"):
    model, sample, meta = tasks.make_task(task, seed=71)
    mesh_in, func_in, mesh_out, target = sample(2)
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    if task == "vorticity":
        mi = mesh_in.cpu().reshape(-1, 2)
        ref = orc.pit_apply(p, "periodic2d", False, 4, 0.02, 0.02, mi, orc.with_coords(mi, func_in.cpu().reshape(2, -1, 10)),
                            model.mesh_ltt.cpu(), mi, norm_after_enc_proc=True)
    else:
        mo = mesh_out.cpu()
        ltt = mo[:, ::4, ::4, :][:, :56, :13, :].reshape(2, -1, 2)
        ref = orc.pit_apply(p, "euclid", True, 4, 0.02, 0.02, mesh_in.cpu(), func_in.cpu(), ltt, mo.reshape(2, -1, 2))
    orc.rel_lp_loss(target.cpu(), ref.reshape(target.shape), meta["out_dim"], meta["p"]).backward()
    res = {}
    for storage in (True, False):
        ops.BF16_STORAGE = storage
        model.zero_grad(set_to_none=True)
        with ops.math_mode("bf16"), ops.head_scale_route("host"):
            out = model(mesh_in, func_in, mesh_out)
            utils.RelLpNorm(meta["out_dim"], meta["p"])(target, out).backward()
        torch.cuda.synchronize()
        res[storage] = {"out": rel(out.detach().cpu().numpy().reshape(-1), ref.detach().numpy().reshape(-1))}
        for k, q in model.named_parameters():
            res[storage][k] = rel(q.grad.cpu().numpy().reshape(-1), p[k].grad.numpy().reshape(-1))
    ops.BF16_STORAGE = True
    print(task)
    for k in res[True]:
        flag = " <--" if max(res[True][k], res[False][k]) > 0.05 and not k.endswith("lmda") else ""
        print(f"  {k:22s} bf16-stored {res[True][k]:.3e}   fp32-stored {res[False][k]:.3e}{flag}")
