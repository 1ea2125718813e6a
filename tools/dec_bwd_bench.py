"""Diagnostic (GPU box): the Darcy decoder (up-projection 1849 <- 256, candidate lists) backward launch in parts -
d(values) alone, d(scale) alone, both, both + the decoder MLP's weight-gradient rider - and the decoder forward pieces."""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from position_induced_transformer_amd import _lib, ops, tasks
sys.path.insert(0, os.path.join(ROOT, "tools"))
from block_bench import graph_time  # noqa: E402  (prints its own line first)

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model, sample, meta = tasks.make_task("darcy", seed=0)
mesh_in, func_in, mesh_out, target = sample(b)
mesh = mesh_in.reshape(-1, 2)
plan = model.up._plan(mesh, model.mesh_ltt, False)
L = _lib.lib()
H, D, N, J = 2, 64, 1849, 256
u = torch.randn(b, J, D, device="cuda", requires_grad=True)
lm = model.up.lmda.detach().clone().requires_grad_(True)
out = ops.posatt_apply(u, lm, plan, H, False)
values, head, rowstat, scale = out.grad_fn.saved_tensors
d_out = torch.randn_like(out)
d_values = torch.empty_like(u)
d_head = torch.zeros(H, device="cuda")
work = torch.zeros(H * 1024, device="cuda", dtype=torch.float64)
de = model.de
rows = b * N
x2, hh = torch.randn(rows, 128, device="cuda"), torch.randn(rows, 64, device="cuda")
scratch = torch.randn(rows * 65, device="cuda")
dy = torch.randn(rows, 1, device="cuda")
gw1, gb1 = torch.zeros(64, 128, device="cuda"), torch.zeros(64, device="cuda")
gw2, gb2 = torch.zeros(1, 64, device="cuda"), torch.zeros(1, device="cuda")
job = _lib.MlpParamsJob(x2.data_ptr(), 128, rows, 128, 64, 1, hh.data_ptr(), 0, dy.data_ptr(), 1, gw1.data_ptr(), gb1.data_ptr(),
                        gw2.data_ptr(), gb2.data_ptr(), 1, scratch.data_ptr(), 0)
jp = ctypes.cast(ctypes.pointer(job), ctypes.c_void_p)


def bwd(dv, dh, rider):
    rc = L.pit_posatt_bwd(plan.mesh_out.data_ptr(), plan.mesh_in.data_ptr(), 1, N, J, 2, 0, 0.0,
                          values.data_ptr(), b, D, values.stride(1), values.stride(0), head.data_ptr(), H, 0, scale.data_ptr(),
                          rowstat.data_ptr(), 1, d_out.data_ptr(), d_out.stride(1), d_out.stride(0), 0,
                          d_values.data_ptr() if dv else None, d_values.stride(1), d_values.stride(0), 0,
                          d_head.data_ptr() if dh else None, 1 | 2, work.data_ptr(),
                          plan.nbr_idx.data_ptr(), plan.nbr_cnt.data_ptr(), plan.nbr_cap, plan.lists_complete(),
                          plan.rev_ptr.data_ptr(), plan.rev_row.data_ptr(), jp if rider else None, 0, 0,
                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


print(f"decoder bwd batch {b}: d(values) only {graph_time(lambda: bwd(True, False, False)):.2f} us | d(scale) only "
      f"{graph_time(lambda: bwd(False, True, False)):.2f} | both {graph_time(lambda: bwd(True, True, False)):.2f} | both + rider "
      f"{graph_time(lambda: bwd(True, True, True)):.2f}")
work.zero_()
with torch.no_grad():
    print(f"decoder fwd attention {graph_time(lambda: ops.posatt_apply(u, lm, plan, H, False)):.2f} us | de MLP fwd "
          f"{graph_time(lambda: de(out)):.2f} us | de rider alone (pit_mlp_bwd_params) "
          f"{graph_time(lambda: L.pit_mlp_bwd_params(x2.data_ptr(), 128, rows, 128, 64, 1, hh.data_ptr(), 0, dy.data_ptr(), 1, gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), 1, scratch.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)):.2f} us")
