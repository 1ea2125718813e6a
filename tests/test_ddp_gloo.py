"""CPU, world_size 2 over gloo: the batch-axis data-parallel path (ddp.FlatGradients).

The compute of each rank is done by the CPU oracle (the HIP path needs a GPU); what is under
test is the sharding + the single flat all-reduce: the W-rank gradient after the all-reduce
must equal the 1-rank gradient of the concatenated batch (RelLpNorm sums
over the batch, so the reduction is SUM)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_grads(params, x, y):
    import pit_oracle as orc
    mesh, ltt = orc.grid_mesh_2d(12), orc.grid_mesh_2d(5)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    f = orc.with_coords(mesh, x.reshape(x.shape[0], -1, 1))
    out = orc.pit_apply(p, "euclid", False, 2, 0.1, 0.1, mesh, f, ltt, mesh)
    orc.rel_lp_loss(y, out, 1, 2).backward()
    return p


def _worker(rank, world, port, out_dir):
    for pth in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import golden_io as gio
    import pit_oracle as orc
    from position_induced_transformer_amd.ddp import FlatGradients, broadcast_parameters, shard_batch

    shapes = orc.param_shapes(2, 1, 1, 16, 2, 2)
    # every rank starts from different parameters; broadcast makes them rank 0's
    params = {k: torch.nn.Parameter(torch.from_numpy(v)) for k, v in gio.synth_params(shapes, 10 + rank).items()}
    holder = torch.nn.ParameterDict({k.replace(".", "_"): v for k, v in params.items()})
    broadcast_parameters(holder)
    ref0 = {k: torch.from_numpy(v) for k, v in gio.synth_params(shapes, 10).items()}
    for k in params:
        assert torch.equal(params[k].data, ref0[k]), k

    gb = 6                                                        # global batch, uneven split for world=4
    x = torch.from_numpy(gio.synth((gb, 144, 1), 77))
    y = torch.from_numpy(gio.synth((gb, 144, 1), 78))
    sl = shard_batch(gb, rank, world)
    flat = FlatGradients(params.values())
    flat.zero_()
    g = _oracle_grads({k: v.data for k, v in params.items()}, x[sl], y[sl])
    for k, p in params.items():
        p.grad.add_(g[k].grad)                                    # accumulates into the flat buffer views
    assert flat.flat.abs().sum() > 0
    flat.all_reduce()                                             # ONE collective for all parameters
    if rank == 0:
        full = _oracle_grads(ref0, x, y)
        want = torch.cat([full[k].grad.reshape(-1) for k in params])
        got = flat.dense()                                        # (the flat buffer pads every parameter to 64 B)
        err = float((got - want).norm() / want.norm())
        np.save(os.path.join(out_dir, "err.npy"), np.asarray([err, float(got.numel())]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_flat_allreduce_equals_single_rank_gradient(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    err, n = np.load(os.path.join(tmp_path, "err.npy"))
    assert n > 1000
    assert err <= 1e-6, err


def _worker_loop(rank, world, port, out_dir):
    """Three steps of the reference loop shape (train_darcy.py:124-134) under data parallelism:
    `optimizer.zero_grad()` (set_to_none=True: drops the flat-buffer views), backward (autograd then
    allocates fresh gradients), `flat.all_reduce()`, `optimizer.step()`.  Rank 0 replays the same
    three steps single-process on the concatenated batch."""
    for pth in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import golden_io as gio
    import pit_oracle as orc
    from position_induced_transformer_amd.ddp import FlatGradients, shard_batch

    shapes = orc.param_shapes(2, 1, 1, 16, 2, 2)
    mesh, ltt = orc.grid_mesh_2d(12), orc.grid_mesh_2d(5)

    def loss_of(p, x, y):
        f = orc.with_coords(mesh, x.reshape(x.shape[0], -1, 1))
        return orc.rel_lp_loss(y, orc.pit_apply(p, "euclid", False, 2, 0.1, 0.1, mesh, f, ltt, mesh), 1, 2)

    def run(params, shard, reduce):
        flat = FlatGradients(params.values())
        opt = torch.optim.SGD(list(params.values()), lr=1e-2)
        for step in range(3):
            x = torch.from_numpy(gio.synth((4, 144, 1), 200 + step))
            y = torch.from_numpy(gio.synth((4, 144, 1), 300 + step))
            opt.zero_grad()                                       # the script's call: .grad -> None
            loss_of(params, x[shard], y[shard]).backward()
            if reduce:
                flat.all_reduce()                                 # re-attaches, then ONE collective
            else:
                flat.attach()
            opt.step()
        return torch.cat([p.detach().reshape(-1) for p in params.values()])

    mk = lambda: {k: torch.nn.Parameter(torch.from_numpy(v)) for k, v in gio.synth_params(shapes, 10).items()}  # noqa: E731
    got = run(mk(), shard_batch(4, rank, world), True)
    if rank == 0:
        want = run(mk(), slice(0, 4), False)
        np.save(os.path.join(out_dir, "err_loop.npy"), np.asarray([float((got - want).norm() / want.norm())]))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_step_loop_with_optimizer_zero_grad_matches_single_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker_loop, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    (err,) = np.load(os.path.join(tmp_path, "err_loop.npy"))
    assert err <= 1e-6, err


def test_shard_batch_partitions_exactly():
    sys.path.insert(0, ROOT)
    from position_induced_transformer_amd.ddp import shard_batch
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            idx = []
            for r in range(w):
                s = shard_batch(n, r, w)
                idx += list(range(n))[s]
            assert idx == list(range(n))


def _worker_buckets(rank, world, port, out_dir):
    """Two-bucket exchange (engine.TrainStep(all_reduce_buckets=2)): the flat buffer laid out with the early
    bucket as its tail, reduced as "tail" then "head", equals the single all-reduce of the default layout."""
    for pth in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import golden_io as gio
    import pit_oracle as orc
    from position_induced_transformer_amd.ddp import FlatGradients, shard_batch

    shapes = orc.param_shapes(2, 1, 1, 16, 2, 2)
    x = torch.from_numpy(gio.synth((4, 144, 1), 77))
    y = torch.from_numpy(gio.synth((4, 144, 1), 78))
    sl = shard_batch(4, rank, world)
    res = {}
    for mode in ("one", "two"):
        params = {k: torch.nn.Parameter(torch.from_numpy(v)) for k, v in gio.synth_params(shapes, 10).items()}
        tail = [v for k, v in params.items() if k.startswith(("mlp.1.", "de."))] if mode == "two" else ()
        flat = FlatGradients(params.values(), tail=tail)
        if mode == "two":
            assert 0 < flat.tail_start < flat.flat.numel()
            assert [id(q) for q in flat.params[-len(tail):]] == [id(q) for q in tail]
        else:
            assert flat.tail_start == flat.flat.numel()
        flat.zero_()
        g = _oracle_grads({k: v.data for k, v in params.items()}, x[sl], y[sl])
        for k, p in params.items():
            p.grad.add_(g[k].grad)
        if mode == "two":
            before_head = flat.flat[:flat.tail_start].clone()
            flat.all_reduce(part="tail")
            assert torch.equal(flat.flat[:flat.tail_start], before_head)      # the late bucket is untouched
            flat.all_reduce(part="head")
        else:
            flat.all_reduce()
        res[mode] = {k: p.grad.clone() for k, p in params.items()}
    if rank == 0:
        err = max(float((res["two"][k] - res["one"][k]).abs().max()) for k in res["one"])
        np.save(os.path.join(out_dir, "err_buckets.npy"), np.asarray([err]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_bucket_exchange_equals_one_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker_buckets, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    (err,) = np.load(os.path.join(tmp_path, "err_buckets.npy"))
    assert err == 0.0, err


def _worker_agree(rank, world, port, out_dir):
    """engine.TrainStep agrees on the SHAPE of its collective sequence when it is built, and every rank of a data-parallel
    step takes part whatever its own bucket count (a rank that ended with one bucket used to skip the MIN
    all-reduce its peers blocked in)."""
    for pth in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["PIT_IMPORT_SIDE_EFFECTS"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from position_induced_transformer_amd import ops
    from position_induced_transformer_amd import pit as P
    from position_induced_transformer_amd.engine import TrainStep
    # (CPU stand-in for "the kernels write this gradient in place": the real predicate wants a device gradient view)
    ops._grad_slot = lambda p: None if (p is None or p._backward_hooks) else p.grad

    def build(buckets, hook_tail=False):
        ltt = torch.rand(16, 2)
        model = P.pit_fixed(2, 1, 1, 16, 2, 4, ltt, 0.1, 0.1)
        if hook_tail:                                         # its gradient then comes from AccumulateGrad, not in place
            model.de.mlp1.weight.register_hook(lambda g: g)
        x = torch.zeros(2, 36, 1)
        return TrainStep(model, (torch.rand(36, 2), x, torch.rand(36, 2), x.clone()), 1, 2, all_reduce=True, all_reduce_buckets=buckets)

    got = []
    got.append(build(2)._early_decision)                      # every rank: two buckets, all in place -> two buckets
    got.append(build(2 if rank == 0 else 1)._early_decision)  # rank 1 has ONE bucket: both must fall back (and nobody hangs)
    got.append(build(2, hook_tail=(rank == 1))._early_decision)   # rank 1's tail gradient is not written in place
    st = build(1)
    got.append(st._early_decision)
    np.save(os.path.join(out_dir, f"agree_{rank}.npy"), np.asarray(got, dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_bucket_agreement_includes_every_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker_agree, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in (0, 1):
        got = np.load(os.path.join(tmp_path, f"agree_{rank}.npy")).tolist()
        assert got == [1, 0, 0, 0], (rank, got)
