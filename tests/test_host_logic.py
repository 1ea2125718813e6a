"""CPU: host logic around the HIP path that needs no device - the torch.compile boundary, the
import-time behaviour of pit.py:1-11, exception safety of the deferred d(lmda) finish, the flat
gradient buffer surviving ``optimizer.zero_grad()``, the per-thread math mode / head-scale route."""
import os
import subprocess
import sys
import threading

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------- torch.compile boundary
def test_task_forward_is_opaque_to_dynamo_and_wrapper_keeps_orig_mod_keys():
    """train_darcy.py:112 `model = torch.compile(model)` and :150 `model.state_dict()`: the wrapper
    must carry the `_orig_mod.` keys and calling it must reach OUR operators (which refuse CPU
    tensors with their own message), not fail inside dynamo."""
    from position_induced_transformer_amd import pit as P, tasks
    model, _, _ = tasks.make_task("darcy", device="cpu")
    assert getattr(type(model).forward, "_pit_eager", False)
    compiled = torch.compile(model)
    keys = list(compiled.state_dict().keys())
    assert keys and all(k.startswith("_orig_mod.") for k in keys)
    assert [k[len("_orig_mod."):] for k in keys] == list(model.state_dict().keys())
    with pytest.raises(RuntimeError, match="HIP device only"):
        compiled(torch.zeros(43, 43, 2), torch.zeros(2, 43, 43, 1), torch.zeros(43, 43, 2))
    # torch._dynamo.disable(model) of train_darcy.py:152 is accepted on the compiled wrapper as well
    again = torch._dynamo.disable(compiled)
    with pytest.raises(RuntimeError, match="HIP device only"):
        again(torch.zeros(43, 43, 2), torch.zeros(2, 43, 43, 1), torch.zeros(43, 43, 2))

    class script_model(P.pit_fixed):              # a class written the way the scripts write theirs
        def forward(self, mesh_in, func_in, mesh_out):
            return self.encoder(mesh_in, func_in, self.mesh_ltt)
    assert getattr(script_model.forward, "_pit_eager", False)


def test_operator_entry_points_are_dynamo_disabled():
    from position_induced_transformer_amd import ops
    for fn in (ops.posatt_apply, ops.mlp_apply, ops.rel_lp_loss, ops.instance_norm_points, ops.rel_max_norm):
        assert getattr(fn, "_torchdynamo_disable", False), fn


def test_reference_checkpoint_through_compile_roundtrip(tmp_path):
    """A checkpoint written as the scripts write it (state_dict of the torch.compile wrapper,
    train_darcy.py:150) loads back into a fresh model."""
    from position_induced_transformer_amd import tasks, utils
    model, _, _ = tasks.make_task("darcy", device="cpu", seed=3)
    path = tmp_path / "model.pth"
    torch.save({"model_state": torch.compile(model).state_dict()}, path)
    fresh, _, _ = tasks.make_task("darcy", device="cpu", seed=4)
    utils.load_reference_checkpoint(fresh, str(path))
    for (k, a), (_, b) in zip(model.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k


# --------------------------------------------------------------------------- import-time behaviour
def _import_probe(env_extra):
    code = ("import torch, numpy as np; torch.manual_seed(1234); np.random.seed(99); "
            "import position_induced_transformer_amd.pit as P; "
            "print(torch.initial_seed(), int(np.random.get_state()[1][0]), torch.get_float32_matmul_precision())")
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    return int(out[0]), int(out[1]), out[2]


def test_import_reproduces_the_reference_side_effects_with_opt_out():
    """pit.py:2-10 seeds torch / numpy with 0 and sets the matmul precision at import; scripts that
    never seed rely on it.  Reproduced by default, switched off by PIT_IMPORT_SIDE_EFFECTS=0."""
    seed, np0, prec = _import_probe({})
    assert seed == 0 and prec == "high"
    import numpy as np
    np.random.seed(0)
    assert np0 == int(np.random.get_state()[1][0])
    seed, _, _ = _import_probe({"PIT_IMPORT_SIDE_EFFECTS": "0"})
    assert seed == 1234


# --------------------------------------------------------------------------- deferred finish, exception safety
def test_deferred_head_finish_survives_a_backward_pass_that_raised(monkeypatch):
    """autograd skips end-of-pass callbacks when a backward raises.  The next pass must
    still queue its flush, must not drain the aborted pass's entries, and must have zeroed the
    accumulators those entries point at."""
    from position_induced_transformer_amd import ops
    flushed = []

    def fake_flush(task):
        flushed.append([p[0] for p in ops._PENDING_HEADS.pop(task)])
    monkeypatch.setattr(ops, "_flush_head_finishes", fake_flush)
    ops._PENDING_HEADS.clear()

    class Layer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, ws, boom):
            ctx.ws, ctx.boom = ws, boom
            return x * 2

        @staticmethod
        def backward(ctx, g):
            ctx.ws += 1.0                                 # "partial sums" of this pass
            ops._defer_head_finish(ctx.ws, None, None, None, 1, 0)
            if ctx.boom:
                raise RuntimeError("boom")
            return g * 2, None, None

    x = torch.ones(3, requires_grad=True)
    ws_a, ws_b = torch.zeros(4, dtype=torch.float64), torch.zeros(4, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="boom"):
        Layer.apply(Layer.apply(x, ws_a, False), ws_a, True).sum().backward()
    assert flushed == [] and len(ops._PENDING_HEADS) >= 1         # the callback never ran: entries are stale
    Layer.apply(x, ws_b, False).sum().backward()                    # a healthy pass on OTHER accumulators (another model)
    assert len(flushed) == 1 and len(flushed[0]) == 1 and flushed[0][0] is ws_b
    assert float(ws_a.abs().sum()) > 0.0 and len(ops._PENDING_HEADS) == 1   # ... does not touch the aborted pass's entries
    Layer.apply(x, ws_a, False).sum().backward()                    # the next pass over the SAME accumulators does:
    assert len(flushed) == 2 and len(flushed[1]) == 1 and flushed[1][0] is ws_a
    assert float(ws_a.abs().sum()) == 0.0                            # the aborted pass's partial sums were cleared
    assert ops._PENDING_HEADS == {}
    Layer.apply(x, ws_b, False).sum().backward()                    # and the pass after that is normal again
    assert len(flushed) == 3


# --------------------------------------------------------------------------- flat gradients vs zero_grad()
def test_flat_gradients_reattach_after_optimizer_zero_grad():
    """`optimizer.zero_grad()` (set_to_none=True) drops the views into the flat buffer;
    all_reduce()/attach() must copy the fresh gradients back and re-point .grad, every step."""
    from position_induced_transformer_amd.ddp import FlatGradients
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    flat = FlatGradients(lin.parameters())
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    for step in range(3):
        opt.zero_grad()                                   # .grad -> None
        x = torch.randn(4, 5)
        lin(x).pow(2).sum().backward()                    # autograd allocates fresh grads outside `flat`
        expect = torch.cat([p.grad.reshape(-1).clone() for p in lin.parameters()])
        assert flat.attach() == 2
        assert torch.equal(flat.dense(), expect), step
        for p, v in zip(flat.params, flat._views):
            assert p.grad.data_ptr() == v.data_ptr()
        assert flat.attach() == 0
        opt.step()
    flat.zero_()
    assert float(flat.flat.abs().sum()) == 0.0 and all(p.grad is not None for p in lin.parameters())


# --------------------------------------------------------------------------- per-thread modes
def test_math_mode_and_head_scale_route_are_per_thread_host_state():
    from position_induced_transformer_amd import ops
    assert ops.get_math_mode() == "fp32" and ops.get_head_scale_route() == "device"
    seen = {}

    def worker():
        seen["before"] = (ops.get_math_mode(), ops.get_head_scale_route())
        ops.set_math_mode("bf16")
        ops.set_head_scale_route("host")
        seen["after"] = (ops.get_math_mode(), ops.get_head_scale_route())
    with ops.math_mode("bf16"), ops.head_scale_route("host"):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert (ops.get_math_mode(), ops.get_head_scale_route()) == ("bf16", "host")
    assert seen["before"] == ("fp32", "device") and seen["after"] == ("bf16", "host")
    assert (ops.get_math_mode(), ops.get_head_scale_route()) == ("fp32", "device")
    with pytest.raises(ValueError):
        ops.set_head_scale_route("gpu")


def test_inplace_gradient_slots_are_opt_in():
    """kernels write into .grad in place only for parameters FlatGradients registered, while
    the registered view is still attached and no hooks sit on the parameter."""
    from position_induced_transformer_amd import ops
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.zeros(4)
    assert ops._grad_slot(p) is None                     # an existing .grad alone is not consent
    ops.mark_inplace_grad(p, p.grad)
    # (CPU tensors never qualify: the check is for device fp32 buffers)
    assert ops._grad_slot(p) is None


# --------------------------------------------------------------------------- round 3
def test_importing_ops_or_utils_does_not_fire_the_drop_in_side_effects():
    """the reference's import-time side effects (pit.py:1-11) belong to the drop-in module `pit` only."""
    import subprocess
    code = ("import sys, torch; torch.manual_seed(123); a = torch.rand(1).item();"
            "torch.manual_seed(123); import position_induced_transformer_amd.ops, position_induced_transformer_amd.utils;"
            "assert 'position_induced_transformer_amd.pit' not in sys.modules;"
            "assert torch.rand(1).item() == a;"                    # the global RNG was not reseeded
            "import position_induced_transformer_amd as P; P.pit;"  # lazy attribute: loads (and seeds, as pit.py:3)
            "assert 'position_induced_transformer_amd.pit' in sys.modules;"
            "torch.manual_seed(0); b = torch.rand(1).item(); import importlib;"
            "importlib.reload(sys.modules['position_induced_transformer_amd.pit']); assert torch.rand(1).item() == b")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]


def test_subclass_overrides_of_dist2att_or_convolution_are_detected():
    """pit.py:42-43 calls self.dist2att / self.convolution: a subclass that overrides either must not be bypassed."""
    from position_induced_transformer_amd import pit as P

    class mine(P.posatt_fixed):
        def convolution(self, A, U):
            return 2.0 * super().convolution(A, U)

    class other(P.posatt_cross):
        def dist2att(self, mesh_out, mesh_in, scale, locality):
            return super().dist2att(mesh_out, mesh_in, scale, 1.0)

    for cls in (P.posatt, P.posatt_cross, P.posatt_fixed, P.posatt_cross_fixed, P.posatt_periodic1d,
                P.posatt_cross_periodic1d, P.posatt_periodic2d, P.posatt_cross_periodic2d):
        assert not cls(2, 4, 0.5)._overridden(), cls
    assert mine(2, 4, 0.5)._overridden() and other(2, 4, 0.5)._overridden()
    a, u = torch.rand(2, 5, 7, dtype=torch.float32), torch.rand(3, 7, 4)
    want = torch.einsum("hnj,bjd->bnhd", a.double(), u.double()).reshape(3, 5, 8).float()
    assert torch.equal(P.posatt_fixed(2, 4, 1.0).convolution(a, u), want)          # exact fp64 contraction, rounded once
    assert torch.get_float32_matmul_precision() == "high"                           # (pit.py:2; the helper does not toggle it)


def test_build_is_stale_after_a_build_with_other_flags_and_the_binding_checks_the_abi_version(tmp_path, monkeypatch):
    from position_induced_transformer_amd import _lib, build
    assert not build._stale()
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DPIT_STAMPS"])
    assert build._stale()                       # a diagnostic build would not be mistaken for this one, nor vice versa
    monkeypatch.undo()
    assert _lib.lib().pit_version() == _lib.ABI_VERSION
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="PIT_ABI_VERSION"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib() is not None


def test_postponed_job_slices_cover_the_rows_once_and_respect_the_early_bucket():
    """ops._dw_slices: a postponed weight-gradient job cut into row slices for the fused processor's launches - every row
    exactly once, slices 16-row aligned, operand pointers advanced by whole rows; and the number of parts a two-bucket step
    allows (ops._PROCESSOR_HOOK = (callback, complete_by)): only the launches up to the early bucket's reduce."""
    from position_induced_transformer_amd import _lib, ops
    rows, n0, n1, n2 = 14792, 128, 64, 1
    job = _lib.MlpParamsJob(0x10000000, n0, rows, n0, n1, n2, 0x20000000, 0, 0x30000000, n2, 1, 2, 3, 4, 1, 0x40000000, 0)
    for parts in (1, 2, 3, 4):
        sl = ops._dw_slices((job,), parts)
        assert len(sl) <= parts and sum(s.rows for s in sl) == rows
        r0 = 0
        for s in sl:
            assert r0 % 16 == 0 and s.accumulate == 1 and s.out_gelu == 0
            assert s.x == job.x + 4 * r0 * n0 and s.h == job.h + 4 * r0 * n1
            assert s.d_y == job.d_y + 4 * r0 * n2 and s.scratch == job.scratch + 4 * r0 * n1
            assert (s.d_w1, s.d_b1, s.d_w2, s.d_b2) == (1, 2, 3, 4)
            r0 += s.rows
    # the number of launches that may carry a slice: all n blocks, or n - complete_by under a two-bucket step
    n = 4
    for hook, want in ((None, 4), ((lambda i: None, 2), 2), ((lambda i: None, 3), 1), ((lambda i: None, 0), 4)):
        parts = n if hook is None else max(1, n - hook[1])
        assert parts == want


def test_fused_processor_gate_refuses_more_blocks_than_one_weights_launch_forms():
    """pit_block_weights takes at most ops.BLOCK_MAX_LAYERS (16) layers; pit._fused_processor must answer None
    for a deeper processor (the reference accepts any n_blocks) BEFORE anything touches the device."""
    import torch
    from position_induced_transformer_amd import ops, pit as P
    assert ops.BLOCK_MAX_LAYERS == 16
    mesh = torch.rand(256, 2)
    model = P.pit_fixed(2, 1, 1, 64, 2, 17, mesh, 0.02, 0.02)
    calls = []
    orig = ops.block_fusion_supported
    ops.block_fusion_supported = lambda *a: calls.append(a) or True
    try:
        assert model._fused_processor(torch.zeros(2, 256, 64), model.mesh_ltt) is None
    finally:
        ops.block_fusion_supported = orig
    assert not calls
