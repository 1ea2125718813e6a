"""Per-parameter relative-L2 error of the bf16 math mode against the golden (fp32 reference)
gradients, for the four model fixtures.  Diagnostic; run on the GPU box from the repo root."""
import sys
sys.path[:0] = [".", "tests", "oracle"]
import numpy as np
import torch
import golden_io as gio
import model_cases as mc
from test_gpu_models import build_model
from position_induced_transformer_amd import ops, utils

for name in mc.CASES:
    fx = gio.load(name)
    cs = mc.build_case(name)
    params = gio.synth_params(cs["shapes"], int(fx["param_seed"]))
    for mode in ("fp32", "bf16"):
        with ops.math_mode(mode):
            model = build_model(cs, params)
            out = model(cs["mesh_in"].cuda(), cs["func_in"].cuda(), cs["mesh_out"].cuda())
            loss = utils.RelLpNorm(cs["cfg"]["out_dim"], cs["p_norm"])(cs["target"].cuda(), out)
            loss.backward()
        e, g, _, _ = gio.expect(fx, "out", out.detach().cpu().numpy())
        line = [f"{name} {mode}: out {gio.rel_l2(e, g):.2e}"]
        for k, p in model.named_parameters():
            e, g, _, _ = gio.expect(fx, "grad/" + k, p.grad.cpu().numpy())
            if k.endswith("lmda"):
                line.append(f"{k} {gio.rel_l2(e, g):.1e} (|g|={np.abs(e).max():.1e})")
        worst = max(gio.rel_l2(*gio.expect(fx, "grad/" + k, p.grad.cpu().numpy())[:2]) for k, p in model.named_parameters() if not k.endswith("lmda"))
        line.append(f"worst non-lmda {worst:.1e}")
        print("  ".join(line))
