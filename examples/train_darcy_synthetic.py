#!/usr/bin/env python3
"""The reference's Darcy training loop (train_darcy.py:62-150) on synthetic fields, written the
way the reference writes it - `from pit import *`, `class pit_darcy(pit_fixed)` with its own
forward, Adam + CosineAnnealingLR, RelLpNorm - but importing the MI355X modules instead.

    python examples/train_darcy_synthetic.py [--epochs 2] [--graph]

`--graph` replaces the inner loop by the hipGraph-captured step (engine.TrainStep) with the
fused Adam; without it the loop is literally the reference's eager loop."""
import argparse
import os
import sys
from timeit import default_timer

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from position_induced_transformer_amd.pit import *      # noqa: F401,F403  (the reference does `from pit import *`)
from position_induced_transformer_amd.utils import *    # noqa: F401,F403


class pit_darcy(pit_fixed):                              # noqa: F405   train_darcy.py:25-59
    def forward(self, mesh_in, func_in, mesh_out):
        size = mesh_out.shape[:-1]
        mesh_in = mesh_in.reshape(-1, self.space_dim)
        func_in = func_in.reshape(func_in.shape[0], -1, self.in_dim)
        mesh_out = mesh_out.reshape(-1, self.space_dim)
        func_in = torch.cat((torch.tile(mesh_in.unsqueeze(0), [func_in.shape[0], 1, 1]), func_in), -1)  # noqa: F405
        func_ltt = self.encoder(mesh_in, func_in, self.mesh_ltt)
        func_ltt = self.processor(func_ltt, self.mesh_ltt)
        func_out = self.decoder(self.mesh_ltt, func_ltt, mesh_out)
        return func_out.reshape(func_in.shape[0], *size, self.out_dim)


def grid(s):
    m = np.vstack([xx.ravel() for xx in np.meshgrid(np.linspace(0, 1, s), np.linspace(0, 1, s))]).T  # noqa: F405
    return torch.tensor(m.reshape(s, s, 2), dtype=torch.float).cuda()  # noqa: F405


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--ntrain", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)  # noqa: F405
    s = 43
    # synthetic "dataset": smooth random coefficient fields and a fixed smoothing of them as targets
    x_train = torch.randn(args.ntrain, s, s, 1)  # noqa: F405
    k = torch.ones(1, 1, 5, 5) / 25.0  # noqa: F405
    y_train = torch.nn.functional.conv2d(x_train.permute(0, 3, 1, 2), k, padding=2).permute(0, 2, 3, 1)  # noqa: F405
    x_normalizer = PixelWiseNormalization(x_train)  # noqa: F405
    x_train = x_normalizer.normalize(x_train)
    y_normalizer = PixelWiseNormalization(y_train)  # noqa: F405
    mesh, mesh_ltt = grid(s), grid(16)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x_train, y_train),  # noqa: F405
                                         batch_size=args.batch, shuffle=True, drop_last=True)
    model = pit_darcy(2, 1, 1, 64, 2, 4, mesh_ltt, 0.02, 0.02).cuda()
    print("parameters:", count_params(model))  # noqa: F405
    iterations = args.epochs * (args.ntrain // args.batch)
    myloss = RelLpNorm(out_dim=1, p=2)  # noqa: F405
    y_normalizer.cuda()

    if args.graph:
        from position_induced_transformer_amd.ddp import FlatAdam, FlatGradients
        from position_induced_transformer_amd.engine import TrainStep
        flat = FlatGradients(model.parameters(), flatten_params=True)
        opt = FlatAdam(flat, lr=1e-3, cosine_t_max=iterations)
        x0, y0 = next(iter(loader))
        step = TrainStep(model, (mesh, x0.cuda(), mesh, y0.cuda()), 1, 2, pred_affine=y_normalizer.affine(),
                         optimizer=opt, flat=flat)
        step.capture()
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)  # noqa: F405
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=iterations)  # noqa: F405

    for ep in range(args.epochs):
        model.train()
        t1 = default_timer()
        train_l2 = torch.zeros((), device="cuda")  # noqa: F405
        for x, y in loader:
            x, y = x.cuda(), y.cuda()
            if args.graph:
                step.set_batch(x, y)
                step.replay()
                train_l2 += step.loss
            else:
                optimizer.zero_grad()
                out = model(mesh, x, mesh)
                out = y_normalizer.denormalize(out)
                loss = myloss(y, out)
                loss.backward()
                optimizer.step()
                scheduler.step()
                train_l2 += loss.detach()
        torch.cuda.synchronize()  # noqa: F405
        t2 = default_timer()
        print(ep, f"{t2 - t1:.3f}s", float(train_l2) / args.ntrain)
    torch.save({"model_state": model.state_dict()}, "/tmp/model.pth")  # noqa: F405


if __name__ == "__main__":
    main()
