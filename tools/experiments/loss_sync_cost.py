#!/usr/bin/env python3
"""experiment: what the loss's device->host copy in the middle of a training step costs (YOLOLoss returns its five components as
Python floats, like the reference: the host blocks until the forward + loss have run, with nothing queued behind them)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from yolo import YOLOv1, YOLOLoss
from yolo import loss as loss_mod
from yolo.optim import Adam
dev = torch.device("cuda")
model = YOLOv1().to(dev).train()
x = torch.randn(64, 3, 448, 448, device=dev)
tgt = torch.from_numpy(synth.synth_targets(64, seed=1)).to(dev)
crit = YOLOLoss()
opt = Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, max_grad_norm=10.0)
opt.attach_plan(model.hip_plan())


def nosync_forward(self, predictions, targets):
    total, out = loss_mod._HipLossFn.apply(predictions, targets, self.S, self.B, self.C, float(self.lambda_coord), float(self.lambda_noobj))
    return total, {}


eager = YOLOLoss.forward


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = crit(model(x), tgt)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
for rnd in range(3):
    for name, fn in (("float dict (sync)", eager), ("no host copy", nosync_forward)):
        YOLOLoss.forward = fn
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        print(f"{name}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)
