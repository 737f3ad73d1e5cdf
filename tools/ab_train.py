#!/usr/bin/env python3
"""A/B of engine switches on the batch-64 YOLOv1 training step inside ONE process (boxes differ by +-3 %).
usage: ab_train.py NAME=v0,v1 [steps]   e.g.  ab_train.py WGRAD_STREAM=False,True 20"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo import YOLOv1, YOLOLoss, ResNetBackbone, engine, optim
from yolo.optim import Adam

name, _, vals = sys.argv[1].partition("=")
vals = [eval(v) for v in vals.split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
torch.manual_seed(0)
resnet = os.environ.get("MODEL", "yolov1") == "resnet50"      # the reference's default training model: ResNet-50 trunk, not frozen
model = (YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=False)) if resnet else YOLOv1()).to(dev).train()
x = torch.randn(64, 3, 448, 448, device=dev)
tgt = torch.from_numpy(synth.synth_targets(64, seed=1)).to(dev)
crit = YOLOLoss()
opt = Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, max_grad_norm=10.0)
opt.attach_plan(model.head.hip_plan() if resnet else model.hip_plan(), overlap=os.environ.get("OVERLAP", "0") == "1")


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = crit(model(x), tgt)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
res = {repr(v): [] for v in vals}
for rnd in range(3):
    for v in vals:
        setattr(optim if name.startswith("optim.") else engine, name.split(".")[-1], v)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res[repr(v)].append(1e3 * (time.perf_counter() - t0) / steps)
for k, v in res.items():
    print(f"{name}={k}: " + " ".join(f"{t:.3f}" for t in v) + f"  ms/step (min {min(v):.3f})")
