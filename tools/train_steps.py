#!/usr/bin/env python3
"""N training steps of YOLOv1 at batch 64 (nothing else): the workload for `rocprofv3 --kernel-trace --stats`
when only the train step's kernel mix is wanted.  usage: train_steps.py [steps]
MODEL=resnet50 runs the reference's default training model instead (ResNet-50 trunk, not frozen, + DetectionHead)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import synth
from yolo import YOLOv1, YOLOLoss, ResNetBackbone
from yolo.optim import Adam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
resnet = os.environ.get("MODEL", "yolov1") == "resnet50"
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


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step over {steps} steps (+3 warm-up)")
print(f"peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
