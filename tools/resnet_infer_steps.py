#!/usr/bin/env python3
"""N inference passes of YOLOv1(ResNetBackbone) at batch 64 (BASELINE configs[4]) -- workload for rocprofv3 --kernel-trace"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, ResNetBackbone, ops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=True)).cuda().eval()
x = torch.randn(64, 3, 448, 448, device="cuda")
def f():
    with torch.no_grad():
        pr = m(x)
        rec, cnt = ops.decode(pr, 0.3, 7, 2, 20)
        return ops.nms(rec, cnt, 0.4, 0)
for _ in range(3):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    f()
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / steps:.3f} ms/pass over {steps} passes")
