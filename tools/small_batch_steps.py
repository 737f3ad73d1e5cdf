#!/usr/bin/env python3
"""N forwards of YOLOv1 at a small batch (predict.py's single image by default) -- workload for rocprofv3 --kernel-trace --stats

    python tools/small_batch_steps.py [N=200] [BATCH=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
m = YOLOv1().cuda().eval()
x = torch.randn(B, 3, 448, 448, device="cuda")
with torch.no_grad():
    for _ in range(10):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m(x)
    torch.cuda.synchronize()
print(f"batch {B}: {1e3 * (time.perf_counter() - t0) / steps:.4f} ms per forward over {steps}")
