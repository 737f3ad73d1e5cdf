#!/usr/bin/env python3
"""micro-benchmark of yolo_conv_stem7_fwd at batch 64 (pool fused / not fused)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import lib, check, ptr, stream
from yolo.engine import Act
N = 64
dev = torch.device("cuda")
x = Act(N, 448, 448, 4, 3, dev); x.t.normal_()
w = torch.randn(64, 7, 8, 4, device=dev).to(torch.bfloat16)
b = torch.randn(64, device=dev)
for pool in (1, 0):
    out = Act(N, 112 if pool else 224, 112 if pool else 224, 64, 1, dev)
    def run():
        check(lib().yolo_conv_stem7_fwd(x.p, ptr(w), ptr(b), N, 224, 224, x.img_stride, x.row_stride, 0.1, pool, out.p, out.img_stride, out.row_stride, out.interior_off(), None, 0, 0, 0, stream()))
    for _ in range(3): run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"pool {pool}: {ms:.4f} ms  {2.0 * N * 224 * 224 * 64 * 147 / ms / 1e9:.0f} TF  out+in {(x.t.numel() * 2 + out.t.numel() * 2) / ms / 1e6:.0f} GB/s")
