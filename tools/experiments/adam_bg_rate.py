#!/usr/bin/env python3
"""experiment: rate of the background Adam pass (yolo_adam_step_multi_bg) on the FC1-sized tensor alone, by number of CUs held"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import AdamTensor, check, lib, ptr, stream

n = 4096 * 50176
dev = torch.device("cuda")
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
tab = (AdamTensor * 1)(AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), pb.data_ptr(), n))


def run(G):
    if G == 0:
        check(lib().yolo_adam_step_multi(tab, 1, 1e-4, 0.9, 0.999, 1e-8, 5e-4, 1, None, 0.0, stream()))
    else:
        check(lib().yolo_adam_step_multi_bg(tab, 1, 1e-4, 0.9, 0.999, 1e-8, 5e-4, 1, None, 0.0, G, stream()))


for G in (0, 256, 128, 96, 64, 48, 32):
    for _ in range(2):
        run(G)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(G)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{'foreground kernel' if G == 0 else f'background, {G:3d} CUs'}: {ms:.3f} ms  {n * 30 / ms / 1e9:.2f} TB/s")
