#!/usr/bin/env python3
"""A/B of an engine switch on the batch-64 YOLOv1 inference forward inside ONE process.  usage: ab_forward.py NAME=v0,v1 [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, engine

name, _, vals = sys.argv[1].partition("=")
vals = [eval(v) for v in vals.split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = YOLOv1().cuda().eval()
x = torch.randn(64, 3, 448, 448, device="cuda")
res = {repr(v): [] for v in vals}
outs = {}
with torch.no_grad():
    for rnd in range(3):
        for v in vals:
            setattr(engine, name, v)
            m.hip_plan()._ws.clear()            # workspaces are built per switch setting
            for _ in range(10):
                y = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                y = m(x)
            torch.cuda.synchronize()
            res[repr(v)].append(1e3 * (time.perf_counter() - t0) / reps)
            outs[repr(v)] = y.clone()
for k, v in res.items():
    print(f"{name}={k}: " + " ".join(f"{t:.4f}" for t in v) + f"  ms/forward (min {min(v):.4f})")
ks = list(outs)
if len(ks) > 1:
    print("outputs equal:", all(torch.equal(outs[ks[0]], outs[k]) for k in ks[1:]), " max |diff|", max((outs[ks[0]] - outs[k]).abs().max().item() for k in ks[1:]))
