#!/usr/bin/env python3
"""run the YOLOv1 forward (and optionally a training step) once at batch N and print the launch plan the autotuner
chose for every yolo_igemm problem signature"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import engine
from yolo.models import YOLOv1

N = int(os.environ.get("N", 64))
m = YOLOv1().cuda().eval()
x = torch.randn(N, 3, 448, 448, device="cuda")
with torch.no_grad():
    for _ in range(2):
        m(x)
torch.cuda.synchronize()
for k, v in engine._TUNED.items():
    print(f"N {k[0]} out {k[1]}x{k[2]} k {k[3]}x{k[4]} cin {k[5]} cout {k[6]} s {k[7]} epi {k[8]} pool {k[9]} -> {v}")

# A/B: the same process with the skew plans replaced by their un-skewed launch
def timeit(reps=20):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(reps):
            m(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

tuned = dict(engine._TUNED)
plain = {k: ((v[1], v[2]) if v[0] == "skew" else v) for k, v in tuned.items()}
for rnd in range(4):
    engine._TUNED.clear(); engine._TUNED.update(tuned)
    a = timeit()
    engine._TUNED.clear(); engine._TUNED.update(plain)
    b = timeit()
    print(f"round {rnd}: with skew plans {a:.3f} ms, without {b:.3f} ms per forward of {N}")
