#!/usr/bin/env python3
"""experiment: inference forward (N=64 by default) replayed from a HIP graph vs launched eagerly (same process)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1

m = YOLOv1().cuda().eval()
x = torch.randn(int(os.environ.get("N", "64")), 3, 448, 448, device="cuda")
with torch.no_grad():
    for _ in range(5):
        y0 = m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            m(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        yg = m(x)
    torch.cuda.synchronize()

    def timeit(fn, reps=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    for rnd in range(3):
        print(f"eager {timeit(lambda: m(x)):.4f} ms   graph {timeit(g.replay):.4f} ms")
    g.replay(); torch.cuda.synchronize()
    print("same result:", torch.equal(yg, m(x)))
