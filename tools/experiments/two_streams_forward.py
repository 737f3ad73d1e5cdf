#!/usr/bin/env python3
"""experiment: a batch of 64 as two half-batches of 32 on two streams (their prologue / K loop / epilogue phases interleave)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, engine

m = YOLOv1().cuda().eval()
x = torch.randn(64, 3, 448, 448, device="cuda")
xa, xb = x[:32].contiguous(), x[32:].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def full():
    return m(x)


def seq():
    return m(xa), m(xb)


def conc():
    main = torch.cuda.current_stream()
    sa.wait_stream(main); sb.wait_stream(main)
    with torch.cuda.stream(sa):
        ya = m(xa)
    with torch.cuda.stream(sb):
        yb = m(xb)
    main.wait_stream(sa); main.wait_stream(sb)
    return ya, yb


def timeit(fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


with torch.no_grad():
    if "--tune" in sys.argv:          # measure launch plans for the batch-32 problems first (in this process only)
        engine.AUTOTUNE = True
        m(xa); torch.cuda.synchronize()
        engine.AUTOTUNE = False
        print("tuned", len(engine._TUNED), "plans")
    for rnd in range(3):
        print(f"batch 64: {timeit(full):.3f} ms   2 x 32 sequential: {timeit(seq):.3f} ms   2 x 32 on two streams: {timeit(conc):.3f} ms")
    y = full(); ya, yb = conc(); torch.cuda.synchronize()
    print("max |diff|", (y - torch.cat([ya, yb])).abs().max().item())
