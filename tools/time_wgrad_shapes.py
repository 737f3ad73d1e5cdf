#!/usr/bin/env python3
"""micro-benchmark: yolo_wgrad on arbitrary conv shapes at N=64 (the ResNet-50 trunk's layers by default) over pixel-split choices.
   SHAPES="co,ci,k,hw;..."  SPLITS="0,64,128,256,512" VARIANT=0 python tools/time_wgrad_shapes.py
prints ms, TFLOP/s and the operand bytes / ms (dy + x read once) per split; split 0 = the library's two-segment schedule."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import lib, check, ptr, stream, WgradDesc
from yolo.engine import Act

N = 64
default = "256,64,1,112;64,256,1,112;64,64,3,112;512,128,1,56;128,512,1,56;128,128,3,56;1024,256,1,28;256,1024,1,28;256,256,3,28;2048,512,1,14;512,2048,1,14;512,512,3,14"
shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ.get("SHAPES", default).split(";")]
splits = [int(v) for v in os.environ.get("SPLITS", "0,32,64,128,256,512").split(",")]
V = int(os.environ.get("VARIANT", "0"))
dev = torch.device("cuda")
for co, ci, k, h in shapes:
    p = k // 2
    x = Act(N, h, h, ci, 1, dev); dy = Act(N, h, h, co, 1, dev)
    x.t.normal_(); dy.t.normal_()
    dwp = torch.zeros((co, k, k, ci), dtype=torch.float32, device=dev)
    res = {}
    for sp in splits:
        geo = k == 3 and h <= 28
        wd = (WgradDesc(N * h * h, dy.px_stride, x.px_stride, co, ci, k, k, p, x.row_stride, sp, 0, V, h, h, dy.Hp * dy.Wp, dy.Wp, 1, dy.Wp + 1) if geo or k == 1 and False
              else WgradDesc(dy.slots, dy.px_stride, x.px_stride, co, ci, k, k, p, x.row_stride, sp, 0, V))
        ts = []
        for rep in range(3):
            for _ in range(2):
                check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dwp), None, stream()))
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dwp), None, stream()))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        res[sp] = min(ts)
    fl = 2.0 * N * h * h * co * ci * k * k
    by = 2.0 * N * h * h * (co + ci)
    print(f"co {co:4d} ci {ci:4d} k {k} {h:3d}x{h:<3d} " + "  ".join(f"s{sp}: {t:6.3f} ms {fl / t / 1e9:5.0f} TF {by / t / 1e9:5.2f} TB/s" for sp, t in res.items()))
    del x, dy
