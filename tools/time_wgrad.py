#!/usr/bin/env python3
"""micro-benchmark: yolo_wgrad (weight part / bias part separately) on the YOLOv1 layer shapes at N=64."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo._hip import lib, check, ptr, stream, WgradDesc
from yolo.engine import Act

N = 64
VARIANTS = [int(v) for v in os.environ.get("VARIANT", "0").split(",")]     # several: interleaved A/B in one process
INDEX = int(os.environ.get("INDEX", "0"))
SPLITS = [int(v) for v in os.environ.get("SPLITS", "").split(",") if v]
ONLY = [int(v) for v in os.environ.get("LAYERS", "").split(",") if v]
dev = torch.device("cuda")
h = 448
rows = []
for item in synth.YOLOV1_BACKBONE_CFG:
    if item == "M":
        h //= 2
        continue
    idx, (co, ci, k, s, p) = item
    hin = h
    h = (h + 2 * p - k) // s + 1
    if idx == 0 or (ONLY and idx not in ONLY):
        continue
    geo = hin if s == 2 else h
    x = Act(N, geo, geo, ci, 1, dev)
    dy = Act(N, geo, geo, co, 1, dev)
    x.t.normal_(); dy.t.normal_()
    dwp = torch.zeros((co, k, k, ci), dtype=torch.float32, device=dev)
    db = torch.zeros((co,), dtype=torch.float32, device=dev)
    tiles = ((co + 127) // 128) * ((ci + 127) // 128) * k * k
    for split in (SPLITS or sorted(set([max(1, min(dy.slots // 256, (1024 + tiles - 1) // tiles)), max(1, (256 + tiles - 1) // tiles), max(1, (512 + tiles - 1) // tiles)]))):
        wds, modes = {}, []
        for V in VARIANTS:
            wds[f"v{V} flat"] = WgradDesc(dy.slots, dy.px_stride, x.px_stride, co, ci, k, k, p, x.row_stride, split, 0, V)
            wds[f"v{V} geo"] = WgradDesc(N * h * h, dy.px_stride, x.px_stride, co, ci, k, k, p, x.row_stride, split, 0, V, h, h, dy.Hp * dy.Wp, dy.Wp * s, s, dy.Wp + 1)
            modes += [f"v{V} flat", f"v{V} geo"] if INDEX == 2 else ([f"v{V} geo"] if INDEX else [f"v{V} flat"])
        res = {m: [] for m in modes}
        for rep in range(3):                      # interleaved A/B inside one process
            for m in modes:
                wd = wds[m]
                for _ in range(2):
                    check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dwp), ptr(db), stream()))
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dwp), ptr(db), stream()))
                e1.record(); torch.cuda.synchronize()
                res[m].append(e0.elapsed_time(e1) / 10)
        fl = 2.0 * N * h * h * co * ci * k * k
        print(f"idx {idx:2d} co {co:4d} ci {ci:4d} k {k} s {s} geo {geo:3d} tiles {tiles:4d} split {split:4d}  " +
              "  ".join(f"{m} {min(v):7.3f} ms ({fl / min(v) / 1e9:6.1f} TF)" for m, v in res.items()))
    del x, dy
