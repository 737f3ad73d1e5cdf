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
VARIANT = int(os.environ.get("VARIANT", "0"))
MAP = int(os.environ.get("MAP", "0"))
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
        wd = WgradDesc(dy.slots, dy.px_stride, x.px_stride, co, ci, k, k, p, x.row_stride, split, 0, VARIANT)
        def run(w, b):
            check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(w) if w is not None else None, ptr(b) if b is not None else None, stream()))
        res = []
        for w, b in ((dwp, None), (None, db)):
            for _ in range(2): run(w, b)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(w, b)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 5)
        fl = 2.0 * N * h * h * co * ci * k * k * (4 if s == 2 else 1) / (4 if s == 2 else 1)
        print(f"idx {idx:2d} co {co:4d} ci {ci:4d} k {k} s {s} geo {geo:3d} tiles {tiles:4d} split {split:4d}  wgrad {res[0]:7.3f} ms ({fl/res[0]/1e9:6.1f} TF)  colsum {res[1]:7.3f} ms ({dy.slots*co*2/res[1]/1e6:7.1f} GB/s)")
    del x, dy
