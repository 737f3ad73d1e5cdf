#!/usr/bin/env python3
"""per-conv time of YOLOv1(ResNetBackbone) inference at batch 64 with the tuned launch plans (engine.TIMERS)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, ResNetBackbone, engine
engine.TILE_HINT = int(os.environ.get("TILE_HINT", "0"))
m = YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=True)).cuda().eval()
x = torch.randn(64, 3, 448, 448, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    engine.TIMERS = []
    m(x)
    torch.cuda.synchronize()
rows = [(tag, k, fl, e0.elapsed_time(e1)) for (tag, k, fl, e0, e1) in engine.TIMERS]
engine.TIMERS = None
tot = sum(r[3] for r in rows)
convs = {str(k): c for k, c in [((li, bi, j), getattr(blk, f"conv{j}")) for li in range(4, 8) for bi, blk in enumerate(m.backbone.extractor[li]) for j in (1, 2, 3)]}
for tag, k, fl, ms in rows:
    c = convs.get(tag)
    desc = f"{c.in_channels:4d}->{c.out_channels:4d} k{c.kernel_size[0]} s{c.stride[0]}" if c is not None else ""
    print(f"{tag:16s} {k:8s} {desc:22s} {ms*1e3:8.1f} us  {fl/ms/1e9 if fl else 0:7.0f} TF")
print("sum ms", tot)
