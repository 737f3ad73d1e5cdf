#!/usr/bin/env python3
"""Forward time of YOLOv1 (and of the ResNet-50 variant) at the batch sizes the shipped entry points use: 1 (predict.py), 16 (evaluate.py's
default), 32, 64 (the benchmarked size) and a ragged 13 (the last batch of a DataLoader without drop_last: no measured plans -- borrowed from
batch 16, plans._borrowed_plan).  Prints a markdown table: ms per batch, images/s, TFLOP/s, per-image time relative to the LAST batch size of the
list, and how many of the batch's yolo_igemm problems ran on a measured plan of yolo/plans/gfx950.json.

    [BATCHES=1,2,4,8,13,16,32,64] [NO_SMALL_SPLIT=1] [YOLO_AMD_BORROW_PLANS=0] python tools/batch_sweep.py [--resnet]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, ResNetBackbone, engine

resnet = "--resnet" in sys.argv
dev = torch.device("cuda")
torch.manual_seed(0)
model = (YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=True)) if resnet else YOLOv1()).to(dev).eval()
gf = 40.57 if not resnet else None
shipped = set(engine._TUNED)
if os.environ.get("NO_SMALL_SPLIT") == "1":      # A/B of the default rule for few-pixel deep-K problems: one plain launch, as before round 3
    from yolo import plans
    _dp = plans._default_plan
    plans._default_plan = lambda d: (0, 0) if (not d.pool2 and d.N * d.Ho * d.Wo < 2048) else _dp(d)
rows = []
for B in tuple(int(b) for b in os.environ.get("BATCHES", "1,13,16,32,64").split(",")):
    x = torch.randn(B, 3, 448, 448, device=dev)
    before = set(engine._TUNED)
    with torch.no_grad():
        for _ in range(5):
            model(x)
        torch.cuda.synchronize()
        n = max(20, int(200 / B))
        t0 = time.perf_counter()
        for _ in range(n):
            model(x)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    keys = [k for k in engine._TUNED if k[0] == B]
    measured = sum(1 for k in keys if k in shipped)
    rows.append((B, ms, B / ms * 1e3, (gf * B / ms) if gf else None, ms / B, measured, len(keys)))
ref = rows[-1][4]
print(f"| batch | ms / batch | images/s | TFLOP/s | per-image time vs batch 64 | problems on a measured plan |")
print("|---|---|---|---|---|---|")
for B, ms, ips, tf, per, meas, tot in rows:
    print(f"| {B} | {ms:.3f} | {ips:.0f} | {'%.0f' % tf if tf else '-'} | {per / ref:.2f} x | {meas} / {tot} |")
