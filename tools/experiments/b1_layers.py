"""Per-launch times of one YOLOv1 forward at batch BATCH (default 1) (engine.TIMERS) next to the shipped batch-1 plans of the deep layers: how the 90 us
few-pixel layers of the untuned table were found.

    python tools/experiments/b1_layers.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, engine
m = YOLOv1().cuda().eval()
B = int(os.environ.get("BATCH", "1"))
x = torch.randn(B, 3, 448, 448, device="cuda")
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize()
    best = None
    for _ in range(5):          # min over five passes: a single pass of ~20 us launches is at the mercy of the clock ramp
        engine.TIMERS = []
        m(x); torch.cuda.synchronize()
        ts = [(tag, k, fl, e0.elapsed_time(e1)) for (tag, k, fl, e0, e1) in engine.TIMERS]
        best = ts if best is None else [(a[0], a[1], a[2], min(a[3], b[3])) for a, b in zip(best, ts)]
rows = best
engine.TIMERS = None
for tag, k, fl, ms in rows: print(f"{tag:14s} {k:8s} {ms*1e3:7.1f} us {fl/ms/1e9 if fl else 0:6.0f} TF")
print("sum", sum(r[3] for r in rows))
import json
d = json.load(open(os.path.join(ROOT, "yolo-v1_amd/yolo/plans/gfx950.json")))["plans"]
for k, v in d.items():
    if k.startswith(f"{B},") and (",3,3,1024," in k or ",3,3,512," in k): print(k, v)
