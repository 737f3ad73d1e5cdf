"""Per-launch times of one batch-1 YOLOv1 forward (engine.TIMERS) next to the shipped batch-1 plans of the deep layers: how the 90 us
few-pixel layers of the untuned table were found.

    python tools/experiments/b1_layers.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, engine
m = YOLOv1().cuda().eval()
x = torch.randn(1, 3, 448, 448, device="cuda")
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize()
    engine.TIMERS = []
    m(x); torch.cuda.synchronize()
rows = [(tag, k, fl, e0.elapsed_time(e1)) for (tag, k, fl, e0, e1) in engine.TIMERS]
engine.TIMERS = None
for tag, k, fl, ms in rows: print(f"{tag:14s} {k:8s} {ms*1e3:7.1f} us")
print("sum", sum(r[3] for r in rows))
import json
d = json.load(open(os.path.join(ROOT, "yolo-v1_amd/yolo/plans/gfx950.json")))["plans"]
for k, v in d.items():
    if k.startswith("1,") and (",3,3,1024," in k or ",3,3,512," in k): print(k, v)
