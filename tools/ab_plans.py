#!/usr/bin/env python3
"""A/B of two launch-plan tables on the batch-64 inference forward (YOLOv1, and the ResNet-50 variant with MODEL=resnet50) inside ONE process;
TRAIN=1: on forward + loss + backward of the training mode instead (ResNet: trainable trunk).
usage: ab_plans.py TABLE_A.json TABLE_B.json [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import YOLOv1, ResNetBackbone, engine

tables = sys.argv[1:3]
for t in tables:
    if not os.path.exists(t):
        sys.exit(f"ab_plans: {t} does not exist (gpurun_out/ does not travel to the GPU box: keep tables to compare under build/)")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
resnet = os.environ.get("MODEL", "yolov1") == "resnet50"
train = os.environ.get("TRAIN", "0") == "1"
m = (YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=not train)) if resnet else YOLOv1()).cuda()
m = m.train() if train else m.eval()
x = torch.randn(int(os.environ.get("BATCH", "64")), 3, 448, 448, device="cuda")
if train:
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    from yolo import YOLOLoss
    tgt = torch.from_numpy(synth.synth_targets(int(os.environ.get("BATCH", "64")), seed=1)).cuda()
    crit = YOLOLoss()
    _fwd = m

    def m(xx):
        for p_ in _fwd.parameters():
            p_.grad = None
        loss, _ = crit(_fwd(xx), tgt)
        loss.backward()
res = {t: [] for t in tables}
with torch.set_grad_enabled(train):
    for rnd in range(3):
        for t in tables:
            engine._TUNED.clear()
            engine.load_plans(t)
            if not train and not resnet:
                m.hip_plan()._ws.clear()
            for _ in range(10):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                m(x)
            torch.cuda.synchronize()
            res[t].append(1e3 * (time.perf_counter() - t0) / reps)
for t, v in res.items():
    print(f"{os.path.basename(t)}: " + " ".join(f"{a:.4f}" for a in v) + f"  ms/forward (min {min(v):.4f})")
