#!/usr/bin/env python3
"""Measure the yolo_igemm launch plans of every problem of the BASELINE configurations on this GPU and write them to
yolo-v1_amd/yolo/plans/gfx950.json (the table every process loads at import; engine.load_plans).

    python tools/tune_plans.py [--batch 64] [--no-resnet] [--out PATH] [--fresh]

Runs: YOLOv1 inference forward, YOLOv1 training step (forward + backward), YOLOv1(ResNetBackbone) inference and training
step -- BASELINE.json configs[1], [2]/[3] and [4] -- at the given per-GPU batch.  --fresh ignores the shipped table."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import synth  # noqa: E402
from yolo import ResNetBackbone, YOLOLoss, YOLOv1, engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-resnet", action="store_true")
    ap.add_argument("--out", default=engine.PLAN_FILE)
    ap.add_argument("--fresh", action="store_true")
    ap.add_argument("--reps", type=int, default=1, help="launches per timed sample (1: one launch behind a cache flush; 8: back-to-back, warm caches)")
    a = ap.parse_args()
    engine.TUNE_REPS = a.reps
    if a.fresh:
        engine._TUNED.clear()
    engine.AUTOTUNE = True
    engine.TUNE_LOG = []
    dev = torch.device("cuda")
    B = a.batch
    torch.manual_seed(0)
    x = torch.randn(B, 3, 448, 448, device=dev)
    tgt = torch.from_numpy(synth.synth_targets(B, 1)).to(dev)
    crit = YOLOLoss()

    def run(model, train):
        if train:
            model.train()
            loss, _ = crit(model(x), tgt)
            loss.backward()
        else:
            model.eval()
            with torch.no_grad():
                model(x)
        torch.cuda.synchronize()

    m = YOLOv1().to(dev)
    run(m, False)
    print(f"YOLOv1 inference: {len(engine._TUNED)} problems", flush=True)
    run(m, True)
    print(f"YOLOv1 training: {len(engine._TUNED)} problems", flush=True)
    del m
    torch.cuda.empty_cache()
    if not a.no_resnet:
        r = YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=True)).to(dev)
        run(r, False)
        print(f"ResNet variant inference: {len(engine._TUNED)} problems", flush=True)
        del r
        torch.cuda.empty_cache()
        t = YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=False)).to(dev)
        run(t, True)
        print(f"ResNet variant training: {len(engine._TUNED)} problems", flush=True)
    for key, best, top in engine.TUNE_LOG:
        print(engine._key_str(key), "->", best, "|", "  ".join(f"{pl}: {t * 1e3:.1f}us" for pl, t in top))
    engine.save_plans(a.out, note=f"measured by tools/tune_plans.py at batch {B} on {torch.cuda.get_device_name(0)}")
    print(f"wrote {len(engine._TUNED)} plans to {a.out}")


if __name__ == "__main__":
    main()
