#!/usr/bin/env python3
"""mAP evaluation of a checkpoint (the reference's src/evaluate.py surface; defaults batch 16,
conf 0.01, nms 0.4 as in src/evaluate.py:18-95) with an additive --backbone / --synthetic flag."""

from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from yolo import YOLOv1, ResNetBackbone, YOLOv1Backbone, evaluate_model  # noqa: E402
from yolo.dataset import SyntheticYOLODataset, create_voc_datasets  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--backbone", choices=["resnet50", "yolov1"], default="resnet50")
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--conf-threshold", type=float, default=0.01)
    ap.add_argument("--nms-threshold", type=float, default=0.4)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--output", default="evaluation_results.txt")
    a = ap.parse_args()
    ds = SyntheticYOLODataset(a.synthetic, seed=2) if a.synthetic else create_voc_datasets([("2007", "test")], augment=False)
    loader = DataLoader(ds, batch_size=a.batch_size, shuffle=False, num_workers=4)
    bb = YOLOv1Backbone() if a.backbone == "yolov1" else ResNetBackbone(pretrained=False)
    model = YOLOv1(backbone=bb, num_classes=20)
    if a.checkpoint:
        model.load_state_dict(torch.load(a.checkpoint, map_location=a.device, weights_only=True)["model_state_dict"])
    model = model.to(a.device)
    res = evaluate_model(model, loader, a.device, num_classes=20, conf_threshold=a.conf_threshold, nms_threshold=a.nms_threshold)
    lines = [f"{k}: {float(v):.6f}" for k, v in res.items()]
    print("\n".join(lines[:12]))
    with open(a.output, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
