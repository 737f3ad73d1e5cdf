"""Data side of the reference's surface (src/yolo/dataset.py).

Only the target layout matters to the hot path: ``encode_target`` restates
``VOCDetectionYOLO._encode_target`` (dataset.py:487-532).  The PASCAL-VOC readers of the reference
need torchvision + a network download (kagglehub) and are outside this build's scope (SURVEY.md
section 2 row 9); the class names exist so that ``from yolo import VOCDetectionYOLO`` resolves and
fail with a clear message when constructed.  ``SyntheticYOLODataset`` feeds benchmarks and tests.
"""

from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset


def encode_target(bboxes, class_ids, S: int = 7, B: int = 2, C: int = 20) -> torch.Tensor:
    """Boxes (x_center, y_center, w, h in [0,1]) + class ids -> (S, S, 5B+C) target.
    The first object that lands in a cell owns it; only box slot 0 is filled; class is one-hot."""
    t = torch.zeros((S, S, 5 * B + C))
    for (xc, yc, w, h), cid in zip(bboxes, class_ids):
        i = min(int(S * yc), S - 1)
        j = min(int(S * xc), S - 1)
        if t[i, j, 4] == 0:
            t[i, j, 0] = S * xc - j
            t[i, j, 1] = S * yc - i
            t[i, j, 2] = w
            t[i, j, 3] = h
            t[i, j, 4] = 1.0
            t[i, j, 5 * B + cid] = 1.0
    return t


class SyntheticYOLODataset(Dataset):
    """Random 448x448 images ~N(0,1) with 0..max_obj encoded objects (the benchmark input of SURVEY.md 8d)."""

    def __init__(self, length: int = 256, S: int = 7, B: int = 2, C: int = 20, max_obj: int = 3, seed: int = 0, size: int = 448):
        self.length, self.S, self.B, self.C, self.max_obj, self.seed, self.size = length, S, B, C, max_obj, seed, size

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int):
        rng = np.random.Generator(np.random.PCG64([self.seed, idx]))
        img = torch.from_numpy(rng.standard_normal((3, self.size, self.size), dtype=np.float32))
        k = int(rng.integers(0, self.max_obj + 1))
        boxes = [(*rng.uniform(0, 1, 2), *rng.uniform(0.05, 0.9, 2)) for _ in range(k)]
        cids = [int(rng.integers(0, self.C)) for _ in range(k)]
        return img, encode_target(boxes, cids, self.S, self.B, self.C)


def _needs_torchvision(name: str):
    raise ImportError(f"{name} reads PASCAL VOC through torchvision/kagglehub, which this build does not vendor; "
                      "use yolo.dataset.SyntheticYOLODataset or bring your own Dataset yielding (image, target)")


class VOCDetectionYOLO(Dataset):
    def __init__(self, *args, **kwargs):
        _needs_torchvision("VOCDetectionYOLO")


class CombinedVOCDataset(Dataset):
    def __init__(self, *args, **kwargs):
        _needs_torchvision("CombinedVOCDataset")


def create_voc_datasets(*args, **kwargs):
    _needs_torchvision("create_voc_datasets")
