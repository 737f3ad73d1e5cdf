"""oracle/torch_ref.py -- TEST INFRASTRUCTURE.  Stock torch.nn restatement of the reference's networks.

The reference builds its models out of stock ``nn.Conv2d`` / ``nn.LeakyReLU(0.1)`` / ``nn.MaxPool2d(2,2)``
/ ``nn.Linear`` / ``nn.Dropout(0.5)`` (src/yolo/models.py:47-84 backbone, :239-245 FC head,
:313-332 DetectionHead), so stock torch on the host CPU *is* the reference's arithmetic for the
conv stack.  This file restates the layer table (it cannot import the reference on the GPU box) and is
pinned by tests/golden/backbone_full.npz, which was produced by the reference's own YOLOv1().
Used by tests/ and by bench.py's cpu_baseline leg only.
"""

from __future__ import annotations

import torch
import torch.nn as nn

# (state_dict index, out_ch, in_ch, kernel, stride, pad) | "M"      -- src/yolo/models.py:47-84
BACKBONE = [
    (0, 64, 3, 7, 2, 3), "M",
    (3, 192, 64, 3, 1, 1), "M",
    (6, 128, 192, 1, 1, 0), (8, 256, 128, 3, 1, 1), (10, 256, 256, 1, 1, 0), (12, 512, 256, 3, 1, 1), "M",
    (15, 256, 512, 1, 1, 0), (17, 512, 256, 3, 1, 1), (19, 256, 512, 1, 1, 0), (21, 512, 256, 3, 1, 1),
    (23, 256, 512, 1, 1, 0), (25, 512, 256, 3, 1, 1), (27, 256, 512, 1, 1, 0), (29, 512, 256, 3, 1, 1),
    (31, 512, 512, 1, 1, 0), (33, 1024, 512, 3, 1, 1), "M",
    (36, 512, 1024, 1, 1, 0), (38, 1024, 512, 3, 1, 1), (40, 512, 1024, 1, 1, 0), (42, 1024, 512, 3, 1, 1),
    (44, 1024, 1024, 3, 1, 1), (46, 1024, 1024, 3, 2, 1), (48, 1024, 1024, 3, 1, 1), (50, 1024, 1024, 3, 1, 1),
]


class RefYOLOv1(nn.Module):
    """YOLOv1Backbone + FC head with the reference's state_dict keys (backbone.features.N / head.N)."""

    def __init__(self, S: int = 7, B: int = 2, C: int = 20):
        super().__init__()
        mods = []
        for item in BACKBONE:
            if item == "M":
                mods.append(nn.MaxPool2d(2, 2))
            else:
                _idx, co, ci, k, s, p = item
                assert _idx == len(mods)
                mods += [nn.Conv2d(ci, co, k, s, p), nn.LeakyReLU(0.1)]
        self.backbone = nn.Module()
        self.backbone.features = nn.Sequential(*mods)
        self.head = nn.Sequential(nn.Flatten(), nn.Linear(1024 * S * S, 4096), nn.LeakyReLU(0.1), nn.Dropout(0.5),
                                  nn.Linear(4096, S * S * (B * 5 + C)))
        self.S, self.B, self.C = S, B, C

    def forward(self, x):
        return self.head(self.backbone.features(x)).view(-1, self.S, self.S, self.B * 5 + self.C)


def conv_macs_per_image() -> int:
    """multiply-accumulates of one 448x448 forward (conv + FC), = 20.285 GMAC."""
    h = 448
    macs = 0
    for item in BACKBONE:
        if item == "M":
            h //= 2
            continue
        _i, co, ci, k, s, p = item
        h = (h + 2 * p - k) // s + 1
        macs += h * h * co * ci * k * k
    macs += 1024 * 49 * 4096 + 4096 * 1470
    return macs
