"""Deterministic synthetic tensors shared by the golden generator, the tests and bench.py.

Nothing here comes from the reference; it only fixes *which numbers* a test feeds to both
sides.  numpy's PCG64 stream is used (same numpy build in the dev container and on the GPU
box), never torch's RNG, so the values do not depend on the torch device or version.
"""

from __future__ import annotations

import numpy as np

# (out_channels, in_channels, kernel, stride, pad) or "M" for MaxPool2d(2, 2); state_dict index
# of each conv inside ``backbone.features`` is listed beside it (reference: src/yolo/models.py:47-84).
YOLOV1_BACKBONE_CFG = [
    (0, (64, 3, 7, 2, 3)), "M",
    (3, (192, 64, 3, 1, 1)), "M",
    (6, (128, 192, 1, 1, 0)), (8, (256, 128, 3, 1, 1)), (10, (256, 256, 1, 1, 0)), (12, (512, 256, 3, 1, 1)), "M",
    (15, (256, 512, 1, 1, 0)), (17, (512, 256, 3, 1, 1)),
    (19, (256, 512, 1, 1, 0)), (21, (512, 256, 3, 1, 1)),
    (23, (256, 512, 1, 1, 0)), (25, (512, 256, 3, 1, 1)),
    (27, (256, 512, 1, 1, 0)), (29, (512, 256, 3, 1, 1)),
    (31, (512, 512, 1, 1, 0)), (33, (1024, 512, 3, 1, 1)), "M",
    (36, (512, 1024, 1, 1, 0)), (38, (1024, 512, 3, 1, 1)),
    (40, (512, 1024, 1, 1, 0)), (42, (1024, 512, 3, 1, 1)),
    (44, (1024, 1024, 3, 1, 1)), (46, (1024, 1024, 3, 2, 1)),
    (48, (1024, 1024, 3, 1, 1)), (50, (1024, 1024, 3, 1, 1)),
]


def synth_uniform(shape, tag: int, bound: float, seed: int = 1234) -> np.ndarray:
    """fp32 array ~ U(-bound, bound) from a stream keyed by (seed, tag)."""
    rng = np.random.Generator(np.random.PCG64([seed, tag]))
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def synth_normal(shape, tag: int, std: float = 1.0, seed: int = 1234) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([seed, tag]))
    return (rng.standard_normal(size=shape) * std).astype(np.float32)


def yolov1_state_dict(S: int = 7, B: int = 2, C: int = 20, seed: int = 1234, gain: float = 1.45) -> dict:
    """Deterministic weights for YOLOv1(YOLOv1Backbone + FC head), keyed like the reference's
    state_dict.  Bound = gain*sqrt(3/fan_in) keeps activations O(1) through 24
    LeakyReLU layers so a wrong layer shows up in the output instead of vanishing."""
    sd = {}
    for item in YOLOV1_BACKBONE_CFG:
        if item == "M":
            continue
        idx, (co, ci, k, _s, _p) = item
        fan_in = ci * k * k
        bound = gain * (3.0 / fan_in) ** 0.5
        sd[f"backbone.features.{idx}.weight"] = synth_uniform((co, ci, k, k), 2 * idx, bound, seed)
        sd[f"backbone.features.{idx}.bias"] = synth_uniform((co,), 2 * idx + 1, 0.1, seed)
    D = B * 5 + C
    k1 = 1024 * S * S
    sd["head.1.weight"] = synth_uniform((4096, k1), 1001, gain * (3.0 / k1) ** 0.5, seed)
    sd["head.1.bias"] = synth_uniform((4096,), 1002, 0.1, seed)
    sd["head.4.weight"] = synth_uniform((S * S * D, 4096), 1003, (3.0 / 4096) ** 0.5, seed)
    sd["head.4.bias"] = synth_uniform((S * S * D,), 1004, 0.1, seed)
    return sd


def synth_targets(N: int, seed: int, S: int = 7, B: int = 2, C: int = 20, max_obj: int = 3) -> np.ndarray:
    """(N,S,S,5B+C) targets following the reference's encoding rule (slot 0 only, one-hot class,
    first object wins a cell; reference: src/yolo/dataset.py:487-532), 0..max_obj objects/image."""
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    t = np.zeros((N, S, S, 5 * B + C), dtype=np.float32)
    for n in range(N):
        for _ in range(int(rng.integers(0, max_obj + 1))):
            xc, yc = rng.uniform(0, 1, 2)
            w, h = rng.uniform(0.05, 0.9, 2)
            cls = int(rng.integers(0, C))
            i = min(int(S * yc), S - 1)
            j = min(int(S * xc), S - 1)
            if t[n, i, j, 4] == 0:
                t[n, i, j, 0] = S * xc - j
                t[n, i, j, 1] = S * yc - i
                t[n, i, j, 2] = w
                t[n, i, j, 3] = h
                t[n, i, j, 4] = 1.0
                t[n, i, j, 5 * B + cls] = 1.0
    return t


def synth_images(N: int, seed: int = 0, hw: int = 448) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([seed, 5]))
    return rng.standard_normal(size=(N, 3, hw, hw), dtype=np.float32)
