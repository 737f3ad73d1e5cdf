"""Image preprocessing on the device (libyolo_hip.so: yolo_preprocess_u8) -- SURVEY.md 8(f)-1.

The reference's transform (src/yolo/inference.py:58-66, src/yolo/dataset.py:224-233) is
``Resize((448, 448)) -> ToTensor() -> Normalize(ImageNet)`` on the host, one PIL image at a time, followed by a
2.4 MB fp32 host->device copy per image.  Here the host only decodes the file; the uint8 pixels (3 B each) go to the
device and one or two kernel launches produce either the fp32 NCHW tensor the reference's transform returns or,
directly, the zero-haloed NHWC4 bf16 buffer the stem convolution reads.

``Resize`` on a PIL image is ``PIL.Image.resize(size, BILINEAR)``; the tables below are Pillow's own
(libImaging/Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc``: triangle filter, support widened by the
down-scaling factor, 22-bit fixed point) and the kernels repeat its two 8-bit passes, so the result is bit-identical
to the host path (tests/test_gpu_preprocess.py; the table code alone vs Pillow in tests/test_preprocess_cpu.py).
"""

from __future__ import annotations

import ctypes
import math
from functools import lru_cache

import numpy as np
import torch

from ._hip import check, lib, ptr, require_cuda, stream
from ._hip import device_guard as _hip_device_guard

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=64)
def bilinear_tables(in_size: int, out_size: int):
    """Pillow's BILINEAR resampling tables for one axis: (bounds int32 [out][2], coeffs int32 [out][ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale                       # bilinear: support 1.0 x filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / filterscale
    one = float(1 << _PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        x0 = max(int(center - support + 0.5), 0)
        x1 = min(int(center + support + 0.5), in_size)
        n = x1 - x0
        w = [max(0.0, 1.0 - abs((x + x0 - center + 0.5) * inv)) for x in range(n)]
        tot = sum(w)
        if tot != 0.0:
            w = [v / tot for v in w]
        bounds[xx] = (x0, n)
        coef[xx, :n] = [int(0.5 + v * one) for v in w]     # weights are >= 0 for this filter; int() truncates like the C cast
    return bounds, coef, ksize


@lru_cache(maxsize=64)
def _device_tables(in_size: int, out_size: int, device_index: int):
    b, c, k = bilinear_tables(in_size, out_size)
    dev = torch.device("cuda", device_index)
    return torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), k


def preprocess_u8_into(images: torch.Tensor, size, act, mean=MEAN, std=STD) -> None:
    """resize + normalise ``images`` (uint8 [N][H][W][3] on the device) into an existing NHWC4 activation buffer."""
    _run(images, size, mean, std, None, act)


@_hip_device_guard
def preprocess_u8(images: torch.Tensor, size=(448, 448), mean=MEAN, std=STD, nchw: bool = True, nhwc4_halo: int | None = None):
    """images: uint8 device tensor [N][H][W][3] (RGB, as decoded).  Returns (fp32 [N][3][h][w] | None, Act | None):
    the reference transform's result and / or the stem-ready NHWC4 bf16 activation (``nhwc4_halo`` = 3 for the stem)."""
    require_cuda(images)
    out = torch.empty((images.shape[0], 3, size[0], size[1]), dtype=torch.float32, device=images.device) if nchw else None
    act = None
    if nhwc4_halo is not None:
        from .engine import Act
        act = Act(images.shape[0], size[0], size[1], 4, nhwc4_halo, images.device)
    _run(images, size, mean, std, out, act)
    return out, act


def _run(images, size, mean, std, out, act):
    require_cuda(images)
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise ValueError("preprocess_u8 expects a uint8 tensor of shape (N, H, W, 3)")
    images = images.contiguous()
    N, Hs, Ws, _ = images.shape
    Ho, Wo = size
    dev = images.device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    hb = hc = vb = vc = tmp = None
    hk = vk = 0
    if Ws != Wo:
        hb, hc, hk = _device_tables(Ws, Wo, idx)
        tmp = torch.empty((N, Hs, Wo, 3), dtype=torch.uint8, device=dev)
    if Hs != Ho:
        vb, vc, vk = _device_tables(Hs, Ho, idx)
    if act is not None and (act.N, act.H, act.W, act.C) != (N, Ho, Wo, 4):
        raise ValueError("activation buffer does not match the batch / target size")
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*std)
    check(lib().yolo_preprocess_u8(ptr(images), N, Hs, Ws, Ho, Wo, ptr(hb), ptr(hc), hk, ptr(vb), ptr(vc), vk, ptr(tmp), m3, s3,
                                   act.p if act is not None else None, act.halo if act is not None else 0, ptr(out), stream()), "yolo_preprocess_u8")
