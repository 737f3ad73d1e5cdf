"""TEST INFRASTRUCTURE (oracle): NumPy restatement of the reference's image preprocessing
(src/yolo/inference.py:58-66, src/yolo/dataset.py:224-233):

    Resize((448, 448))  ->  ToTensor()  ->  Normalize(mean, std)

torchvision (un-vendored dependency of the reference, absent here) implements Resize on PIL inputs as
``PIL.Image.resize(size, BILINEAR)``; Pillow IS installed, so the restatement below -- Pillow's two-pass 8-bit
resampling (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc:
22-bit fixed-point triangle-filter coefficients, support widened by the down-scaling factor, uint8 intermediate) --
is pinned bit for bit against ``Image.resize`` itself in tests/test_preprocess_cpu.py.  ToTensor = uint8 -> fp32 / 255,
Normalize = (x - mean) / std in fp32, both restated with NumPy fp32 arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def bilinear_coeffs(in_size: int, out_size: int):
    """(bounds int32 [out][2] = (first input index, count), coeffs int32 [out][ksize]) of Pillow's BILINEAR filter."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.abs((np.arange(xmax) + xmin - center + 0.5) * ss)
        w = np.where(w < 1.0, 1.0 - w, 0.0)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS))
    return bounds, np.trunc(fixed).astype(np.int32)          # the C (int) cast truncates toward zero


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img uint8 [H][W][C] -> uint8 [out_h][out_w][C], bit-identical to PIL.Image.resize((out_w, out_h), BILINEAR)."""
    h, w, _ = img.shape
    cur = img
    if out_w != w:
        bounds, kk = bilinear_coeffs(w, out_w)
        out = np.empty((h, out_w, img.shape[2]), dtype=np.uint8)
        src = cur.astype(np.int64)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1)
            out[:, xx, :] = _clip8(acc)
        cur = out
    if out_h != h:
        bounds, kk = bilinear_coeffs(h, out_h)
        out = np.empty((out_h, cur.shape[1], img.shape[2]), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = (1 << (PRECISION_BITS - 1)) + (src[y0:y0 + n, :, :] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0)
            out[yy] = _clip8(acc)
        cur = out
    return cur if cur is not img else img.copy()


def to_tensor_normalize(img_u8: np.ndarray, mean=MEAN, std=STD) -> np.ndarray:
    """uint8 [H][W][3] -> fp32 [3][H][W]: ToTensor (/255) then Normalize, every step in fp32."""
    t = img_u8.astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, dtype=np.float32)
    s = np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(((t - m) / s).transpose(2, 0, 1))


def preprocess(img_u8: np.ndarray, size=(448, 448)) -> np.ndarray:
    return to_tensor_normalize(resize_bilinear_u8(img_u8, size[0], size[1]))
