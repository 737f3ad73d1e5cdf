"""ctypes/numpy front-end of oracle/yolo_oracle.c (TEST INFRASTRUCTURE, see that file's header).

Each wrapper cites the reference lines its C function restates.  The library is built by
``make -C oracle`` (also done by ``__graft_entry__.build()``); it is loaded lazily so that
importing this module never compiles anything.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "yolo_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            so = build()
        L = ctypes.CDLL(so)
        L.oracle_loss_fwd_bwd.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, _f32p, _f32p]
        L.oracle_loss_fwd_bwd.restype = ctypes.c_int
        L.oracle_loss_iou.argtypes = [_f32p, _f32p, ctypes.c_long, _f32p]
        L.oracle_decode.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _f64p]
        L.oracle_decode.restype = ctypes.c_int
        L.oracle_decode_gt.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f64p]
        L.oracle_decode_gt.restype = ctypes.c_int
        L.oracle_iou.argtypes = [_f64p, _f64p, ctypes.c_int]
        L.oracle_iou.restype = ctypes.c_double
        L.oracle_nms.argtypes = [_f64p, ctypes.c_int, ctypes.c_double, ctypes.c_int, _i32p]
        L.oracle_nms.restype = ctypes.c_int
        L.oracle_conv2d.argtypes = [_f32p, _f32p, _f32p, _f32p] + [ctypes.c_int] * 8 + [ctypes.c_float]
        L.oracle_maxpool2.argtypes = [_f32p, _f32p] + [ctypes.c_int] * 4
        L.oracle_linear.argtypes = [_f32p, _f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


INFERENCE, METRICS = 0, 1


def loss_fwd_bwd(pred, tgt, S=7, B=2, C=20, lambda_coord=5.0, lambda_noobj=0.5, want_grad=True):
    """src/yolo/loss.py:55-212 -> (out5 fp32[5] = total,coord,conf_obj,conf_noobj,class ; dpred or None)."""
    pred, tgt = _f32(pred), _f32(tgt)
    N = pred.shape[0]
    out5 = np.zeros(5, np.float32)
    dpred = np.zeros_like(pred) if want_grad else None
    rc = lib().oracle_loss_fwd_bwd(_p(pred, _f32p), _p(tgt, _f32p), N, S, B, C, lambda_coord, lambda_noobj,
                                   _p(out5, _f32p), _p(dpred, _f32p) if want_grad else None)
    if rc != 0:
        raise RuntimeError("index out of bounds: a target cell selects box slot >= B (reference gather raises here)")
    return out5, dpred


def loss_iou(b1, b2):
    """src/yolo/loss.py:174-212 (broadcast done by the caller): (...,4),(...,4) -> (...)"""
    b1, b2 = np.broadcast_arrays(_f32(b1), _f32(b2))
    b1, b2 = _f32(b1), _f32(b2)
    out = np.zeros(b1.shape[:-1], np.float32)
    lib().oracle_loss_iou(_p(b1, _f32p), _p(b2, _f32p), out.size, _p(out, _f32p))
    return out


def decode(pred, conf_thr, S=7, B=2, C=20):
    """src/yolo/inference.py:170-210 == src/yolo/metrics.py:185-218 -> (n,6) f64 [cls,conf,x,y,w,h]."""
    pred = _f32(pred)
    rec = np.zeros((S * S * B, 6), np.float64)
    n = lib().oracle_decode(_p(pred, _f32p), S, B, C, float(conf_thr), _p(rec, _f64p))
    return rec[:n].copy()


def decode_gt(tgt, S=7, B=2, C=20):
    """src/yolo/metrics.py:232-256 -> (n,5) f64 [cls,x,y,w,h]."""
    tgt = _f32(tgt)
    rec = np.zeros((S * S, 5), np.float64)
    n = lib().oracle_decode_gt(_p(tgt, _f32p), S, B, C, _p(rec, _f64p))
    return rec[:n].copy()


def iou(a, b, variant):
    """variant INFERENCE: inference.py:212-249 + schemas.py:18-55 ; METRICS: metrics.py:298-341."""
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return float(lib().oracle_iou(_p(a, _f64p), _p(b, _f64p), variant))


def nms(rec, thr, variant):
    """variant INFERENCE: inference.py:298-317 ; METRICS: metrics.py:270-296 -> kept indices (output order)."""
    rec = np.ascontiguousarray(rec, np.float64).reshape(-1, 6)
    keep = np.zeros(max(len(rec), 1), np.int32)
    n = lib().oracle_nms(_p(rec, _f64p), len(rec), float(thr), variant, _p(keep, _i32p))
    return keep[:n].copy()


def conv2d(x, w, b, stride, pad, slope=1.0):
    """nn.Conv2d(+LeakyReLU(slope)) NCHW/OIHW (models.py:47-84 hyper-parameters)."""
    x, w = _f32(x), _f32(w)
    N, Ci, H, W = x.shape
    Co, _, K, _ = w.shape
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    y = np.zeros((N, Co, Ho, Wo), np.float32)
    bb = _f32(b) if b is not None else None
    lib().oracle_conv2d(_p(x, _f32p), _p(w, _f32p), _p(bb, _f32p) if bb is not None else None, _p(y, _f32p),
                        N, Ci, H, W, Co, K, stride, pad, slope)
    return y


def maxpool2(x):
    x = _f32(x)
    N, C, H, W = x.shape
    y = np.zeros((N, C, H // 2, W // 2), np.float32)
    lib().oracle_maxpool2(_p(x, _f32p), _p(y, _f32p), N, C, H, W)
    return y


def linear(x, w, b, slope=1.0):
    x, w = _f32(x), _f32(w)
    N, K = x.shape
    O = w.shape[0]
    y = np.zeros((N, O), np.float32)
    bb = _f32(b) if b is not None else None
    lib().oracle_linear(_p(x, _f32p), _p(w, _f32p), _p(bb, _f32p) if bb is not None else None, _p(y, _f32p), N, K, O, slope)
    return y
