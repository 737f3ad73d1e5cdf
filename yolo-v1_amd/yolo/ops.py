"""Tensor-level wrappers over the C ABI (include/yolo_hip.h).  Device tensors in, device tensors out;
nothing here synchronises except the explicit ``*_host`` helpers, which do ONE device->host copy
where the reference does hundreds of ``.item()`` syncs (src/yolo/inference.py:184-191,
src/yolo/metrics.py:200-208, src/yolo/loss.py:165-169)."""

from __future__ import annotations

import ctypes

import torch

from . import _hip
from ._hip import check, lib, ptr, stream

# ------------------------------------------------------------------------------------------------
# decode / IoU / NMS
# ------------------------------------------------------------------------------------------------


def _as_f32_dev(t: torch.Tensor) -> torch.Tensor:
    _hip.require_cuda(t)
    return t.detach().to(torch.float32).contiguous()


@_hip.device_guard
def decode(pred: torch.Tensor, conf_thr: float, S: int, B: int, C: int):
    """(N,S,S,5B+C) fp32 -> rec (N,S*S*B,6) f64 {cls,conf,x,y,w,h}, counts (N,) i32.
    Replaces src/yolo/inference.py:170-210 / src/yolo/metrics.py:185-218."""
    pred = _as_f32_dev(pred)
    N = pred.shape[0]
    rec = torch.empty((N, S * S * B, 6), dtype=torch.float64, device=pred.device)
    counts = torch.empty((N,), dtype=torch.int32, device=pred.device)
    check(lib().yolo_decode(ptr(pred), N, S, B, C, float(conf_thr), ptr(rec), ptr(counts), stream()), "yolo_decode")
    return rec, counts


@_hip.device_guard
def decode_gt(tgt: torch.Tensor, S: int, B: int, C: int):
    """(N,S,S,5B+C) fp32 -> rec (N,S*S,5) f64 {cls,x,y,w,h}, counts.  Replaces src/yolo/metrics.py:232-256."""
    tgt = _as_f32_dev(tgt)
    N = tgt.shape[0]
    rec = torch.empty((N, S * S, 5), dtype=torch.float64, device=tgt.device)
    counts = torch.empty((N,), dtype=torch.int32, device=tgt.device)
    check(lib().yolo_decode_gt(ptr(tgt), N, S, B, C, ptr(rec), ptr(counts), stream()), "yolo_decode_gt")
    return rec, counts


@_hip.device_guard
def nms(rec: torch.Tensor, counts: torch.Tensor, thr: float, variant: int):
    """rec (N,M,6) f64 + counts -> keep (N,M) i32 (reference output order), keep_counts (N,).
    Replaces src/yolo/inference.py:298-317 (variant 0) / src/yolo/metrics.py:270-296 (variant 1)."""
    _hip.require_cuda(rec, counts)
    assert rec.dtype == torch.float64 and rec.is_contiguous() and counts.dtype == torch.int32
    N, M, _ = rec.shape
    keep = torch.empty((N, M), dtype=torch.int32, device=rec.device)
    kc = torch.empty((N,), dtype=torch.int32, device=rec.device)
    check(lib().yolo_nms(ptr(rec), ptr(counts), N, M, float(thr), int(variant), ptr(keep), ptr(kc), stream()), "yolo_nms")
    return keep, kc


@_hip.device_guard
def pairwise_iou(a: torch.Tensor, b: torch.Tensor, variant: int) -> torch.Tensor:
    """a (na,4), b (nb,4) f64 (x,y,w,h) -> (na,nb) f64.  src/yolo/inference.py:229-249 / metrics.py:313-341."""
    _hip.require_cuda(a, b)
    a = a.to(torch.float64).contiguous()
    b = b.to(torch.float64).contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=a.device)
    check(lib().yolo_pairwise_iou(ptr(a), a.shape[0], ptr(b), b.shape[0], int(variant), ptr(out), stream()), "yolo_pairwise_iou")
    return out


@_hip.device_guard
def map_match(rec: torch.Tensor, keep: torch.Tensor, kc: torch.Tensor, grec: torch.Tensor, gcnt: torch.Tensor, thresholds, extra_thr: float,
              small_area: float, medium_area: float):
    """TP bits of every kept prediction (see yolo_map_match) + the size bucket of every ground truth, on the device.
    Replaces the matching loops of src/yolo/metrics.py:343-442, 444-491, 568-651."""
    import ctypes
    _hip.require_cuda(rec, keep, kc, grec, gcnt)
    N, M, _ = rec.shape
    G = grec.shape[1]
    tp = torch.zeros((N, M), dtype=torch.int64, device=rec.device)
    bucket = torch.zeros((N, G), dtype=torch.int32, device=rec.device)
    thr = (ctypes.c_double * len(thresholds))(*[float(t) for t in thresholds])
    check(lib().yolo_map_match(ptr(rec), ptr(keep), ptr(kc), N, M, ptr(grec), ptr(gcnt), G, thr, len(thresholds), float(extra_thr), float(small_area),
                               float(medium_area), ptr(tp), ptr(bucket), stream()), "yolo_map_match")
    return tp, bucket


def postprocess_host(pred: torch.Tensor, conf_thr: float, nms_thr: float, variant: int, S: int, B: int, C: int):
    """decode + NMS on the device for a batch, ONE device->host copy.
    Returns per image (rec[n] as a (cnt,6) float64 numpy array, kept indices int32 numpy array)."""
    rec, counts = decode(pred, conf_thr, S, B, C)
    keep, kc = nms(rec, counts, nms_thr, variant)
    rec_h, counts_h, keep_h, kc_h = rec.cpu().numpy(), counts.cpu().numpy(), keep.cpu().numpy(), kc.cpu().numpy()
    return [(rec_h[n, : counts_h[n]], keep_h[n, : kc_h[n]]) for n in range(rec_h.shape[0])]


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------


@_hip.device_guard
def loss_fwd_bwd(pred: torch.Tensor, tgt: torch.Tensor, S: int, B: int, C: int, lambda_coord: float, lambda_noobj: float,
                 want_grad: bool = True):
    """Fused loss forward+backward (src/yolo/loss.py:87-212).  Returns (out8 fp32 device tensor, dpred or None)."""
    pred = _as_f32_dev(pred)
    tgt = _as_f32_dev(tgt)
    N = pred.shape[0]
    if pred.dim() != 4 or tuple(pred.shape) != (N, S, S, 5 * B + C) or tgt.shape != pred.shape:
        # the kernel cannot know the buffers' extents: e.g. YOLOLoss(S=14) on S=7 predictions would read out of bounds
        raise RuntimeError(f"YOLOLoss: predictions {tuple(pred.shape)} / targets {tuple(tgt.shape)} must both be (N, {S}, {S}, {5 * B + C})")
    out = torch.empty((8,), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    work = torch.empty((N * 8,), dtype=torch.float64, device=pred.device)
    check(lib().yolo_loss_fwd_bwd(ptr(pred), ptr(tgt), N, S, B, C, float(lambda_coord), float(lambda_noobj),
                                  ptr(out), ptr(dpred), ptr(work), stream()), "yolo_loss_fwd_bwd")
    return out, dpred


@_hip.device_guard
def loss_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """YOLOLoss.compute_iou (src/yolo/loss.py:174-212) with broadcasting done here."""
    b1, b2 = torch.broadcast_tensors(b1, b2)
    b1 = _as_f32_dev(b1)
    b2 = _as_f32_dev(b2)
    out = torch.empty(b1.shape[:-1], dtype=torch.float32, device=b1.device)
    check(lib().yolo_loss_iou(ptr(b1), ptr(b2), out.numel(), ptr(out), stream()), "yolo_loss_iou")
    return out
