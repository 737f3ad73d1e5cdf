"""Host (numpy, float64) decode / IoU / NMS used when the tensors live on the CPU.

This is the reference's ``--device cpu`` behaviour for post-processing (Python-float arithmetic on
fp32 values widened by ``.item()``: src/yolo/inference.py:170-317, src/yolo/metrics.py:185-341),
written array-wise.  It is selected by the tensor's device, never as a substitute for a missing HIP
library: device tensors always go through yolo_decode / yolo_nms (postprocess.hip).
"""

from __future__ import annotations

import numpy as np

INFERENCE, METRICS = 0, 1


def decode(pred: np.ndarray, conf_thr: float, S: int, B: int) -> np.ndarray:
    """(S,S,5B+C) fp32 -> (n,6) float64 rows [class_id, conf*prob, x, y, w, h] in (i,j,b) scan order."""
    p = np.asarray(pred, dtype=np.float32).reshape(S, S, -1)
    cls = p[..., B * 5:].argmax(-1)                              # first maximum, like torch.argmax
    prob = np.take_along_axis(p[..., B * 5:], cls[..., None], -1)[..., 0].astype(np.float64)
    boxes = p[..., : B * 5].reshape(S, S, B, 5).astype(np.float64)
    jj, ii = np.meshgrid(np.arange(S, dtype=np.float64), np.arange(S, dtype=np.float64))
    rec = np.empty((S, S, B, 6), np.float64)
    rec[..., 0] = cls[..., None]
    rec[..., 1] = boxes[..., 4] * prob[..., None]
    rec[..., 2] = (jj[..., None] + boxes[..., 0]) / S
    rec[..., 3] = (ii[..., None] + boxes[..., 1]) / S
    rec[..., 4] = boxes[..., 2]
    rec[..., 5] = boxes[..., 3]
    rec = rec.reshape(-1, 6)
    return rec[rec[:, 1] > conf_thr]


def decode_gt(tgt: np.ndarray, S: int, B: int) -> np.ndarray:
    """(S,S,5B+C) fp32 -> (n,5) float64 rows [class_id, x, y, w, h]; a cell holds an object iff conf0 > 0."""
    t = np.asarray(tgt, dtype=np.float32).reshape(S, S, -1)
    jj, ii = np.meshgrid(np.arange(S, dtype=np.float64), np.arange(S, dtype=np.float64))
    rec = np.empty((S, S, 5), np.float64)
    rec[..., 0] = t[..., B * 5:].argmax(-1)
    rec[..., 1] = (jj + t[..., 0].astype(np.float64)) / S
    rec[..., 2] = (ii + t[..., 1].astype(np.float64)) / S
    rec[..., 3] = t[..., 2]
    rec[..., 4] = t[..., 3]
    return rec[t[..., 4] > 0]


def iou_one_to_many(a: np.ndarray, b: np.ndarray, variant: int) -> np.ndarray:
    """IoU of box a (4,) against boxes b (m,4), centre format, float64, a is the first argument."""
    ax1, ay1, ax2, ay2 = a[0] - a[2] / 2, a[1] - a[3] / 2, a[0] + a[2] / 2, a[1] + a[3] / 2
    bx1, by1 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2
    bx2, by2 = b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    dw = np.where(bx2 < ax2, bx2, ax2) - np.where(bx1 > ax1, bx1, ax1)
    dh = np.where(by2 < ay2, by2, ay2) - np.where(by1 > ay1, by1, ay1)
    inter = np.where(dw > 0, dw, 0.0) * np.where(dh > 0, dh, 0.0)
    a1, a2 = a[2] * a[3], b[:, 2] * b[:, 3]
    if variant == INFERENCE:
        return inter / (a1 + a2 - inter + 1e-6)
    union = a1 + a2 - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union == 0, 0.0, inter / np.where(union == 0, 1.0, union))


def iou_scalar(a, b, variant: int) -> float:
    return float(iou_one_to_many(np.asarray(a, np.float64), np.asarray(b, np.float64).reshape(1, 4), variant)[0])


def nms(rec: np.ndarray, thr: float, variant: int) -> np.ndarray:
    """Greedy NMS over (n,6) records -> kept indices in the reference's output order
    (variant INFERENCE: confidence order; METRICS: grouped by class in first-appearance order)."""
    n = len(rec)
    if n == 0:
        return np.zeros(0, np.int32)
    order = np.argsort(-rec[:, 1], kind="stable")            # ties keep scan order
    cls, box = rec[order, 0], rec[order, 2:6]
    alive = np.ones(n, bool)
    kept = []
    for a in range(n):
        if not alive[a]:
            continue
        kept.append(a)
        later = np.nonzero(alive[a + 1:] & (cls[a + 1:] == cls[a]))[0] + a + 1
        if len(later):
            alive[later[~(iou_one_to_many(box[a], box[later], variant) < thr)]] = False
    kept = np.asarray(kept, np.int64)
    if variant == METRICS:
        first = {}
        for pos, c in enumerate(cls):
            first.setdefault(c, pos)
        kept = kept[np.argsort([first[cls[k]] for k in kept], kind="stable")]
    return order[kept].astype(np.int32)
