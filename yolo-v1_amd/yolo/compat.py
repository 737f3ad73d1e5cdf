"""Functional spellings used by BASELINE.json's north_star -- ``cellboxes_to_boxes``, ``non_max_suppression``,
``mean_average_precision`` (and ``YoloLoss`` in ``yolo/__init__.py``).  The reference has no functions of these names
(SURVEY.md 0.2): its decode is ``YOLOInference.parse_predictions`` / ``mAPMetric._parse_predictions``, its NMS
``YOLOInference.non_max_suppression`` / ``mAPMetric._apply_nms`` and its mAP the class ``mAPMetric``.  These thin
wrappers expose the same arithmetic -- the HIP kernels for device tensors, the host restatement for CPU tensors -- under
the names the north_star uses, so that code written against either spelling runs.
"""

from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import _post_cpu
from .metrics import mAPMetric


def cellboxes_to_boxes(predictions: torch.Tensor, conf_threshold: float = 0.0, S: int = 7, B: int = 2, C: int = 20) -> List[np.ndarray]:
    """(N, S, S, 5B+C) raw predictions -> per image a float64 array (k, 6) of [class_id, confidence, x, y, w, h]
    (image-relative centre format) for the boxes with confidence * class-probability > conf_threshold, in (row, col, box)
    scan order -- reference src/yolo/inference.py:170-210."""
    if predictions.dim() == 3:
        predictions = predictions.unsqueeze(0)
    if predictions.is_cuda:
        from . import ops
        rec, cnt = ops.decode(predictions, conf_threshold, S, B, C)
        rec, cnt = rec.cpu().numpy(), cnt.cpu().numpy()
        return [rec[n, : cnt[n]] for n in range(rec.shape[0])]
    return [_post_cpu.decode(p.detach().numpy(), conf_threshold, S, B) for p in predictions]


def non_max_suppression(bboxes, iou_threshold: float = 0.4, class_agnostic_output: bool = True) -> np.ndarray:
    """[class_id, confidence, x, y, w, h] rows -> the rows that survive greedy per-class NMS (a box is dropped when its IoU
    with a kept box of the same class is >= iou_threshold), in global confidence order (``class_agnostic_output``, the
    reference's inference variant, inference.py:251-317) or grouped by class (its metrics variant, metrics.py:258-296)."""
    rec = np.asarray(bboxes, dtype=np.float64).reshape(-1, 6)
    if len(rec) == 0:
        return rec
    variant = _post_cpu.INFERENCE if class_agnostic_output else _post_cpu.METRICS
    return rec[_post_cpu.nms(rec, iou_threshold, variant)]


def mean_average_precision(predictions: torch.Tensor, targets: torch.Tensor, num_classes: int = 20, iou_thresholds=None, conf_threshold: float = 0.01,
                           nms_threshold: float = 0.4, S: int = 7, B: int = 2) -> Dict[str, float]:
    """One-shot ``mAPMetric``: (N, S, S, 5B+C) predictions and targets -> the metric dictionary (metrics.py:78-171)."""
    m = mAPMetric(num_classes=num_classes, iou_thresholds=iou_thresholds, conf_threshold=conf_threshold, nms_threshold=nms_threshold, S=S, B=B)
    m.update(predictions, targets)
    return m.compute()
