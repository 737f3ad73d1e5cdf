"""oracle/post_py.py -- TEST INFRASTRUCTURE (imported only by tests/ and bench.py's cpu_baseline leg).

Pure-Python restatement of the reference's metrics-side post-processing as the reference itself executes it on the host:
one ``.item()`` per scalar of a torch tensor, Python floats (= doubles) afterwards, Python lists for the greedy NMS.

  decode_py   mAPMetric._parse_predictions   src/yolo/metrics.py:185-218  (cell loop i, j; box loop b; first-max class)
  iou_py      mAPMetric._calculate_iou       src/yolo/metrics.py:296-341  (union == 0 -> 0.0, no epsilon)
  nms_py      mAPMetric._apply_nms           src/yolo/metrics.py:258-294  (stable descending sort, class buckets in
                                                                          first-appearance order, keep while IoU < thr)

It exists for two reasons: SURVEY.md 8(d) asks for the reference's Python path timed beside the C restatement, and it pins
oracle/yolo_oracle.c's decode / NMS from a second, independent angle (tests/test_oracle_golden.py compares the two and the
reference-run fixtures).  Output format = oracle.decode / oracle.nms: records [class_id, conf, x, y, w, h] (float64) in scan order,
kept indices into those records in the reference's output order.
"""

from __future__ import annotations

import numpy as np
import torch


def decode_py(pred: torch.Tensor, conf_thr: float, S: int = 7, B: int = 2) -> np.ndarray:
    rows = []
    for i in range(S):
        for j in range(S):
            cell = pred[i, j]
            probs = cell[B * 5:]
            for b in range(B):
                x, y, w, h, c = (cell[b * 5 + k].item() for k in range(5))
                best = int(torch.argmax(probs).item())
                score = c * probs[best].item()
                if score > conf_thr:
                    rows.append((float(best), score, (j + x) / S, (i + y) / S, w, h))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 6)


def iou_py(a, b) -> float:
    ax0, ay0, ax1, ay1 = a[0] - a[2] / 2, a[1] - a[3] / 2, a[0] + a[2] / 2, a[1] + a[3] / 2
    bx0, by0, bx1, by1 = b[0] - b[2] / 2, b[1] - b[3] / 2, b[0] + b[2] / 2, b[1] + b[3] / 2
    inter = max(0, min(ax1, bx1) - max(ax0, bx0)) * max(0, min(ay1, by1) - max(ay0, by0))
    union = a[2] * a[3] + b[2] * b[3] - inter
    return 0.0 if union == 0 else inter / union


def nms_py(rec: np.ndarray, thr: float) -> np.ndarray:
    dets = [(int(r[0]), float(r[1]), tuple(float(v) for v in r[2:6]), n) for n, r in enumerate(rec)]
    dets.sort(key=lambda d: d[1], reverse=True)          # stable: equal confidences keep scan order
    buckets: dict[int, list] = {}
    for d in dets:
        buckets.setdefault(d[0], []).append(d)
    kept = []
    for members in buckets.values():                      # dict order = first appearance in the sorted list
        while members:
            top = members.pop(0)
            kept.append(top[3])
            members[:] = [d for d in members if iou_py(top[2], d[2]) < thr]
    return np.asarray(kept, dtype=np.int32)
