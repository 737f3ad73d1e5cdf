"""mAP evaluation with the reference's surface (src/yolo/metrics.py:17-714).

``update`` is the hot part: the reference decodes and NMS-filters every image in Python with a
device sync per scalar (metrics.py:187-218, 270-296).  Here a batch on a ROCm device is decoded,
NMS-filtered (metrics variant: IoU without epsilon, output grouped by class) and its ground truth
parsed by three kernel launches and one device->host copy.  The AP arithmetic stays on the host in
float64 NumPy, restated from the reference's protocol: VOC-07 11-point interpolation, greedy
matching in confidence order, classes without ground truth or predictions contribute AP 0.

One deliberate restructuring: the best-IoU ground truth of a prediction does not depend on the IoU
threshold, so it is found once per class instead of once per (class, threshold).

When every ``update`` ran on a ROCm device, the TP / FP matching itself (all classes, the 10 thresholds, the three
size buckets and the overall precision / recall) is also done there, per image, by ``yolo_map_match`` -- a prediction
only ever meets the ground truths of its own image, and an image's kept list is the global confidence order restricted
to that image -- and ``compute`` reduces to sorts, cumulative sums and the 11-point interpolation in vectorised NumPy
(same arithmetic, same floats as the host path, which remains for lists filled by hand or on the CPU).
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from . import _post_cpu

EPSILON = 1e-6

Box = Tuple[float, float, float, float]
Pred = Tuple[int, float, Box]
GT = Tuple[int, Box]


def _records_to_preds(rec: np.ndarray) -> List[Pred]:
    return [(int(c), conf, (x, y, w, h)) for c, conf, x, y, w, h in rec.tolist()]


def _records_to_gts(rec: np.ndarray) -> List[GT]:
    return [(int(c), (x, y, w, h)) for c, x, y, w, h in rec.tolist()]


class mAPMetric:
    """PASCAL-VOC style mAP at one or several IoU thresholds (default 0.50:0.05:0.95)."""

    def __init__(self, num_classes: int, iou_thresholds: List[float] = None, conf_threshold: float = 0.01,
                 nms_threshold: float = 0.4, S: int = 7, B: int = 2):
        self.num_classes = num_classes
        if iou_thresholds is None:
            self.iou_thresholds = [0.5 + 0.05 * i for i in range(10)]
        elif isinstance(iou_thresholds, (int, float)):
            self.iou_thresholds = [float(iou_thresholds)]
        else:
            self.iou_thresholds = list(iou_thresholds)
        self.conf_threshold = conf_threshold
        self.nms_threshold = nms_threshold
        self.S = S
        self.B = B
        self.all_predictions: List[List[Pred]] = []
        self.all_ground_truths: List[List[GT]] = []
        self._dev: List[tuple] = []      # per device batch: (cls, conf, tp_bits) of the kept predictions, (cls, bucket) of the GTs
        self._dev_images = 0

    def reset(self):
        self.all_predictions = []
        self.all_ground_truths = []
        self._dev = []
        self._dev_images = 0

    # ------------------------------------------------------------------ accumulation
    def update(self, predictions: torch.Tensor, targets: torch.Tensor):
        """predictions, targets: (batch, S, S, 5B+C)."""
        if predictions.is_cuda:
            from . import ops
            C = predictions.shape[-1] - 5 * self.B
            rec, counts = ops.decode(predictions, self.conf_threshold, self.S, self.B, C)
            keep, kc = ops.nms(rec, counts, self.nms_threshold, ops._hip.NMS_METRICS)
            grec, gcnt = ops.decode_gt(targets.to(predictions.device), self.S, self.B, C)
            tp = bucket = None
            T = len(self.iou_thresholds)
            if T <= 15 and rec.shape[1] <= 128 and grec.shape[1] <= 64 and self._dev_images == len(self.all_predictions):
                tp, bucket = ops.map_match(rec, keep, kc, grec, gcnt, self.iou_thresholds, 0.5, (32 / 448) ** 2, (96 / 448) ** 2)
            rec_h, keep_h, kc_h = rec.cpu().numpy(), keep.cpu().numpy(), kc.cpu().numpy()      # (the batch's only device->host copies)
            grec, gcnt = grec.cpu().numpy(), gcnt.cpu().numpy()
            kept = [rec_h[n][keep_h[n, : kc_h[n]]] for n in range(rec_h.shape[0])]
            for n, k in enumerate(kept):
                self.all_predictions.append(_records_to_preds(k))
                self.all_ground_truths.append(_records_to_gts(grec[n, : gcnt[n]]))
            if tp is not None:
                tp_h, bucket_h = tp.cpu().numpy().view(np.uint64), bucket.cpu().numpy()
                allk = np.concatenate(kept) if kept else np.zeros((0, 6))
                self._dev.append((allk[:, 0].astype(np.int64), allk[:, 1].copy(),
                                  np.concatenate([tp_h[n, : kc_h[n]] for n in range(len(kept))]) if kept else np.zeros(0, np.uint64),
                                  np.concatenate([grec[n, : gcnt[n], 0] for n in range(len(kept))]).astype(np.int64),
                                  np.concatenate([bucket_h[n, : gcnt[n]] for n in range(len(kept))])))
                self._dev_images += len(kept)
            return
        for i in range(predictions.shape[0]):
            self.all_predictions.append(self._apply_nms(self._parse_predictions(predictions[i])))
            self.all_ground_truths.append(self._parse_ground_truth(targets[i]))

    # ---- per-image helpers (same names/returns as the reference; used by its tests) -------------
    def _parse_predictions(self, pred: torch.Tensor) -> List[Pred]:
        if pred.is_cuda:
            from . import ops
            rec, cnt = ops.decode(pred.unsqueeze(0), self.conf_threshold, self.S, self.B, pred.shape[-1] - 5 * self.B)
            return _records_to_preds(rec[0, : int(cnt[0])].cpu().numpy())
        return _records_to_preds(_post_cpu.decode(pred.detach().numpy(), self.conf_threshold, self.S, self.B))

    def _parse_ground_truth(self, target: torch.Tensor) -> List[GT]:
        if target.is_cuda:
            from . import ops
            rec, cnt = ops.decode_gt(target.unsqueeze(0), self.S, self.B, target.shape[-1] - 5 * self.B)
            return _records_to_gts(rec[0, : int(cnt[0])].cpu().numpy())
        return _records_to_gts(_post_cpu.decode_gt(target.detach().numpy(), self.S, self.B))

    def _apply_nms(self, detections: List[Pred]) -> List[Pred]:
        if len(detections) == 0:
            return []
        rec = np.array([[c, f, *b] for c, f, b in detections], np.float64)
        keep = _post_cpu.nms(rec, self.nms_threshold, _post_cpu.METRICS)
        return [detections[k] for k in keep]

    def _calculate_iou(self, box1: Box, box2: Box) -> float:
        """IoU of two centre-format boxes; 0.0 when the union is empty (no epsilon in this variant)."""
        return _post_cpu.iou_scalar(box1, box2, _post_cpu.METRICS)

    # ------------------------------------------------------------------ AP machinery
    @staticmethod
    def _best_matches(preds: List[Tuple[int, float, Box]], gts: List[Tuple[int, Box]]):
        """preds: (img, conf, box) sorted by confidence; gts: (img, box).
        For every prediction the first ground truth of its image with the largest IoU > 0."""
        by_img: Dict[int, List[int]] = {}
        for g, (img, _) in enumerate(gts):
            by_img.setdefault(img, []).append(g)
        gt_boxes = np.array([b for _, b in gts], np.float64).reshape(-1, 4)
        best_iou = np.zeros(len(preds), np.float64)
        best_gt = np.full(len(preds), -1, np.int64)
        for k, (img, _conf, box) in enumerate(preds):
            cand = by_img.get(img)
            if not cand:
                continue
            ious = _post_cpu.iou_one_to_many(np.asarray(box, np.float64), gt_boxes[cand], _post_cpu.METRICS)
            j = int(np.argmax(ious))                   # first maximum == the reference's strict '>' scan
            if ious[j] > 0:
                best_iou[k], best_gt[k] = ious[j], cand[j]
        return best_iou, best_gt

    @staticmethod
    def _ap_at(best_iou, best_gt, n_gt: int, thr: float):
        """11-point AP + final precision/recall for one threshold given threshold-free matches."""
        n = len(best_iou)
        tp = np.zeros(n)
        taken = np.zeros(n_gt, bool)
        for k in range(n):
            g = best_gt[k]
            if best_iou[k] >= thr and g >= 0 and not taken[g]:
                tp[k] = 1
                taken[g] = True
        fp = 1.0 - tp
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        prec = np.concatenate(([1.0], ctp / (ctp + cfp + EPSILON)))
        rec = np.concatenate(([0.0], ctp / n_gt))
        ap = 0.0
        for t in np.linspace(0, 1, 11):
            sel = rec >= t
            ap += (np.max(prec[sel]) if np.any(sel) else 0) / 11
        return ap, (prec[-1] if n else 0.0), (rec[-1] if n else 0.0)

    def _class_lists(self, class_id: int):
        preds = [(i, p[1], p[2]) for i, ps in enumerate(self.all_predictions) for p in ps if p[0] == class_id]
        gts = [(i, g[1]) for i, gs in enumerate(self.all_ground_truths) for g in gs if g[0] == class_id]
        preds.sort(key=lambda t: t[1], reverse=True)  # stable, like sorted(..., reverse=True)
        return preds, gts

    def _calculate_ap_for_class(self, class_id: int, iou_threshold: float) -> Tuple[float, float, float]:
        preds, gts = self._class_lists(class_id)
        if len(gts) == 0 or len(preds) == 0:
            return 0.0, 0.0, 0.0
        bi, bg = self._best_matches(preds, gts)
        return self._ap_at(bi, bg, len(gts), iou_threshold)

    def _calculate_ap_for_size_class(self, class_id: int, iou_threshold: float,
                                     size_filtered_gts: List[Tuple[int, int, Box]]) -> float:
        gts = [(img, box) for img, c, box in size_filtered_gts if c == class_id]
        if len(gts) == 0:
            return 0.0
        preds, _ = self._class_lists(class_id)
        if len(preds) == 0:
            return 0.0
        bi, bg = self._best_matches(preds, gts)
        return self._ap_at(bi, bg, len(gts), iou_threshold)[0]

    def _calculate_overall_metrics(self, iou_threshold: float) -> Tuple[float, float]:
        """Micro precision / recall over all classes, predictions taken in stored order."""
        tp = fp = n_gt = 0
        for preds, gts in zip(self.all_predictions, self.all_ground_truths):
            n_gt += len(gts)
            taken = [False] * len(gts)
            gt_cls = np.array([g[0] for g in gts], np.int64)
            gt_box = np.array([g[1] for g in gts], np.float64).reshape(-1, 4)
            for c, _conf, box in preds:
                cand = np.nonzero(gt_cls == c)[0]
                hit = False
                if len(cand):
                    ious = _post_cpu.iou_one_to_many(np.asarray(box, np.float64), gt_box[cand], _post_cpu.METRICS)
                    j = int(np.argmax(ious))
                    if ious[j] > 0 and ious[j] >= iou_threshold and not taken[cand[j]]:
                        taken[cand[j]] = True
                        hit = True
                tp += hit
                fp += not hit
        return tp / (tp + fp + EPSILON), tp / (n_gt + EPSILON)

    def _compute_size_based_metrics(self) -> Dict[str, float]:
        """mAP for small / medium / large ground truths (area limits (32/448)^2 and (96/448)^2)."""
        small_t, medium_t = (32 / 448) ** 2, (96 / 448) ** 2
        buckets = {"small": [], "medium": [], "large": []}
        for img, gts in enumerate(self.all_ground_truths):
            for c, (x, y, w, h) in gts:
                area = w * h
                name = "small" if area < small_t else ("medium" if area < medium_t else "large")
                buckets[name].append((img, c, (x, y, w, h)))
        out: Dict[str, float] = {}
        for name in ("small", "medium", "large"):
            sg = buckets[name]
            if len(sg) == 0:
                out[f"mAP50:95_{name}"] = 0.0
                out[f"mAP50_{name}"] = 0.0
                out[f"mAP75_{name}"] = 0.0
                continue
            per_thr = {t: [] for t in self.iou_thresholds}
            for c in range(self.num_classes):
                gts = [(img, box) for img, cc, box in sg if cc == c]
                preds = self._class_lists(c)[0] if gts else []
                match = self._best_matches(preds, gts) if gts and preds else None
                for t in self.iou_thresholds:
                    per_thr[t].append(self._ap_at(match[0], match[1], len(gts), t)[0] if match is not None else 0.0)
            if 0.5 in self.iou_thresholds:
                out[f"mAP50_{name}"] = np.mean(per_thr[0.5])
            if 0.75 in self.iou_thresholds:
                out[f"mAP75_{name}"] = np.mean(per_thr[0.75])
            out[f"mAP50:95_{name}"] = np.mean([a for v in per_thr.values() for a in v])
        out["num_small_objects"] = len(buckets["small"])
        out["num_medium_objects"] = len(buckets["medium"])
        out["num_large_objects"] = len(buckets["large"])
        return out

    def compute(self) -> Dict[str, float]:
        """mAP50:95 / mAP50 / mAP75, per-class AP, overall precision & recall at IoU 0.5, size buckets."""
        if len(self.all_predictions) == 0:
            return {"mAP50:95": 0.0, "mAP50": 0.0, "mAP75": 0.0, "precision": 0.0, "recall": 0.0}
        if self._dev and self._dev_images == len(self.all_predictions) == len(self.all_ground_truths) \
                and sum(len(d[0]) for d in self._dev) == sum(len(p) for p in self.all_predictions):
            return self._compute_from_device_matches()
        results: Dict[str, float] = {}
        per_thr = {t: [] for t in self.iou_thresholds}
        for c in range(self.num_classes):
            preds, gts = self._class_lists(c)
            match = self._best_matches(preds, gts) if (preds and gts) else None
            aps = []
            for t in self.iou_thresholds:
                ap = self._ap_at(match[0], match[1], len(gts), t)[0] if match is not None else 0.0
                per_thr[t].append(ap)
                aps.append(ap)
                if t == 0.5:
                    results[f"AP50_class_{c}"] = ap
                elif t == 0.75:
                    results[f"AP75_class_{c}"] = ap
            results[f"AP50:95_class_{c}"] = np.mean(aps)
        if 0.5 in self.iou_thresholds:
            results["mAP50"] = np.mean(per_thr[0.5])
        if 0.75 in self.iou_thresholds:
            results["mAP75"] = np.mean(per_thr[0.75])
        results["mAP50:95"] = np.mean([a for v in per_thr.values() for a in v])
        results["precision"], results["recall"] = self._calculate_overall_metrics(iou_threshold=0.5)
        results.update(self._compute_size_based_metrics())
        return results


    # ------------------------------------------------------------------ device-matched fast path
    @staticmethod
    def _ap_from_tp(tp: np.ndarray, n_gt: int) -> float:
        """the arithmetic of _ap_at on a ready 0/1 vector (predictions already in confidence order)."""
        fp = 1.0 - tp
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        prec = np.concatenate(([1.0], ctp / (ctp + cfp + EPSILON)))
        rec = np.concatenate(([0.0], ctp / n_gt))
        ap = 0.0
        for t in np.linspace(0, 1, 11):
            sel = rec >= t
            ap += (np.max(prec[sel]) if np.any(sel) else 0) / 11
        return ap

    def _compute_from_device_matches(self) -> Dict[str, float]:
        cls = np.concatenate([d[0] for d in self._dev])
        conf = np.concatenate([d[1] for d in self._dev])
        bits = np.concatenate([d[2] for d in self._dev])
        gcls = np.concatenate([d[3] for d in self._dev])
        gbkt = np.concatenate([d[4] for d in self._dev])
        T = len(self.iou_thresholds)
        T1 = T + 1
        # per class: the predictions in (stable) confidence order -- images and lists are already in the reference's order
        order = {}
        for c in range(self.num_classes):
            idx = np.nonzero(cls == c)[0]
            order[c] = idx[np.argsort(-conf[idx], kind="stable")]

        def ap(c, v, t, n_gt):
            o = order[c]
            if n_gt == 0 or len(o) == 0:
                return 0.0
            return self._ap_from_tp(((bits[o] >> np.uint64(v * T1 + t)) & np.uint64(1)).astype(np.float64), n_gt)

        results: Dict[str, float] = {}
        per_thr = {t: [] for t in self.iou_thresholds}
        for c in range(self.num_classes):
            n_gt = int(np.count_nonzero(gcls == c))
            aps = []
            for ti, t in enumerate(self.iou_thresholds):
                a = ap(c, 0, ti, n_gt)
                per_thr[t].append(a)
                aps.append(a)
                if t == 0.5:
                    results[f"AP50_class_{c}"] = a
                elif t == 0.75:
                    results[f"AP75_class_{c}"] = a
            results[f"AP50:95_class_{c}"] = np.mean(aps)
        if 0.5 in self.iou_thresholds:
            results["mAP50"] = np.mean(per_thr[0.5])
        if 0.75 in self.iou_thresholds:
            results["mAP75"] = np.mean(per_thr[0.75])
        results["mAP50:95"] = np.mean([a for v in per_thr.values() for a in v])
        tp = int(np.count_nonzero((bits >> np.uint64(T)) & np.uint64(1)))      # set `all`, the extra 0.5 threshold
        fp = len(bits) - tp
        results["precision"], results["recall"] = tp / (tp + fp + EPSILON), tp / (len(gcls) + EPSILON)
        for v, name in ((1, "small"), (2, "medium"), (3, "large")):
            if not np.any(gbkt == v):
                results[f"mAP50:95_{name}"] = 0.0
                results[f"mAP50_{name}"] = 0.0
                results[f"mAP75_{name}"] = 0.0
                continue
            pt = {t: [] for t in self.iou_thresholds}
            for c in range(self.num_classes):
                n_gt = int(np.count_nonzero((gcls == c) & (gbkt == v)))
                for ti, t in enumerate(self.iou_thresholds):
                    pt[t].append(ap(c, v, ti, n_gt))
            if 0.5 in self.iou_thresholds:
                results[f"mAP50_{name}"] = np.mean(pt[0.5])
            if 0.75 in self.iou_thresholds:
                results[f"mAP75_{name}"] = np.mean(pt[0.75])
            results[f"mAP50:95_{name}"] = np.mean([a for vv in pt.values() for a in vv])
        results["num_small_objects"] = int(np.count_nonzero(gbkt == 1))
        results["num_medium_objects"] = int(np.count_nonzero(gbkt == 2))
        results["num_large_objects"] = int(np.count_nonzero(gbkt == 3))
        return results


def evaluate_model(model: nn.Module, dataloader: DataLoader, device: str, num_classes: int = 20,
                   iou_thresholds: List[float] = None, conf_threshold: float = 0.01, nms_threshold: float = 0.4,
                   S: int = 7, B: int = 2) -> Dict[str, float]:
    """Run ``model`` over ``dataloader`` and return the ``mAPMetric.compute()`` dictionary."""
    model.eval()
    metric = mAPMetric(num_classes=num_classes, iou_thresholds=iou_thresholds, conf_threshold=conf_threshold,
                       nms_threshold=nms_threshold, S=S, B=B)
    try:
        from tqdm import tqdm
        batches = tqdm(dataloader, desc="Evaluating", unit="batch")
    except ImportError:  # pragma: no cover
        batches = dataloader
    with torch.no_grad():
        for images, targets in batches:
            images = images.to(device)
            targets = targets.to(device)
            metric.update(model(images), targets)
    return metric.compute()
