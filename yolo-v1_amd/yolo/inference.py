"""Single-image inference pipeline with the reference's surface (src/yolo/inference.py:12-317).

``predict`` = load -> resize/normalise -> model -> decode -> NMS.  On a ROCm device the model runs on
the HIP engine and decode + NMS run as two kernel launches with ONE device->host copy, instead of
~880 ``.item()`` syncs per image (inference.py:184-191).
"""

from __future__ import annotations

import warnings

import numpy as np
import torch
import torch.nn as nn
from PIL import Image

from . import _post_cpu
from .schemas import BoundingBox, Detection

EPSILON = 1e-6  # in the IoU denominator of this variant (reference inference.py:9,248)

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


class _Preprocess:
    """Resize((448,448)) -> ToTensor -> Normalize(ImageNet) without torchvision.
    PIL's bilinear ``resize`` is what torchvision's ``Resize`` calls for PIL inputs."""

    def __init__(self, size=(448, 448), mean=_MEAN, std=_STD):
        self.size = size
        self.mean_t, self.std_t = tuple(mean), tuple(std)
        self.mean = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)

    def __call__(self, image: Image.Image) -> torch.Tensor:
        img = image.resize((self.size[1], self.size[0]), Image.BILINEAR)
        arr = np.asarray(img, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
        return (t - self.mean) / self.std


def _default_device() -> str:
    if torch.backends.mps.is_available():
        return "mps"
    return "cuda" if torch.cuda.is_available() else "cpu"


class YOLOInference:
    """Inference engine: ``device`` / ``model`` / ``transform`` attributes and the methods of the reference."""

    def __init__(self, model: nn.Module, device: str | None = None) -> None:
        self.device = _default_device() if device is None else device
        self.model = model.to(self.device)
        self.model.eval()
        self.transform = _Preprocess()

    # ------------------------------------------------------------------ image handling
    def load_image(self, image_path: str) -> Image.Image:
        return Image.open(image_path).convert("RGB")

    def preprocess_image(self, image: Image.Image) -> torch.Tensor:
        if self._on_gpu() and isinstance(self.transform, _Preprocess) and image.mode == "RGB":
            # device path: ship the decoded uint8 pixels (3 B each, not 12) and resize + normalise there --
            # bit-identical to the host transform (yolo/preprocess.py)
            from .preprocess import preprocess_u8
            u8 = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).unsqueeze(0).to(self.device)
            out, _ = preprocess_u8(u8, self.transform.size, self.transform.mean_t, self.transform.std_t)
            return out
        return self.transform(image).unsqueeze(0).to(self.device)

    def _on_gpu(self) -> bool:
        return torch.device(self.device).type == "cuda"

    # ------------------------------------------------------------------ pipeline
    def predict(self, image_path: str, conf_threshold: float = 0.5, nms_threshold: float = 0.4,
                class_names: list[str] | None = None) -> list[Detection]:
        image = self.load_image(image_path)
        x = self.preprocess_image(image)
        with torch.no_grad():
            pred = self.model(x)
        if pred.is_cuda:
            from . import ops
            S, B = self.model.S, self.model.B
            C = pred.shape[-1] - 5 * B
            (rec, keep), = ops.postprocess_host(pred[:1], conf_threshold, nms_threshold, ops._hip.NMS_INFERENCE, S, B, C)
            dets = self._records_to_detections(rec, class_names)   # validates every decoded box like the reference
            return [dets[k] for k in keep]
        dets = self.parse_predictions(pred[0], conf_threshold, class_names)
        return self.non_max_suppression(dets, nms_threshold)

    @staticmethod
    def _records_to_detections(rec: np.ndarray, class_names) -> list[Detection]:
        out = []
        for c, conf, x, y, w, h in rec.tolist():
            cid = int(c)
            out.append(Detection(class_id=cid, class_name=class_names[cid] if class_names else f"class_{cid}",
                                 confidence=conf, bbox=BoundingBox(x=x, y=y, width=w, height=h)))
        return out

    def parse_predictions(self, pred: torch.Tensor, conf_threshold: float, class_names: list[str] | None = None) -> list[Detection]:
        """(S,S,5B+C) -> Detections with conf*prob > threshold, (row, col, box) scan order."""
        S, B = self.model.S, self.model.B
        if pred.is_cuda:
            from . import ops
            rec, cnt = ops.decode(pred.unsqueeze(0), conf_threshold, S, B, pred.shape[-1] - 5 * B)
            rec = rec[0, : int(cnt[0])].cpu().numpy()
        else:
            rec = _post_cpu.decode(pred.detach().numpy(), conf_threshold, S, B)
        return self._records_to_detections(rec, class_names)

    def iou(self, bbox1: BoundingBox, bbox2: BoundingBox) -> float:
        """IoU of two boxes with +EPSILON in the denominator (Python-float arithmetic)."""
        ax1, ay1, ax2, ay2 = bbox1.to_corners()
        bx1, by1, bx2, by2 = bbox2.to_corners()
        iw = max(0, min(ax2, bx2) - max(ax1, bx1))
        ih = max(0, min(ay2, by2) - max(ay1, by1))
        inter = iw * ih
        return inter / (bbox1.area + bbox2.area - inter + EPSILON)

    def non_max_suppression(self, detections: list[Detection], nms_threshold: float = None, iou_threshold: float = None) -> list[Detection]:
        """Greedy per-class NMS; a box survives a kept one iff its class differs or IoU < threshold.
        ``iou_threshold`` is the deprecated spelling and wins when given."""
        if iou_threshold is not None:
            warnings.warn("Parameter 'iou_threshold' is deprecated, use 'nms_threshold' instead.", DeprecationWarning, stacklevel=2)
            thr = iou_threshold
        elif nms_threshold is not None:
            thr = nms_threshold
        else:
            thr = 0.4
        if len(detections) == 0:
            return []
        rec = np.array([[d.class_id, d.confidence, d.bbox.x, d.bbox.y, d.bbox.width, d.bbox.height] for d in detections], np.float64)
        if self._on_gpu():
            # a GPU model never takes a host loop silently: the device kernels cover up to 1024 boxes per image
            # (S = 14, B = 3 -> 588); a longer list is refused by yolo_nms and that error propagates
            from . import ops
            cap = 128 if len(rec) <= 128 else max(len(rec), 129)
            rec_d = torch.zeros((1, cap, 6), dtype=torch.float64, device=self.device)
            rec_d[0, : len(rec)] = torch.from_numpy(rec)
            cnt = torch.tensor([len(rec)], dtype=torch.int32, device=self.device)
            keep, kc = ops.nms(rec_d, cnt, thr, ops._hip.NMS_INFERENCE)
            keep = keep[0, : int(kc[0])].cpu().numpy()
        else:
            keep = _post_cpu.nms(rec, thr, _post_cpu.INFERENCE)
        return [detections[k] for k in keep]
