"""MI355X-native YOLOv1: the module surface of mattiaskvist/yolo-v1's ``src/yolo`` package
(``__all__`` = reference ``yolo/__init__.py:3-31``) on hand-written gfx950 HIP kernels."""

from .dataset import CombinedVOCDataset, VOCDetectionYOLO, create_voc_datasets
from .loss import YOLOLoss
from .metrics import evaluate_model, mAPMetric
from .models import YOLOv1, Backbone, DetectionHead, ResNetBackbone, YOLOv1Backbone
from .schemas import BoundingBox, Detection

# north_star spellings as aliases of the reference's names
YoloLoss = YOLOLoss
from .compat import cellboxes_to_boxes, mean_average_precision, non_max_suppression  # noqa: E402,F401

__all__ = [
    "Backbone",
    "BoundingBox",
    "CombinedVOCDataset",
    "Detection",
    "DetectionHead",
    "ResNetBackbone",
    "VOCDetectionYOLO",
    "YOLOLoss",
    "YOLOv1",
    "YOLOv1Backbone",
    "create_voc_datasets",
    "evaluate_model",
    "mAPMetric",
]
