"""Presentation helpers re-exported for predict.py (reference src/yolo/utils/__init__.py)."""

from .visualization import VOC_CLASSES, draw_detections

__all__ = ["VOC_CLASSES", "draw_detections"]
