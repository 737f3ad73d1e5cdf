"""Minimal drawing helpers (presentation only, outside the hot path; reference
src/yolo/utils/visualization.py).  ``VOC_CLASSES`` is the PASCAL VOC label list."""

from __future__ import annotations

from PIL import Image, ImageDraw

VOC_CLASSES = [
    "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
    "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor",
]


def draw_detections(image: Image.Image, detections, line_width: int = 3) -> Image.Image:
    """Return a copy of ``image`` with one rectangle + label per detection."""
    out = image.copy()
    d = ImageDraw.Draw(out)
    W, H = out.size
    for det in detections:
        x1, y1, x2, y2 = det.bbox.to_pixel_coords(W, H)
        d.rectangle([x1, y1, x2, y2], outline="red", width=line_width)
        d.text((x1 + 2, max(0, y1 - 12)), f"{det.class_name or det.class_id}: {det.confidence:.2f}", fill="red")
    return out
