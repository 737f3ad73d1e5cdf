"""Drawing helpers with the reference's call surface (src/yolo/utils/visualization.py:34-147): presentation only, outside
the hot path.  ``draw_detections(image, detections, class_names=None, conf_threshold=0.5, box_width=3, font_size=20)`` is
what ``src/predict.py:113`` calls; ``VOC_CLASSES`` is the PASCAL VOC label list."""

from __future__ import annotations

import os

from PIL import Image, ImageDraw, ImageFont

VOC_CLASSES = [
    "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
    "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor",
]

_PALETTE = ("red", "green", "blue", "magenta", "cyan", "orange", "purple", "pink", "lime")   # colour = class_id mod 9, as the reference
_FONT_FILES = ("/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf", "/usr/share/fonts/truetype/liberation/LiberationSans-Regular.ttf",
               "/usr/share/fonts/truetype/freefont/FreeSans.ttf", "/System/Library/Fonts/Helvetica.ttc", "/Library/Fonts/Arial.ttf",
               r"C:\Windows\Fonts\arial.ttf")


def _load_font(size: int = 20):
    """first TrueType font found among the usual system locations at ``size`` points, else PIL's built-in bitmap font"""
    for path in _FONT_FILES + ("DejaVuSans.ttf",):
        try:
            if os.path.isabs(path) and not os.path.exists(path):
                continue
            return ImageFont.truetype(path, size)
        except OSError:
            continue
    return ImageFont.load_default()


def _fields(det, class_names):
    """(class_id, confidence, class_name, (x1, y1, x2, y2) normalised) of a Detection or of a legacy tuple
    (class_id, confidence, x, y, w, h)"""
    if hasattr(det, "bbox"):
        name = getattr(det, "class_name", None)
        if name is None and class_names is not None:
            name = class_names[det.class_id]
        return det.class_id, det.confidence, name, det.bbox.to_corners()
    cid, conf, x, y, w, h = det
    cid = int(cid)
    name = class_names[cid] if class_names is not None else None
    return cid, float(conf), name, (x - w / 2, y - h / 2, x + w / 2, y + h / 2)


def draw_detections(image: Image.Image, detections: list, class_names: list[str] | None = None, conf_threshold: float = 0.5,
                    box_width: int = 3, font_size: int = 20) -> Image.Image:
    """Return a COPY of ``image`` with a coloured box and a "name: 0.95" label per detection whose confidence is at least
    ``conf_threshold``.  Boxes are clamped to the image; boxes narrower or lower than 2 pixels are skipped."""
    out = image.copy()
    draw = ImageDraw.Draw(out)
    font = _load_font(font_size)
    W, H = image.size
    for det in detections:
        cid, conf, name, (cx1, cy1, cx2, cy2) = _fields(det, class_names)
        if conf < conf_threshold:
            continue
        xs = sorted((int(cx1 * W), int(cx2 * W)))
        ys = sorted((int(cy1 * H), int(cy2 * H)))
        x1, x2 = (max(0, min(v, W - 1)) for v in xs)
        y1, y2 = (max(0, min(v, H - 1)) for v in ys)
        if x2 - x1 < 2 or y2 - y1 < 2:
            continue
        colour = _PALETTE[cid % len(_PALETTE)]
        draw.rectangle([x1, y1, x2, y2], outline=colour, width=box_width)
        label = f"{name if name is not None else cid}: {conf:.2f}"
        anchor = (x1, y1 - 25)
        draw.rectangle(draw.textbbox(anchor, label, font=font), fill=colour)
        draw.text(anchor, label, fill="white", font=font)
    return out
