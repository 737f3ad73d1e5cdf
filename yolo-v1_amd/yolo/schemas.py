"""Value types returned by the detection hot path (API of the reference's src/yolo/schemas.py:6-89).

pydantic models with the same field names, range validation and helper methods, so callers that
build or consume ``Detection`` / ``BoundingBox`` objects keep working unchanged.
"""

from __future__ import annotations

from pydantic import BaseModel, Field


class BoundingBox(BaseModel):
    """Centre-format box in normalised image coordinates; every field is validated to [0, 1]."""

    x: float = Field(..., ge=0.0, le=1.0, description="centre x (normalised)")
    y: float = Field(..., ge=0.0, le=1.0, description="centre y (normalised)")
    width: float = Field(..., ge=0.0, le=1.0, description="width (normalised)")
    height: float = Field(..., ge=0.0, le=1.0, description="height (normalised)")

    def to_corners(self) -> tuple[float, float, float, float]:
        """(x1, y1, x2, y2), still normalised."""
        hw, hh = self.width / 2, self.height / 2
        return (self.x - hw, self.y - hh, self.x + hw, self.y + hh)

    def to_pixel_coords(self, img_width: int, img_height: int) -> tuple[int, int, int, int]:
        """Corner box scaled to an image of the given size, truncated to ints."""
        x1, y1, x2, y2 = self.to_corners()
        return (int(x1 * img_width), int(y1 * img_height), int(x2 * img_width), int(y2 * img_height))

    @property
    def area(self) -> float:
        return self.width * self.height

    @classmethod
    def from_corners(cls, x1: float, y1: float, x2: float, y2: float) -> "BoundingBox":
        w, h = x2 - x1, y2 - y1
        return cls(x=x1 + w / 2, y=y1 + h / 2, width=w, height=h)

    def __str__(self) -> str:
        x1, y1, x2, y2 = self.to_corners()
        return f"({x1:.2f}, {y1:.2f}, {x2:.2f}, {y2:.2f})"


class Detection(BaseModel):
    """One detected object."""

    class_id: int = Field(..., ge=0, description="predicted class index")
    class_name: str | None = Field(None, description="human-readable class label")
    confidence: float = Field(..., ge=0.0, le=1.0, description="objectness x class probability")
    bbox: BoundingBox = Field(..., description="box in normalised coordinates")
