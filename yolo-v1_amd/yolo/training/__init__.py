"""Training orchestration (glue around the HIP hot path; reference: src/yolo/training/)."""

from .checkpoints import save_best_map_model, save_best_model, save_checkpoint
from .trainer import train, train_epoch, validate

__all__ = ["save_best_map_model", "save_best_model", "save_checkpoint", "train", "train_epoch", "validate"]
