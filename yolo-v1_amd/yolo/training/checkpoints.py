"""Checkpoint files with the reference's dictionary keys (src/yolo/training/checkpoints.py:32-113), so
checkpoints are interchangeable in both directions: ``epoch, model_state_dict, optimizer_state_dict,
[scheduler_state_dict], [train_loss], val_loss, [mAP50:95, mAP50, mAP75]``."""

from __future__ import annotations

import os
from pathlib import Path

import torch


def _atomic_save(data: dict, path) -> None:
    """write next to the target and rename: a reader (or a crash) never sees a half-written checkpoint"""
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(data, tmp)
    os.replace(tmp, path)


def _with_map(data: dict, val_losses: dict) -> dict:
    if "mAP50:95" in val_losses:
        for k in ("mAP50:95", "mAP50", "mAP75"):
            data[k] = float(val_losses[k])       # plain floats: NumPy scalars would need pickle to load (weights_only=True refuses them)
    return data


def save_checkpoint(checkpoint_path: Path, epoch: int, model, optimizer, scheduler, train_losses: dict, val_losses: dict) -> None:
    data = {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
            "scheduler_state_dict": scheduler.state_dict(), "train_loss": float(train_losses["total"]), "val_loss": float(val_losses["total"])}
    _atomic_save(_with_map(data, val_losses), checkpoint_path)
    print(f"  checkpoint saved: {checkpoint_path}")


def save_best_model(checkpoint_path: Path, epoch: int, model, optimizer, val_losses: dict, metric_name: str, metric_value: float) -> None:
    data = {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "val_loss": float(val_losses["total"])}
    _atomic_save(_with_map(data, val_losses), checkpoint_path)
    print(f"  new best model ({metric_name}={metric_value:.4f}) saved: {checkpoint_path}")


def save_best_map_model(checkpoint_path: Path, epoch: int, model, optimizer, val_losses: dict, best_map: float) -> None:
    save_best_model(checkpoint_path, epoch, model, optimizer, val_losses, "mAP50:95", best_map)
