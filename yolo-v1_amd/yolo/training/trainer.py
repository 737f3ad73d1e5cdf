"""Epoch loop with the reference's call signatures (src/yolo/training/trainer.py:23,133,220).

One train step = zero_grad -> model(images) -> criterion -> backward -> clip_grad_norm_(10) -> step.
On a ROCm device every piece of that runs on the HIP kernels: the model and the loss as single
autograd nodes, and -- when the optimizer is ``yolo.optim.Adam`` built with ``max_grad_norm=10`` --
clipping + Adam as one fused pass (the separate clip call is then skipped).  With
``torch.distributed`` initialised, gradients are averaged over the ranks before the update
(yolo.parallel; the reference is single-device).
"""

from __future__ import annotations

import time

import torch
import torch.distributed as dist

from ..metrics import evaluate_model
from ..parallel import make_grad_reducer
from .checkpoints import save_best_map_model, save_best_model, save_checkpoint

_PARTS = ("total", "coord", "conf_obj", "conf_noobj", "class")
_CLIP = 10.0


def train_epoch(model, dataloader, criterion, optimizer, device, epoch: int, writer=None, scaler=None) -> dict[str, float]:
    """One pass over ``dataloader``; returns the mean of each loss component.

    ``scaler`` (the reference's fp16 autocast + GradScaler branch, trainer.py:69-83) is accepted and not used: on a ROCm
    device the engine already computes in bf16 with fp32 accumulation and fp32 master weights, which needs no loss scaling;
    on the CPU the stock fp32 path runs."""
    model.train()
    sums = dict.fromkeys(_PARTS, 0.0)
    n = 0
    fused_clip = getattr(optimizer, "max_grad_norm", None) is not None
    # data parallel: the reducer is chosen once per model (it may attach a gradient arena to the model's plan) -- on a GPU
    # the all-reduce overlaps the backward pass, the path bench.py measures
    allreduce = None
    if dist.is_available() and dist.is_initialized():
        allreduce = getattr(model, "_yolo_grad_reducer", None)
        if allreduce is None:
            allreduce = make_grad_reducer(model, device)
            model._yolo_grad_reducer = allreduce
    t0 = time.time()
    for batch_idx, (images, targets) in enumerate(dataloader):
        images = images.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        optimizer.zero_grad(set_to_none=True)
        loss, parts = criterion(model(images), targets)
        loss.backward()
        if allreduce is not None:
            allreduce.all_reduce_mean()
        if not fused_clip:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=_CLIP)
        if hasattr(optimizer, "skip_if"):
            # yolo.optim.Adam: a step whose targets the loss kernel flagged as invalid updates nothing (the error itself surfaces at
            # the read of `parts` below; the reference raises inside the loss, before any update)
            optimizer.skip_if = getattr(parts, "device_flag", None)
        optimizer.step()
        for k in _PARTS:
            sums[k] += parts[k]
        n += 1
        if (batch_idx + 1) % 10 == 0:
            print(f"Epoch [{epoch}] Batch [{batch_idx + 1}/{len(dataloader)}] Loss: {parts['total']:.4f} "
                  f"(coord: {parts['coord']:.4f}, conf_obj: {parts['conf_obj']:.4f}, conf_noobj: {parts['conf_noobj']:.4f}, "
                  f"class: {parts['class']:.4f}) Time: {time.time() - t0:.2f}s")
            if writer is not None:
                step = (epoch - 1) * len(dataloader) + batch_idx
                for k in _PARTS:
                    writer.add_scalar(f"batch/{k}_loss", parts[k], step)
            t0 = time.time()
    return {k: v / max(n, 1) for k, v in sums.items()}


def validate(model, dataloader, criterion, device, compute_map: bool = False, num_classes: int = 20) -> dict[str, float]:
    """Mean loss components over ``dataloader`` (+ mAP metrics when ``compute_map``)."""
    model.eval()
    sums = dict.fromkeys(_PARTS, 0.0)
    n = 0
    with torch.no_grad():
        for images, targets in dataloader:
            _, parts = criterion(model(images.to(device)), targets.to(device))
            for k in _PARTS:
                sums[k] += parts[k]
            n += 1
    results = {k: v / max(n, 1) for k, v in sums.items()}
    if compute_map:
        m = evaluate_model(model=model, dataloader=dataloader, device=device, num_classes=num_classes,
                           iou_thresholds=None, conf_threshold=0.01, nms_threshold=0.4)
        for k in ("mAP50:95", "mAP50", "mAP75", "precision", "recall", "mAP50:95_small", "mAP50:95_medium", "mAP50:95_large"):
            if k in m:
                results[k] = m[k]
    return results


def train(model, train_loader, val_loader, criterion, optimizer, scheduler, device, num_epochs: int, checkpoint_dir,
          save_frequency: int = 5, writer=None, compute_map: bool = False, map_frequency: int = 5, num_classes: int = 20,
          start_epoch: int = 1, best_val_loss_init: float = None, best_map_init: float = None, scaler=None) -> dict[str, float]:
    """Epoch loop with the reference's checkpoint policy: latest every epoch, every ``save_frequency``
    epochs, best validation loss, best mAP50:95."""
    best_val = float("inf") if best_val_loss_init is None else best_val_loss_init
    best_map = 0.0 if best_map_init is None else best_map_init
    final_train = None
    for epoch in range(start_epoch, num_epochs + 1):
        print(f"\n===== Epoch {epoch}/{num_epochs} =====")
        tr = train_epoch(model, train_loader, criterion, optimizer, device, epoch, writer, scaler)
        print("  train:", {k: round(v, 4) for k, v in tr.items()})
        want_map = compute_map and (epoch % map_frequency == 0 or epoch == num_epochs)
        va = validate(model, val_loader, criterion, device, compute_map=want_map, num_classes=num_classes)
        print("  val:  ", {k: round(float(v), 4) for k, v in va.items()})
        scheduler.step()
        lr = optimizer.param_groups[0]["lr"]
        print(f"  learning rate: {lr:.6f}")
        if writer is not None:
            for k in _PARTS:
                writer.add_scalar(f"epoch/train_{k}", tr[k], epoch)
                writer.add_scalar(f"epoch/val_{k}", va[k], epoch)
            writer.add_scalar("epoch/lr", lr, epoch)
        # checkpoints are written by rank 0 only (every rank holds the same parameters and optimizer state; BatchNorm running
        # statistics are per rank -- there is no SyncBN, as in the reference -- and rank 0's are the ones saved); the other
        # ranks wait, so that nobody runs ahead of a file that a later --resume on all ranks would read
        writer_rank = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        if writer_rank:
            save_checkpoint(checkpoint_dir / "yolo_latest.pth", epoch, model, optimizer, scheduler, tr, va)
            if epoch % save_frequency == 0:
                save_checkpoint(checkpoint_dir / f"yolo_epoch_{epoch}.pth", epoch, model, optimizer, scheduler, tr, va)
        if va["total"] < best_val:
            best_val = va["total"]
            if writer_rank:
                save_best_model(checkpoint_dir / "yolo_best.pth", epoch, model, optimizer, va, "val_loss", best_val)
        if "mAP50:95" in va and va["mAP50:95"] > best_map:
            best_map = va["mAP50:95"]
            if writer_rank:
                save_best_map_model(checkpoint_dir / "yolo_best_map.pth", epoch, model, optimizer, va, best_map)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        final_train = tr["total"]
    out = {"best_val_loss": best_val, "final_train_loss": final_train}
    if best_map > 0:
        out["best_mAP50:95"] = best_map
    return out
