#!/usr/bin/env python3
"""Training entry point (the reference's src/train.py without the Modal cloud wrapper).

    python yolo-v1_amd/train.py --device cuda --backbone yolov1 --synthetic 512 --epochs 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 yolo-v1_amd/train.py --device cuda ...

Defaults follow src/train.py:269-338: batch 64, lr 1e-4, weight decay 5e-4, LR decay x0.1 at epochs
75 and 105, lambda_coord 5, lambda_noobj 0.5.  ``--backbone`` is additive (the reference hard-codes
ResNet50, which needs torchvision); ``--synthetic N`` trains on N random images instead of PASCAL VOC.
"""

from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402
from torch.utils.data.distributed import DistributedSampler  # noqa: E402

from yolo import YOLOLoss, YOLOv1, ResNetBackbone, YOLOv1Backbone  # noqa: E402
from yolo import training  # noqa: E402
from yolo.dataset import SyntheticYOLODataset, create_voc_datasets  # noqa: E402
from yolo.parallel import broadcast_parameters  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--backbone", choices=["resnet50", "yolov1"], default="resnet50")
    ap.add_argument("--batch-size", type=int, default=64, help="per process")
    ap.add_argument("--num-workers", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=135)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--weight-decay", type=float, default=5e-4)
    ap.add_argument("--lr-decay-epochs", default="75,105")
    ap.add_argument("--lambda-coord", type=float, default=5.0)
    ap.add_argument("--lambda-noobj", type=float, default=0.5)
    ap.add_argument("--save-frequency", type=int, default=10)
    ap.add_argument("--freeze-backbone", action="store_true")
    ap.add_argument("--no-pretrained", action="store_true", help="random-init ResNet50 (no torchvision / ImageNet weights available)")
    ap.add_argument("--compute-map", action="store_true")
    ap.add_argument("--checkpoint-dir", default="checkpoints")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="train on N synthetic images (no dataset needed)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = a.device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device == "cuda":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if device == "cuda" else "gloo")

    if a.synthetic:
        train_ds, val_ds = SyntheticYOLODataset(a.synthetic, seed=0), SyntheticYOLODataset(max(a.batch_size, a.synthetic // 8), seed=1)
    else:
        train_ds = create_voc_datasets([("2007", "trainval"), ("2012", "train")], augment=True)     # the reference's splits (src/train.py:106-122)
        val_ds = create_voc_datasets([("2012", "val")], augment=False)
    sampler = DistributedSampler(train_ds, num_replicas=world, rank=rank) if world > 1 else None
    train_loader = DataLoader(train_ds, batch_size=a.batch_size, shuffle=sampler is None, sampler=sampler, num_workers=a.num_workers,
                              pin_memory=device == "cuda", drop_last=True)
    val_loader = DataLoader(val_ds, batch_size=a.batch_size, shuffle=False, num_workers=a.num_workers, pin_memory=device == "cuda")

    backbone = YOLOv1Backbone() if a.backbone == "yolov1" else ResNetBackbone(pretrained=not a.no_pretrained, freeze=a.freeze_backbone)
    model = YOLOv1(backbone=backbone, num_classes=20, S=7, B=2).to(device)
    if world > 1:
        broadcast_parameters(model)
    criterion = YOLOLoss(S=7, B=2, C=20, lambda_coord=a.lambda_coord, lambda_noobj=a.lambda_noobj)
    params = [p for p in model.parameters() if p.requires_grad]
    if device == "cuda":
        from yolo.optim import Adam          # fused clip(10) + Adam on the HIP kernels
        optimizer = Adam(params, lr=a.lr, weight_decay=a.weight_decay, max_grad_norm=10.0)
        if model._fusable():
            # the Linear layers' update runs as a background pass beside the next forward's conv stack (11.40 vs 11.58 ms per step
            # at batch 64: the persistent conv kernels draw their tiles from a queue, so the held CUs cost only their share)
            optimizer.attach_plan(model.hip_plan(), overlap=True)
        elif hasattr(model.head, "hip_plan"):          # DetectionHead on a ResNet trunk: its Linear layers' bf16 operands
            optimizer.attach_plan(model.head.hip_plan())
    else:
        optimizer = torch.optim.Adam(params, lr=a.lr, weight_decay=a.weight_decay)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[int(e) for e in a.lr_decay_epochs.split(",")], gamma=0.1)

    start_epoch, best_val, best_map = 1, None, None
    if a.resume:
        ck = torch.load(a.resume, map_location=device, weights_only=True)
        model.load_state_dict(ck["model_state_dict"])
        optimizer.load_state_dict(ck["optimizer_state_dict"])
        if "scheduler_state_dict" in ck:
            scheduler.load_state_dict(ck["scheduler_state_dict"])
        start_epoch = ck["epoch"] + 1
        best_val, best_map = ck.get("val_loss"), ck.get("mAP50:95")

    ckdir = Path(a.checkpoint_dir)
    if rank == 0:
        ckdir.mkdir(parents=True, exist_ok=True)
    res = training.train(model, train_loader, val_loader, criterion, optimizer, scheduler, device, a.epochs, ckdir,
                         save_frequency=a.save_frequency, compute_map=a.compute_map, start_epoch=start_epoch,
                         best_val_loss_init=best_val, best_map_init=best_map)
    if rank == 0:
        print("done:", res)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
