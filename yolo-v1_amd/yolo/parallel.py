"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-device (src/train.py:83,146); this is new capability: every
rank holds a full replica, takes 1/world of the global batch, and the fp32 gradients are averaged
with all-reduce before clip_grad_norm_/Adam -- identical to the single-device step on the global
batch because YOLOLoss divides by the local N and the shards are equal.

xGMI is point-to-point (7 links/GPU): few, large messages.  The 822 MB FC1 gradient and the other
>= 32 MB tensors are reduced in place, each as one collective; everything smaller (~100 tensors,
biases and early convs) is packed into one flat buffer -> one more collective.
``backend="nccl"`` is RCCL on ROCm; the same code runs on ``gloo`` for the CPU tests.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)


class GradAllReduce:
    """Average ``.grad`` of ``params`` across ranks (call between backward() and the optimizer)."""

    def __init__(self, params, big_bytes: int = 32 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.big_bytes = big_bytes
        self.group = group
        self._flat = None

    def all_reduce_mean(self) -> None:
        world = dist.get_world_size(self.group)
        grads = [p.grad for p in self.params if p.grad is not None]
        big = [g for g in grads if g.numel() * g.element_size() >= self.big_bytes]
        small = [g for g in grads if g.numel() * g.element_size() < self.big_bytes]
        handles = []
        for g in big:                                   # largest first: it is also produced first by backward
            handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if small:
            n = sum(g.numel() for g in small)
            if self._flat is None or self._flat.numel() != n or self._flat.device != small[0].device:
                self._flat = torch.empty(n, dtype=small[0].dtype, device=small[0].device)
            views = list(torch.split(self._flat, [g.numel() for g in small]))
            torch._foreach_copy_(views, [g.reshape(-1) for g in small])
            handles.append(dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        inv = 1.0 / world
        if small:
            torch._foreach_copy_([g.reshape(-1) for g in small], views)
        torch._foreach_mul_(grads, inv)


def shard_batch(n_global: int, rank: int, world: int) -> slice:
    """Equal contiguous shards of a global batch (n_global must divide by world)."""
    if n_global % world:
        raise ValueError(f"global batch {n_global} does not divide over {world} ranks")
    per = n_global // world
    return slice(rank * per, (rank + 1) * per)
