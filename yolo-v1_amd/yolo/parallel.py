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


class OverlappedGradAllReduce:
    """Gradient averaging that runs WHILE backward is still producing gradients.

    Works on an engine.Plan with a gradient arena (``plan.attach_grad_arena``): gradients land in one flat
    buffer in the order backward produces them, so finished layers form a contiguous, growing prefix.  Every
    time ``bucket_bytes`` more are final (the 822 MB FC1 gradient is the very first bucket) an asynchronous
    all-reduce of that slice is enqueued; RCCL runs it on its own stream behind the kernels already queued
    and beside the rest of the conv backward.  ``finish()`` (call it between ``backward()`` and the
    optimizer) reduces the tail + the bias region and waits.  xGMI is point-to-point, so buckets are
    large (default 64 MB) and few."""

    def __init__(self, plan, device, bucket_bytes: int = 64 << 20, group=None, stream_id=None):
        self.plan = plan
        self.group = group
        self.bucket = max(1, bucket_bytes // 4)
        self.arena = plan.arena if plan.arena is not None else plan.attach_grad_arena(device)
        plan.on_grad_ready = self._ready
        plan.on_backward_done = self._backward_done
        plan.on_stream_wait = self._stream_wait
        self._sent = 0
        self._final = 0
        self._handles = []
        self._avg = dist.get_backend(group) == "nccl"     # RCCL has ReduceOp.AVG; gloo does not
        # Stream bookkeeping.  A collective is ordered behind the work of the stream that is CURRENT when it is enqueued, and the plan's
        # backward produces gradients on two streams (weight gradients of the conv layers on a low-priority side stream).  Every
        # announced range remembers its producing stream and a tick; every stream wait the plan performs is noted with its tick; a
        # bucket may only be enqueued from a stream that produced each of its pieces or has waited for the producer since.
        self._stream_id = stream_id or (lambda: torch.cuda.current_stream(self.arena.device).cuda_stream if self.arena.is_cuda else 0)
        self._tick = 0
        self._pieces: list[tuple] = []        # (lo, hi, producing stream, tick) of the ranges announced in this backward pass
        self._waits: dict[tuple, int] = {}    # (waiter, waited) -> tick of the last wait
        self.log: list | None = None          # tests: (lo, hi, stream at the call, pieces) per enqueued bucket

    def _stream_wait(self, waiter, waited):
        self._tick += 1
        self._waits[(waiter, waited)] = self._tick

    def _check_ordered(self, lo: int, hi: int, cur):
        for (a, b, s, t) in self._pieces:
            if a < hi and b > lo and s != cur and self._waits.get((cur, s), -1) < t:
                raise RuntimeError(f"gradient bucket [{lo}, {hi}) is being all-reduced from stream {cur:#x}, but its range [{a}, {b}) was "
                                   f"produced on stream {s:#x}, which that stream has not waited for since: the collective could read an "
                                   "unfinished gradient")

    def _reduce(self, lo: int, hi: int):
        if hi > lo:
            cur = self._stream_id()
            self._check_ordered(lo, hi, cur)
            if self.log is not None:
                self.log.append((lo, hi, cur, [p for p in self._pieces if p[0] < hi and p[1] > lo]))
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self._handles.append(dist.all_reduce(self.arena[lo:hi], op=op, group=self.group, async_op=True))

    def _ready(self, lo: int, hi: int):
        # layers complete in arena order, so [0, hi) is final; the caller is on the stream that produced [lo, hi)
        self._tick += 1
        self._pieces.append((lo, hi, self._stream_id(), self._tick))
        self._final = max(self._final, hi)
        if self._final - self._sent >= self.bucket:
            self._reduce(self._sent, self._final)
            self._sent = self._final

    def _backward_done(self):
        # remaining weights + the whole bias region (bias gradients are accumulated by every layer's kernel, on either stream: the
        # plan calls this after its main stream has waited for the side stream, which _check_ordered verifies for the weight ranges)
        self._reduce(self._sent, self.arena.numel())
        self._sent = self.arena.numel()

    def finish(self) -> None:
        for h in self._handles:
            h.wait()
        self._handles = []
        if not self._avg:
            self.arena.mul_(1.0 / dist.get_world_size(self.group))
        # The plan's backward left |g|^2 of the big Linear gradient as computed by the kernel that stored it -- the LOCAL gradient's.
        # A collective does not bump the tensor's version counter (RCCL AVG works in place, no mul_), so the optimizer's
        # "same memory, same version" test would take the stale per-rank norm for the clip and the replicas would drift apart.
        # Drop the hint and bump the version: clip_grad_norm_ re-reads the averaged gradient.
        self.plan.grad_norm_sq.clear()
        torch.autograd.graph.increment_version(self.arena)
        self._sent = 0
        self._final = 0
        self._pieces.clear()
        self._waits.clear()

    # same entry point as GradAllReduce, so training loops can use either
    all_reduce_mean = finish


class _Both:
    """two reducers behind one ``all_reduce_mean`` (the overlapped one first: its collectives are already in flight)"""

    def __init__(self, *reducers):
        self.reducers = reducers

    def all_reduce_mean(self) -> None:
        for r in self.reducers:
            r.all_reduce_mean()


def make_grad_reducer(model: torch.nn.Module, device, group=None):
    """The gradient averaging a training loop should use for ``model`` (call once, after the model is on its device):

    * a fused YOLOv1 (``model._fusable()``) on a GPU: the whole network is one engine plan -> gradient arena +
      ``OverlappedGradAllReduce`` (all-reduce of finished layers while backward continues, FC1's 822 MB first);
    * ``DetectionHead`` on a ResNet trunk on a GPU: the head's plan overlapped, the trunk's parameters (which autograd
      hands over after the head) with ``GradAllReduce``;
    * anything else (CPU tensors, custom modules): ``GradAllReduce`` over all parameters.

    The arena path OVERWRITES gradients every backward (no accumulation across backward calls), which is what the
    reference's loop does (zero_grad before every backward, trainer.py:64)."""
    on_gpu = torch.device(device).type == "cuda"
    if on_gpu and hasattr(model, "_fusable") and model._fusable():
        return OverlappedGradAllReduce(model.hip_plan(), device, group=group)
    head = getattr(model, "head", None)
    if on_gpu and head is not None and hasattr(head, "hip_plan"):
        head_ids = {id(p) for p in head.parameters()}
        rest = [p for p in model.parameters() if id(p) not in head_ids]
        return _Both(OverlappedGradAllReduce(head.hip_plan(), device, group=group), GradAllReduce(rest, group=group))
    return GradAllReduce(model.parameters(), group=group)
