"""YOLOLoss with the reference's surface (src/yolo/loss.py:7-212) on top of the fused HIP kernel.

Device tensors: forward AND backward come from one launch of ``yolo_loss_fwd_bwd`` (loss.hip) and
the five ``.item()`` syncs of the reference (loss.py:165-169) become one 32-byte copy, made by a side
stream and awaited when the components are first read (LossParts).
CPU tensors: the same formula in stock torch ops -- the reference's ``--device cpu`` behaviour; it is
an explicit device choice, never a fallback for a missing HIP library.
"""

from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

_KEYS = ("total", "coord", "conf_obj", "conf_noobj", "class")


class _HipLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, tgt, S, B, C, lc, ln):
        out, dpred = ops.loss_fwd_bwd(pred, tgt, S, B, C, lc, ln, want_grad=ctx.needs_input_grad[0])
        ctx.dpred = dpred
        ctx.in_dtype = pred.dtype
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_out):
        d = ctx.dpred
        ctx.dpred = None
        if d is None:
            return (None,) * 7
        return (d * g_total).to(ctx.in_dtype), None, None, None, None, None, None


_BAD_SLOT = ("index out of bounds: a target cell selects a box slot >= B "
             "(targets[..., 4::5] also covers class channels; reference gather raises here)")
_COPY_STREAMS: dict = {}


class LossParts(dict):
    """The five loss components as Python floats -- the dict the reference builds with five ``.item()`` calls (loss.py:165-169) --
    fetched from the device when first READ.  The 32-byte result is copied to pinned host memory by a side stream right behind the
    loss kernel; reading waits for that copy only.  A training loop that looks at the components after ``optimizer.step()`` (the
    reference's loop does, trainer.py:84-90) therefore never stalls the host in the middle of a step with nothing queued behind the
    loss: 12.32 -> 11.86 ms per step at batch 64 (tools/experiments/loss_sync_cost.py).  The reference's IndexError for a target
    that selects a box slot >= B surfaces as RuntimeError at that first read (or at the next loss call, if the dict is never read);
    ``device_flag`` lets the optimizer skip the update of such a step on the device."""

    def __init__(self, event, host, device_flag=None):
        super().__init__((k, None) for k in _KEYS)
        self._pending = (event, host)
        # the kernel's error word on the device (one float32, non-zero = a target selected a box slot >= B): an optimizer that takes it
        # (yolo.optim.Adam.skip_if; training.train_epoch passes it) updates nothing in a flagged step, so that the error raised at the
        # first read finds the parameters as the reference's IndexError would have left them
        self.device_flag = device_flag

    def _fetch(self):
        if self._pending is not None:
            event, host = self._pending
            self._pending = None
            event.synchronize()
            vals = host.tolist()
            if vals[5] != 0.0:
                raise RuntimeError(_BAD_SLOT)
            dict.update(self, zip(_KEYS, vals[:5]))
        return self

    def done(self) -> bool:
        """values are on the host (or were already read)"""
        return self._pending is None or self._pending[0].query()

    def __getitem__(self, k):
        return dict.__getitem__(self._fetch(), k)

    def __iter__(self):          # (also keeps dict(parts) / {**parts} off CPython's raw-storage fast path, which would copy the placeholders)
        return dict.__iter__(self._fetch())

    def keys(self):
        return dict.keys(self._fetch())

    def get(self, k, default=None):
        return dict.get(self._fetch(), k, default)

    def items(self):
        return dict.items(self._fetch())

    def values(self):
        return dict.values(self._fetch())

    def copy(self):
        return dict(dict.items(self._fetch()))

    def __eq__(self, other):
        return dict.__eq__(self._fetch(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self._fetch())

    def __reduce__(self):
        return (dict, (self.copy(),))


class YOLOLoss(nn.Module):
    """Sum-squared YOLOv1 loss; see the reference docstring for the five components."""

    def __init__(self, S: int = 7, B: int = 2, C: int = 20, lambda_coord: float = 5.0, lambda_noobj: float = 0.5):
        super().__init__()
        self.S, self.B, self.C = S, B, C
        self.lambda_coord = lambda_coord
        self.lambda_noobj = lambda_noobj
        self.eager_parts = False      # True: copy the components to the host inside forward() (the reference's timing of the IndexError)
        self._last_parts = None

    def forward(self, predictions: torch.Tensor, targets: torch.Tensor) -> tuple[torch.Tensor, dict[str, float]]:
        if predictions.is_cuda:
            prev, self._last_parts = self._last_parts, None
            if prev is not None and prev._pending is not None and prev.done():
                prev._fetch()          # a dict nobody read: its error flag must not get lost (no wait: the copy has landed)
            total, out = _HipLossFn.apply(predictions, targets, self.S, self.B, self.C, float(self.lambda_coord), float(self.lambda_noobj))
            if self.eager_parts:
                host = out.tolist()
                if host[5] != 0.0:
                    raise RuntimeError(_BAD_SLOT)
                return total, dict(zip(_KEYS, host[:5]))
            dev = out.device
            cs = _COPY_STREAMS.get(dev.index)
            if cs is None:
                cs = _COPY_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
            cs.wait_stream(torch.cuda.current_stream(dev))          # behind the loss kernel; nothing of the backward pass is queued yet
            host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
            with torch.cuda.stream(cs):
                host.copy_(out, non_blocking=True)
                event = torch.cuda.Event()
                event.record(cs)
            out.record_stream(cs)
            parts = LossParts(event, host, out[5:6])
            self._last_parts = parts
            return total, parts
        return self._forward_cpu(predictions, targets)

    # ---- stock-torch formulation for CPU tensors (reference semantics)
    def _forward_cpu(self, pred: torch.Tensor, tgt: torch.Tensor):
        N, S, B = pred.size(0), self.S, self.B
        pb = pred[..., : B * 5].reshape(N, S, S, B, 5)
        tb = tgt[..., : B * 5].reshape(N, S, S, B, 5)
        slot_mask = tgt[..., 4::5] > 0                       # runs over ALL channels, like the reference
        obj = slot_mask.any(-1)
        sel = slot_mask.float().argmax(-1)
        if bool((sel[obj] >= B).any()):
            raise RuntimeError("index out of bounds: a target cell selects a box slot >= B")
        sel = sel.clamp(max=B - 1)
        tbox = tb[..., :4].gather(3, sel[..., None, None].expand(-1, -1, -1, 1, 4)).squeeze(3)
        ious = self.compute_iou(pb[..., :4], tbox.unsqueeze(3))
        best = ious.argmax(3, keepdim=True)
        resp = torch.zeros_like(ious, dtype=torch.bool).scatter_(3, best, True) & obj.unsqueeze(-1)
        respf = resp.to(pred.dtype)
        objf = obj.to(pred.dtype)
        zero = pred.new_zeros(())
        if bool(obj.any()):
            t4 = tbox.unsqueeze(3)
            xy = ((pb[..., :2] - t4[..., :2]) ** 2).sum(-1)
            wh = ((pb[..., 2:4].clamp(min=1e-6).sqrt() - t4[..., 2:4].clamp(min=1e-6).sqrt()) ** 2).sum(-1)
            coord = self.lambda_coord * ((xy + wh) * respf).sum()
            conf_obj = (((pb[..., 4] - ious) ** 2) * respf).sum()
            cls = (((pred[..., B * 5:] - tgt[..., B * 5:]) ** 2).sum(-1) * objf).sum()
        else:
            coord = conf_obj = cls = zero
        conf_noobj = self.lambda_noobj * ((pb[..., 4] ** 2) * (1 - respf)).sum()
        total = (coord + conf_obj + conf_noobj + cls) / N
        d = {"total": total.detach().item(), "coord": (coord / N).detach().item(), "conf_obj": (conf_obj / N).detach().item(),
             "conf_noobj": (conf_noobj / N).detach().item(), "class": (cls / N).detach().item()}
        return total, d

    @staticmethod
    def compute_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
        """IoU of centre-format boxes, broadcasting (..., B, 4) against (..., 1, 4) -> (..., B)."""
        if boxes1.is_cuda and not (boxes1.requires_grad or boxes2.requires_grad):
            return ops.loss_iou(boxes1, boxes2)
        a1, a2 = boxes1[..., :2] - boxes1[..., 2:4] / 2, boxes1[..., :2] + boxes1[..., 2:4] / 2
        b1, b2 = boxes2[..., :2] - boxes2[..., 2:4] / 2, boxes2[..., :2] + boxes2[..., 2:4] / 2
        wh = (torch.minimum(a2, b2) - torch.maximum(a1, b1)).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        union = boxes1[..., 2] * boxes1[..., 3] + boxes2[..., 2] * boxes2[..., 3] - inter
        return inter / (union + 1e-6)
