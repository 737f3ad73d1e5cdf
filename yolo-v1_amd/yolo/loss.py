"""YOLOLoss with the reference's surface (src/yolo/loss.py:7-212) on top of the fused HIP kernel.

Device tensors: forward AND backward come from one launch of ``yolo_loss_fwd_bwd`` (loss.hip) and
the five ``.item()`` syncs of the reference (loss.py:165-169) become one 32-byte copy.
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


class YOLOLoss(nn.Module):
    """Sum-squared YOLOv1 loss; see the reference docstring for the five components."""

    def __init__(self, S: int = 7, B: int = 2, C: int = 20, lambda_coord: float = 5.0, lambda_noobj: float = 0.5):
        super().__init__()
        self.S, self.B, self.C = S, B, C
        self.lambda_coord = lambda_coord
        self.lambda_noobj = lambda_noobj

    def forward(self, predictions: torch.Tensor, targets: torch.Tensor) -> tuple[torch.Tensor, dict[str, float]]:
        if predictions.is_cuda:
            total, out = _HipLossFn.apply(predictions, targets, self.S, self.B, self.C, float(self.lambda_coord), float(self.lambda_noobj))
            host = out.tolist()  # the ONE device->host sync of a training step's loss
            if host[5] != 0.0:
                raise RuntimeError("index out of bounds: a target cell selects a box slot >= B "
                                   "(targets[..., 4::5] also covers class channels; reference gather raises here)")
            return total, dict(zip(_KEYS, host[:5]))
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
