"""Optimizer step on the HIP kernels (optim.hip): Adam + global-norm gradient clipping in one HBM pass.

``Adam`` takes the constructor arguments of ``torch.optim.Adam`` (the reference builds
``optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4)``, src/train.py:177-179) and keeps the
same per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``), so optimizer ``state_dict``s are
interchangeable with the reference's checkpoints.  ``max_grad_norm`` folds
``clip_grad_norm_(params, max_norm)`` (trainer.py:79,93) into the same pass: the global norm is
reduced on the device and read by the update kernel, no host sync.
"""

from __future__ import annotations

import torch

from . import _hip
from ._hip import check, lib, ptr, stream


def grad_norm_sq(params) -> torch.Tensor:
    """device double holding sum over all gradients of g^2 (enqueued, not synchronised)."""
    grads = [p.grad for p in params if p.grad is not None]
    acc = torch.zeros((), dtype=torch.float64, device=grads[0].device)
    st = stream()
    for g in grads:
        g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        check(lib().yolo_sumsq_f32(ptr(g), g.numel(), ptr(acc), st), "yolo_sumsq_f32")
    return acc


def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """HIP version of torch.nn.utils.clip_grad_norm_ (L2): returns the total norm as a device tensor."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return torch.zeros(())
    acc = grad_norm_sq(params)
    st = stream()
    for p in params:
        check(lib().yolo_clip_scale_f32(ptr(p.grad), p.grad.numel(), ptr(acc), float(max_norm), st), "yolo_clip_scale_f32")
    return acc.sqrt().float()


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (amsgrad=False, L2 weight decay) on yolo_adam_step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm: float | None = None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.bf16_shadow: dict[int, torch.Tensor] = {}   # id(param) -> bf16 tensor to refresh in the same pass

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        all_params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not all_params:
            return loss
        _hip.require_cuda(*all_params)
        norm = grad_norm_sq(all_params) if self.max_grad_norm is not None else None
        st = stream()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                shadow = self.bf16_shadow.get(id(p))
                check(lib().yolo_adam_step(ptr(p), ptr(g), ptr(state["exp_avg"]), ptr(state["exp_avg_sq"]), p.numel(),
                                           float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                           int(state["step"].item()), ptr(norm), float(self.max_grad_norm or 0.0), ptr(shadow), st), "yolo_adam_step")
                # the kernel updated p through a raw pointer: bump the autograd version so that the
                # engine's packed bf16 copies notice (no memory traffic)
                torch.autograd.graph.increment_version(p)
        return loss
