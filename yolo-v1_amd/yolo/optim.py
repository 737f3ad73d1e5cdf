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

import ctypes

from . import _hip
from ._hip import AdamTensor, check, lib, ptr, stream


def _f32c(g: torch.Tensor) -> torch.Tensor:
    return g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()


def grad_norm_sq(params, known=None) -> torch.Tensor:
    """device double holding sum over all gradients of g^2 (enqueued, not synchronised): one launch
    for the whole list (yolo_sumsq_f32_multi).

    ``known``: {id(param): ((data_ptr, shape) of the gradient, its version counter, device double)} -- squared norms the producer of a gradient
    already has (engine.Plan.backward: yolo_wgrad sums the squares of the 205 M-element gradient of the Linear behind
    nn.Flatten while it stores it, yolo_wgrad_desc.dw_sumsq: 822 MB less to read).  An entry is used only while the parameter's
    .grad is still that very memory, unmodified (autograd hands over a detached alias that shares the version counter; accumulation,
    an all-reduce or clipping in place bump it)."""
    grads, extra = [], []
    for p in params:
        if p.grad is None:
            continue
        k = known.get(id(p)) if known else None
        if k is not None and k[0] == (p.grad.data_ptr(), tuple(p.grad.shape)) and k[1] == p.grad._version:
            extra.append(k[2])
        else:
            grads.append(_f32c(p.grad))
    dev = (grads[0] if grads else extra[0]).device
    _hip.require_cuda(*grads)
    with torch.cuda.device(dev):
        acc = torch.zeros((), dtype=torch.float64, device=dev)
        if grads:
            gp = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            gn = (ctypes.c_long * len(grads))(*[g.numel() for g in grads])
            check(lib().yolo_sumsq_f32_multi(gp, gn, len(grads), ptr(acc), stream()), "yolo_sumsq_f32_multi")
        for e in extra:
            acc += e
    return acc


def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """HIP version of torch.nn.utils.clip_grad_norm_ (L2): returns the total norm as a device tensor."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return torch.zeros(())
    acc = grad_norm_sq(params)
    with torch.cuda.device(acc.device):
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
        self.bf16_shadow: dict[int, tuple] = {}   # id(param) -> (bf16 tensor refreshed in the same pass, callback(param) | None)
        self.plans: list = []                     # attached engine plans: their backward passes leave squared gradient norms (grad_norm_sq)

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
        with torch.cuda.device(all_params[0].device):
            return self._step_on_device(all_params, loss)

    def _step_on_device(self, all_params, loss):
        known = {}
        for plan in self.plans:
            known.update(plan.grad_norm_sq)
            plan.grad_norm_sq.clear()          # one backward pass, one use
        norm = grad_norm_sq(all_params, known) if self.max_grad_norm is not None else None
        st = stream()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            # one launch per (group, step count): normally ONE launch for the whole model
            by_step: dict[int, list] = {}
            keep = []                      # temporaries the launch reads must outlive the enqueue
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("yolo.optim.Adam needs contiguous fp32 parameters")
                g = _f32c(p.grad)
                keep.append(g)
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                hook = self.bf16_shadow.get(id(p))
                shadow = hook[0] if hook is not None else None
                by_step.setdefault(int(state["step"].item()), []).append(
                    (p, AdamTensor(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                   shadow.data_ptr() if shadow is not None else None, p.numel()), hook))
            for step, items in by_step.items():
                tab = (AdamTensor * len(items))(*[it[1] for it in items])
                check(lib().yolo_adam_step_multi(tab, len(items), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                 float(group["weight_decay"]), step, ptr(norm), float(self.max_grad_norm or 0.0), st), "yolo_adam_step_multi")
                for p, _, hook in items:
                    # the kernel updated p through a raw pointer: bump the autograd version so that the
                    # engine's packed bf16 copies notice (no memory traffic) ...
                    torch.autograd.graph.increment_version(p)
                    if hook is not None and hook[1] is not None:
                        hook[1](p)         # ... and tell the owner of a shadow that it is already current
        return loss

    def attach_plan(self, plan) -> None:
        """Let this optimizer refresh the engine's bf16 forward operands of Linear layers in the same
        pass that updates their fp32 masters (``plan``: ``model.hip_plan()``)."""
        for p, shadow, fresh in plan.bf16_shadows():
            self.bf16_shadow[id(p)] = (shadow, fresh)
        if all(pl is not plan for pl in self.plans):
            self.plans.append(plan)
