"""autograd bridges: a whole plan (or the whole trainable ResNet trunk) is ONE autograd node -- no per-layer graph."""

from __future__ import annotations

import torch

from . import _hip
from .executor import Plan

class PlanFunction(torch.autograd.Function):
    """autograd bridge: forward/backward of a whole plan as ONE node (no per-layer autograd graph)."""

    @staticmethod
    def forward(ctx, plan: Plan, drop_training: bool, need_grad: bool, x: torch.Tensor, *params):
        out, saved = plan.forward(x, need_grad, drop_training)
        ctx.plan = plan
        ctx.saved = saved
        ctx.x_needs = x.requires_grad
        return out

    @staticmethod
    @_hip.device_guard
    def backward(ctx, gout):
        if ctx.saved is None:
            raise RuntimeError("backward through a plan that ran without grad")
        gx, pg = ctx.plan.backward(ctx.saved, gout, ctx.x_needs)
        ctx.saved = None
        return (None, None, None, gx, *pg)


class ResNetTrainFunction(torch.autograd.Function):
    """autograd bridge of the trainable ResNet trunk: one node for the whole trunk (ResNetPlan.forward_train / backward_train)."""

    @staticmethod
    def forward(ctx, plan, frozen: bool, x: torch.Tensor, *params):
        out, saved = plan.forward_train(x, frozen)
        ctx.plan, ctx.saved, ctx.params = plan, saved, params
        return out

    @staticmethod
    @_hip.device_guard
    def backward(ctx, gout):
        if ctx.saved is None:
            raise RuntimeError("backward through a ResNet trunk forward that was already consumed")
        grads = ctx.plan.backward_train(ctx.saved, gout)
        ctx.saved = None
        return (None, None, None) + tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)


@_hip.device_guard
def run_plan(plan: Plan, x: torch.Tensor, drop_training: bool) -> torch.Tensor:
    _hip.require_cuda(x)
    # grad mode must be sampled here: inside Function.forward it is always off
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in plan.params))
    return PlanFunction.apply(plan, drop_training, need, x, *plan.params)

