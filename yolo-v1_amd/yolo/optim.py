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
import os

from . import _hip
from ._hip import AdamTensor, check, lib, ptr, stream


BG_CUS = 128       # CUs the background update of the Linear layers holds (yolo_adam_step_multi_bg): a CU streams ~42 GB/s whatever it keeps in flight,
                   # so the pass runs at ~4.5 TB/s there -- under the conv stack of the next forward, whose persistent kernels draw their tiles from a
                   # queue and so lose only the share of the chip the pass holds.  Step at batch 64 (tools/ab_train.py, one process): 32 CUs 12.16 ms,
                   # 48: 11.20, 64: 11.02-11.10, 96: 10.92, 128: 10.89-10.99, 160: 11.06, 192: 11.17, 256: 11.37.  (Round 2, statically scheduled conv
                   # kernels: 64 was the optimum and 96 no better.)
OVERLAP = os.environ.get("YOLO_ADAM_OVERLAP", "1") != "0"     # attach_plan(overlap=True) takes effect (switch for A/B runs)


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
    grads, extra = _split_known(params, known)
    dev = (grads[0] if grads else extra[0]).device
    if dev.type != "cuda":          # CPU parameters: stock torch ops (an explicit device choice, like the models' CPU path)
        acc = torch.zeros((), dtype=torch.float64)
        for g in grads:
            acc += g.double().pow(2).sum()
        for e in extra:
            acc += e
        return acc
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


def _split_known(params, known):
    """(gradients whose squares must be summed, squared norms taken over from ``known``): an entry of ``known`` counts only while
    the parameter's .grad is still the very memory it was computed from, at the same version"""
    grads, extra = [], []
    for p in params:
        if p.grad is None:
            continue
        k = known.get(id(p)) if known else None
        if k is not None and k[0] == (p.grad.data_ptr(), tuple(p.grad.shape)) and k[1] == p.grad._version:
            extra.append(k[2])
        else:
            grads.append(_f32c(p.grad))
    return grads, extra


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
        # attach_plan(plan, overlap=True): id(param) -> plan.  The update of these parameters (the Linear layers: 76 % of the model's
        # optimizer bytes, first used at the END of the next forward) runs as a background pass on BG_CUS CUs of a second stream
        # (yolo_adam_step_multi_bg) beside the next forward's conv stack; the plan's forward waits for it in front of its first
        # Linear layer.
        self.deferred: dict[int, object] = {}
        self._side = None
        self._pending = None                      # event behind the last background launch
        # device float (or None), consumed by the next step(): non-zero = the producer of the gradients found its input invalid and
        # the step must update nothing.  The training loop hands over YOLOLoss's error word (LossParts.device_flag), which the host
        # reads only after the step was enqueued -- the reference raises inside the loss forward, before any update
        self.skip_if = None
        self._hooked: set = set()                 # ids of the modules that carry this optimizer's state_dict / load_state_dict hooks

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        all_params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not all_params:
            return loss
        if all(not p.is_cuda for p in all_params):
            self._step_on_cpu(all_params)
            return loss
        _hip.require_cuda(*all_params)
        with torch.cuda.device(all_params[0].device):
            return self._step_on_device(all_params, loss)

    def _step_on_cpu(self, all_params):
        """CPU parameters (the reference's ``--device cpu`` runs, the gloo tests of the data-parallel path): the same step in stock
        torch ops -- clip coefficient min(1, max_norm / (|g| + 1e-6)) from the squared norms (incl. the plans' hints), then the
        arithmetic of adam1 in optim.hip, which is torch.optim.Adam's."""
        known = {}
        for plan in self.plans:
            known.update(plan.grad_norm_sq)
            plan.grad_norm_sq.clear()
        clip = 1.0
        if self.max_grad_norm is not None:
            total = float(grad_norm_sq(all_params, known).sqrt())
            clip = min(1.0, self.max_grad_norm / (total + 1e-6))
        skip, self.skip_if = self.skip_if, None
        if skip is not None and float(skip) != 0.0:
            return
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                t = int(state["step"].item())
                g = p.grad.float() * clip + group["weight_decay"] * p
                state["exp_avg"].lerp_(g, 1.0 - b1)
                state["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                denom = state["exp_avg_sq"].sqrt() / ((1.0 - b2 ** t) ** 0.5) + group["eps"]
                p.addcdiv_(state["exp_avg"], denom, value=-group["lr"] / (1.0 - b1 ** t))

    def _step_on_device(self, all_params, loss):
        known = {}
        for plan in self.plans:
            known.update(plan.grad_norm_sq)
            plan.grad_norm_sq.clear()          # one backward pass, one use
        norm = grad_norm_sq(all_params, known) if self.max_grad_norm is not None else None
        skip, self.skip_if = self.skip_if, None
        if skip is not None:
            if not (skip.is_cuda and skip.dtype == torch.float32 and skip.numel() == 1):
                raise ValueError("skip_if must be one float32 on the device")
            _hip.require_cuda(all_params[0], skip)
        st = stream()
        main_t = torch.cuda.current_stream()
        side_used = False
        for group in self.param_groups:
            b1, b2 = group["betas"]
            # one launch per (group, step count, foreground / background): normally one for the conv stack and one for the Linear layers
            by_step: dict[tuple, list] = {}
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
                late = OVERLAP and id(p) in self.deferred
                if late:
                    g.record_stream(self._side_stream())     # read on the second stream after the caller may have dropped it
                by_step.setdefault((int(state["step"].item()), late), []).append(
                    (p, AdamTensor(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                   shadow.data_ptr() if shadow is not None else None, p.numel()), hook))
            for (step, late), items in sorted(by_step.items(), key=lambda kv: kv[0][1]):      # the foreground launch first
                tab = (AdamTensor * len(items))(*[it[1] for it in items])
                if late and len(items) <= 48:
                    side = self._side_stream()
                    if not side_used:
                        side.wait_stream(main_t)             # behind the gradients, the norm and the last forward's / backward's reads
                        if norm is not None:
                            norm.record_stream(side)
                        if skip is not None:
                            skip.record_stream(side)
                        side_used = True
                    check(lib().yolo_adam_step_multi_bg(tab, len(items), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                        float(group["weight_decay"]), step, ptr(norm), float(self.max_grad_norm or 0.0), ptr(skip), BG_CUS,
                                                        ctypes.c_void_p(side.cuda_stream)), "yolo_adam_step_multi_bg")
                else:
                    check(lib().yolo_adam_step_multi(tab, len(items), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                     float(group["weight_decay"]), step, ptr(norm), float(self.max_grad_norm or 0.0), ptr(skip), st), "yolo_adam_step_multi")
                for p, _, hook in items:
                    # the kernel updated p through a raw pointer: bump the autograd version so that the
                    # engine's packed bf16 copies notice (no memory traffic) ...
                    torch.autograd.graph.increment_version(p)
                    if hook is not None and hook[1] is not None:
                        hook[1](p)         # ... and tell the owner of a shadow that it is already current
        if side_used:
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._pending = ev
            for plan in {id(pl): pl for pl in self.deferred.values()}.values():
                plan.params_ready.event = ev            # Plan.forward waits for it in front of its first Linear layer
        return loss

    def _side_stream(self):
        if self._side is None:
            self._side = _hip.side_stream(torch.device("cuda", torch.cuda.current_device()), low=False)
        return self._side

    def synchronize(self) -> None:
        """Make the CURRENT stream wait for a background update still running on the second stream (attach_plan(overlap=True)).
        The attached plan's forward does this by itself in front of its Linear layers; call it before reading those layers'
        parameters or the optimizer state in any other way.  Waiting by themselves: this optimizer's state_dict() / load_state_dict() /
        zero_grad(set_to_none=False), and the owning model's state_dict() / load_state_dict() / copy.deepcopy (hooks set by attach_plan)."""
        if self._pending is not None:
            torch.cuda.current_stream().wait_event(self._pending)
            self._pending = None
            for plan in self.deferred.values():
                plan.params_ready.event = None

    def _hook_owner(self, plan) -> None:
        """the module that owns ``plan`` waits for a background update by itself wherever torch reads or writes its parameters in
        bulk: ``state_dict()`` (checkpoints, ``torch.save(model.state_dict())``), ``load_state_dict()`` (resume) and
        ``copy.deepcopy`` (EMA copies; models.YOLOv1.__deepcopy__ asks the plan) -- no caller has to know about the second stream."""
        owner = plan.owner() if getattr(plan, "owner", None) is not None else None
        if owner is None or id(owner) in self._hooked:
            return
        self._hooked.add(id(owner))
        import weakref
        me = weakref.ref(self)

        def wait(*_a, **_k):
            opt = me()
            if opt is not None:
                opt.synchronize()
        owner.register_state_dict_pre_hook(wait)
        owner.register_load_state_dict_pre_hook(wait)

    def state_dict(self):
        self.synchronize()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self.synchronize()
        return super().load_state_dict(state_dict)

    def zero_grad(self, set_to_none: bool = True):
        if not set_to_none:
            self.synchronize()             # zeroing in place would race with the background pass, which still reads the gradients
        return super().zero_grad(set_to_none=set_to_none)

    def attach_plan(self, plan, overlap: bool = False) -> None:
        """Let this optimizer refresh the engine's bf16 forward operands of Linear layers in the same
        pass that updates their fp32 masters (``plan``: ``model.hip_plan()``).

        ``overlap``: update the Linear layers as a background pass on a second stream, beside the next forward's conv stack (same
        floats; see ``synchronize`` for what then has to wait).  Worth 0.15 ms of the 11.9-ms YOLOv1 step (train.py, bench.py); behind the HBM-heavy ResNet trunk it loses."""
        for p, shadow, fresh in plan.bf16_shadows():
            self.bf16_shadow[id(p)] = (shadow, fresh)
            if overlap:
                self.deferred[id(p)] = plan
        if overlap:
            for b in plan.fc_biases():
                self.deferred[id(b)] = plan
            self._hook_owner(plan)
        if all(pl is not plan for pl in self.plans):
            self.plans.append(plan)
