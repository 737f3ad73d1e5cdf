"""Facade of the engine (round 3: the 2,000-line engine.py of round 2 is now five modules).

    config.py           EngineConfig -- every switch and measurement hook as ONE object (process-wide ``CONFIG``, per plan ``plan.cfg``)
    runtime.py          workspaces + streams: Act, scratch pools, side streams, per-launch timers, the patchable hooks ``RT``
    plans.py            launch plans of yolo_igemm: shipped table, deterministic default, tuner, ``igemm_call``
    executor.py         Layer / Plan: the YOLOv1 conv / pool / FC executor (forward, backward on two streams)
    resnet_executor.py  ResNetPlan: ResNet-50 trunk (inference, frozen-training, trainable forward / backward)
    autograd.py         PlanFunction, ResNetTrainFunction, run_plan

This module re-exports their names, and reads / writes of a switch (``engine.FUSE_POOL = False``, ``engine.TIMERS = []``) or of a runtime hook
(``engine.lib = fake``) go to ``config.CONFIG`` / ``runtime.RT``, so code written against round 2's module-level globals keeps working."""

from __future__ import annotations

import sys
import types

from . import autograd as _autograd
from . import config as _config
from . import executor as _executor
from . import plans as _plans
from . import resnet_executor as _resnet_executor
from . import runtime as _runtime
from .autograd import PlanFunction, ResNetTrainFunction, run_plan  # noqa: F401
from .config import CONFIG, EngineConfig  # noqa: F401
from .executor import Layer, Plan  # noqa: F401
from .plans import (PLAN_FILE, _TILE, _TILE_COST, _TUNED, _default_plan, _key_str, _persist_ok, _pipe_ok, _pipe_pool_ok, _run_plan_igemm,  # noqa: F401
                    _tune, _tune_key, igemm_call, load_plans, save_plans)
from .resnet_executor import ResNetPlan  # noqa: F401
from .runtime import RT, Act, _attach_wgrad_slabs, _EventSlot, _igemm, _on_side_stream, _round_up, _Streams, _timed  # noqa: F401


class _EngineModule(types.ModuleType):
    """attribute access of the switch names goes to config.CONFIG, of the runtime hooks to runtime.RT"""

    def __getattr__(self, name):          # only reached for names the module itself does not define
        if name in _config.SWITCHES:
            return getattr(_config.CONFIG, name)
        if name in _runtime.HOOKS:
            return getattr(_runtime.RT, name)
        raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in _config.SWITCHES:
            setattr(_config.CONFIG, name, value)
        elif name in _runtime.HOOKS:
            setattr(_runtime.RT, name, value)
        else:
            super().__setattr__(name, value)

    def __delattr__(self, name):          # (pytest's monkeypatch restores with setattr; a stray delattr of a forwarded name is a no-op)
        if name in _config.SWITCHES or name in _runtime.HOOKS:
            return
        super().__delattr__(name)


sys.modules[__name__].__class__ = _EngineModule
