"""Switches and measurement hooks of the engine (round 3: one object instead of module-level globals of engine.py).

``CONFIG`` is the process-wide default; an executor (``executor.Plan``, ``resnet_executor.ResNetPlan``) reads ``plan.cfg`` if it was given its own
``EngineConfig`` (``plan.cfg = dataclasses.replace(CONFIG, WGRAD_STREAM=False)``) and ``CONFIG`` otherwise, so a measurement can pin the schedule of ONE
plan without touching the others.  The measurement hooks (``TIMERS``) never change the schedule: round 2's ``TIMERS is not None`` silently
switched the second stream of the backward pass off; now the per-launch events are recorded on whichever stream a launch runs on, and whoever wants
every launch alone on the chip says so (``WGRAD_STREAM = False``, bench.py's "each launch alone" figures).  ``yolo.engine`` forwards attribute reads and
writes of these names to ``CONFIG`` (``engine.FUSE_POOL = False`` keeps working)."""

from __future__ import annotations

from dataclasses import dataclass, fields


@dataclass
class EngineConfig:
    # ---- measurement hooks (do not change what runs)
    TIMERS: list | None = None       # a list: every MFMA / pool launch is bracketed by events ON ITS LAUNCH STREAM, (tag, kernel, flops, e0, e1) appended
    IGEMM_LAUNCHES: int = 0          # yolo_igemm launches so far (bench.py: launches per step of the dominant kernel)
    TUNE_LOG: list | None = None     # tools/tune_plans.py: (key, winner, six fastest candidates with their times)
    # ---- launch plans
    AUTOTUNE: bool = False           # tools/tune_plans.py only: time the candidate plans of a problem without a table entry
    TILE_HINT: int = 0               # tests / tuning: force a tile configuration of yolo_igemm (0 = plan table; yolo_igemm_desc.tile_hint)
    TILE_PX: int = 0                 # with TILE_HINT: yolo_igemm_desc.tile_px of the forced configuration
    PLAN_TABLE: bool = True          # read at import: load the measured plans of yolo/plans/gfx950.json (False, i.e. YOLO_AMD_PLAN_TABLE=0: every problem takes the deterministic
                                     # default plan, whose fp32 summation order does not depend on the batch size unless SMALL_SPLIT splits K ranges)
    BORROW_PLANS: bool = True        # a problem of a batch size without measured plans (a DataLoader's ragged last batch, a user's own batch size) runs the plan measured
                                     # for the same layer at the nearest measured batch size (plans._borrowed_plan) instead of the default rule
    SMALL_SPLIT: bool = True         # problems without a table entry, < 2048 pixels under a deep K (small batches on the 14x14 / 7x7 maps): K ranges as slabs (False: one plain launch)
    PERSIST: bool = True             # plans with the persistent kernels (tile_hint 20 / 21) run them (False: the pipelined kernels 15 / 16 -- A/B runs)
    # ---- what is fused
    BN_STATS_IN_CONV: bool = True    # ResNet trunk in batch-statistics mode: BatchNorm's sums come out of the conv's epilogue (yolo_igemm_desc.bn_stats)
    STEM_KERNEL: bool = True         # 7x7/s2 stem through yolo_conv_stem7_fwd (False: the generic row-segment implicit GEMM; tests compare)
    STEM_POOL_BWD_FUSED: bool = True # backward of the pool + LeakyReLU behind the stem inside yolo_wgrad_stem7_pooled (False: separate pass)
    STRIDE2_CLASSES: bool = True     # data gradient of a stride-2 3x3 conv as four parity-class convs over the non-zero gradient slots
    STEM_F32_INPUT: bool = True      # inference: the stem kernel reads the NCHW fp32 input itself (no separate layout pass)
    FLATTEN_FREE: bool = True        # inference: conv -> nn.Flatten -> Linear without the flatten pass (dense NHWC conv output + K-permuted weight panels)
    POOL_CODES: bool = True          # training: a fused conv + pool stores the pooled map and 2-bit arg-max codes, not the un-pooled activation
    FUSE_POOL: bool = True           # inference: fold MaxPool2d(2,2) into the preceding conv's epilogue where the geometry allows
    FC_NORM_IN_WGRAD: int = 1 << 26  # Linear layers with at least this many weights: yolo_wgrad also sums the squares of the gradient it stores
    # ---- the schedule of the backward pass
    WGRAD_STREAM: bool = True        # weight gradients run on a second HIP stream beside the data-gradient chain (they are off its critical path)
    SIDE_LOW: bool = True            # ... of the lowest scheduling priority: the dispatcher prefers the data-gradient chain (12.73 -> 12.55 ms per step)
    WGRAD_SLABS: bool = False        # pipelined weight-gradient kernel: partial tiles as slabs summed in fixed order instead of fp32 atomics (bit-reproducible,
                                     # 13-16 % faster per launch alone, slower inside the two-stream step: 12.06 -> 12.28 ms)
    WGRAD_PIPE: bool = True          # weight gradient of the big deep 3x3 layers through the 256 x 256 pipelined kernels (yolo_wgrad_desc.variant = 5 / 6)
    WGRAD_CHOICE: object = None      # dict {(N, Hout, Wout, Cout, Cin, K, stride): (variant, flat)}: overrides the shape rule of Plan._wgrad_desc (tools/search_wgrad.py; the
                                     # shipped choices live in yolo/plans/gfx950.json under "wgrad")
    PLAN_TIMES: object = None        # dict: igemm_call records (start, end) events per problem key -- in-situ time of the plan in use (tools/merge_plans.py)
    TUNE_REPS: int = 1               # tuner: launches per timed sample (1: a single launch behind a cache flush; > 1: back-to-back launches, warm caches -- the regime inside a forward pass)
    FC_WGRAD_SIDE: bool = True       # the Linear layers' weight gradients on the second stream too: the 822 MB store of the big one runs beside the 411 MB read of its data gradient
    WGRAD_WIDE: bool = True          # ... variant 6 (wgrad_wide.hip: four waves of 128 x 128, accumulators in AGPRs) instead of 5, and on more layers


def _from_env(cfg: EngineConfig) -> EngineConfig:
    """YOLO_AMD_<SWITCH>=0|1|<int> in the environment sets a bool / int switch of the process-wide default (A/B runs of entry points and of
    the multi-process tests without editing code); anything else in such a variable is an error, not ignored"""
    import os
    for f in fields(EngineConfig):
        v = os.environ.get("YOLO_AMD_" + f.name)
        if v is None:
            continue
        if f.type not in ("bool", "int"):
            raise ValueError(f"YOLO_AMD_{f.name}: only the bool / int switches can be set from the environment")
        setattr(cfg, f.name, bool(int(v)) if f.type == "bool" else int(v))
    return cfg


CONFIG = _from_env(EngineConfig())
SWITCHES = frozenset(f.name for f in fields(EngineConfig))
