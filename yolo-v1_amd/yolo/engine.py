"""Layer-plan executor: runs a conv / pool / FC stack of the reference's models on the HIP kernels.

A *plan* is built once from the PyTorch-layout modules (``nn.Conv2d``/``nn.LeakyReLU``/``nn.MaxPool2d``
inside ``backbone.features`` / ``head`` -- those modules stay the owners of the fp32 parameters so
that ``state_dict`` keys and shapes are the reference's).  On a device tensor the
modules' ``forward`` is bypassed and the plan drives libyolo_hip.so:

  * activations: zero-haloed NHWC bf16 buffers with a guard band (``Act``), allocated once per
    (batch, mode) and reused; producers only ever write the interior, so halos stay zero;
  * weights: bf16 panels re-packed from the fp32 masters only when a parameter's version changes;
  * forward = one yolo_igemm per conv / Linear (+ pool / flatten helpers);
  * backward = per conv one yolo_wgrad (flat pixel indexing) + one yolo_igemm data-gradient whose
    epilogue applies the previous LeakyReLU's derivative (or a pool backward).

Everything is enqueued on the current PyTorch stream; there is no host synchronisation.
"""

from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field

import torch
import torch.nn as nn

from . import _hip
from ._hip import (EPI_BIAS, EPI_BIAS_ADD_LRELU, EPI_BIAS_LRELU, EPI_MUL_DLRELU, EPI_NONE, ConvPackItem, ConvUnpackItem, IgemmDesc, PoolDesc, WgradDesc, check, lib, ptr,
                   stream)


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


# bench.py / profiling: when TIMERS is a list, every MFMA / pool launch is bracketed by events on the
# launch stream and (tag, kernel, flops, e0, e1) is appended.  None (default) = no events at all.
TIMERS: list | None = None
# tests / tuning: force a tile configuration of yolo_igemm (0 = library heuristic, see yolo_igemm_desc.tile_hint)
TILE_HINT = 0
TILE_PX = 0     # with TILE_HINT: yolo_igemm_desc.tile_px of the forced configuration
BN_STATS_IN_CONV = True      # ResNet trunk in batch-statistics mode: BatchNorm's sums come out of the conv's epilogue (yolo_igemm_desc.bn_stats)
IGEMM_LAUNCHES = 0  # yolo_igemm launches so far (bench.py: launches per step of the dominant kernel)
STEM_KERNEL = True  # 7x7/s2 stem through yolo_conv_stem7_fwd (False: the generic row-segment implicit GEMM; tests compare)
STEM_POOL_BWD_FUSED = True  # backward of the pool + LeakyReLU behind the stem inside yolo_wgrad_stem7_pooled (False: separate pass)
STRIDE2_CLASSES = True  # data gradient of a stride-2 3x3 conv as four parity-class convs over the non-zero gradient slots
_SIDE_STREAMS: dict = {}
SIDE_LOW = True      # ... of the lowest scheduling priority: the dispatcher prefers the data-gradient chain (12.73 -> 12.55 ms per step)
WGRAD_STREAM = True  # backward: weight gradients run on a second HIP stream beside the data-gradient chain (they are off its critical path)
WGRAD_SLABS = False  # pipelined weight-gradient kernel: partial tiles stored as slabs and summed in fixed order instead of fp32 atomics on the
                     # gradient (yolo_wgrad_desc.slabs): bit-reproducible gradients, and 13-16 % faster per launch when the launch has the
                     # chip to itself (conv10: 0.178 -> 0.150 ms) -- but the step as scheduled (weight gradients beside the data-gradient
                     # chain) gets SLOWER, 12.06 -> 12.28 ms: the slabs' 2 x 64 MB per layer compete with the chain's HBM-bound epilogues,
                     # while the atomics' traffic overlaps them.  Off by default; switch on for reproducible training runs.
WGRAD_PIPE = True  # weight gradient of the big deep 3x3 layers through wgrad_pipe.hip (yolo_wgrad_desc.variant = 5)
FC_NORM_IN_WGRAD = 1 << 26   # Linear layers with at least this many weights: yolo_wgrad also sums the squares of the gradient it stores
STEM_F32_INPUT = True  # inference: the stem kernel reads the NCHW fp32 input itself (no separate layout pass)
FLATTEN_FREE = True  # inference: conv -> nn.Flatten -> Linear without the flatten pass (dense NHWC conv output + K-permuted weight panels)
POOL_CODES = True  # training: a fused conv + pool stores the pooled map and 2-bit arg-max codes, not the un-pooled activation
FUSE_POOL = True   # inference: fold MaxPool2d(2,2) into the preceding conv's epilogue where the geometry allows
PERSIST = True     # plans with the persistent kernels (tile_hint 20 / 21) run them (False: the pipelined kernels 15 / 16 instead -- for A/B runs)


def _igemm(L_, d, inp, w, bias, aux, out, st, what):
    global IGEMM_LAUNCHES
    IGEMM_LAUNCHES += 1
    check(L_.yolo_igemm(ctypes.byref(d), inp, w, bias, aux, out, st), what)


class _Streams:
    """where the executors get their streams from (tests of the stream schedule put recording stand-ins here)"""

    @staticmethod
    def current(dev):
        return torch.cuda.current_stream(dev)

    @staticmethod
    def side(dev, low):
        return _hip.side_stream(torch.device(dev), low=low)

    @staticmethod
    def use(s):
        return torch.cuda.stream(s)


STREAMS = _Streams()


class _on_side_stream:
    """``with _on_side_stream(main, side) as st:`` -- work issued inside goes to ``side`` (None: stays on ``main``), behind everything
    queued on ``main`` so far; ``st`` is the hipStream_t to launch on.  The caller joins with ``main.wait_stream(side)``.
    ``note(waiter, waited)``: told about the wait (the gradient reducer keeps track of which stream has seen which)."""

    def __init__(self, main_t, side_t, note=None):
        self.main_t, self.side_t, self.note = main_t, side_t, note

    def __enter__(self):
        if self.side_t is None:
            return ctypes.c_void_p(self.main_t.cuda_stream)
        self.side_t.wait_stream(self.main_t)
        if self.note is not None:
            self.note(self.side_t.cuda_stream, self.main_t.cuda_stream)
        self.ctx = STREAMS.use(self.side_t)
        self.ctx.__enter__()
        return ctypes.c_void_p(self.side_t.cuda_stream)

    def __exit__(self, *exc):
        if self.side_t is not None:
            self.ctx.__exit__(*exc)
        return False


class _timed:
    def __init__(self, tag: str, kernel: str, flops: float = 0.0):
        self.tag, self.kernel, self.flops = tag, kernel, flops

    def __enter__(self):
        if TIMERS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if TIMERS is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            TIMERS.append((self.tag, self.kernel, self.flops, self.e0, e1))
        return False


# ---- per-problem launch plans ---------------------------------------------------------------------------
# The best yolo_igemm configuration depends on the layer shape (tile quantisation over 256 CUs, K depth).  Plans are DATA:
# ``yolo/plans/gfx950.json`` ships the plans of every problem of the BASELINE configurations (measured once on MI355X by
# tools/tune_plans.py) and is loaded at import, so every process and every rank runs the same launches -- outputs are
# bit-identical across processes and nothing is timed, flushed or synchronised at run time.  A problem without an entry
# takes ``_default_plan`` (a deterministic function of the shape).  AUTOTUNE = True (tools/tune_plans.py only) times the
# candidates on first use and records the winner in _TUNED.
#
# plan forms (tuples; JSON lists):
#   (hint, order)                          one launch of tile configuration `hint`
#   (hint, order, px_cut, tail_hint)       pixels [0, px_cut) with `hint` (whole rounds of the chip), the rest with `tail_hint`
#   ("skew", hint, order, phases, step)    one launch, first-round workgroups start phase * step cycles apart
#   ("tile", hint, order, tile_px[, phases, step])   one launch whose tiles cover tile_px pixels (yolo_igemm_desc.tile_px)
#   ("splitk", hint, S)                    S <= 2 K-splits with fp32 atomics into a zeroed scratch + yolo_igemm_finish
#   ("slabs", hint, S, tile_px)            S K-splits stored as slabs + fixed-order reduce in yolo_igemm_finish (deterministic)
AUTOTUNE = False
_TUNE_CANDIDATES = (5, 11, 12, 3, 4)      # 128x128 | 256x128 staggered | 256x256 staggered | 128x64 | 64x128
_TUNED: dict = {}
_FLUSH = None
PLAN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans", "gfx950.json")


def _key_str(key) -> str:
    return ",".join(str(int(v)) for v in key)


def load_plans(path: str = PLAN_FILE) -> int:
    """merge the plans of a JSON file ({"<key>": [plan...]}) into _TUNED; returns the number of entries read"""
    import json
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        data = json.load(f)
    for k, v in data.get("plans", {}).items():
        _TUNED[tuple(int(t) for t in k.split(","))] = tuple(v)
    return len(data.get("plans", {}))


def save_plans(path: str = PLAN_FILE, note: str = "") -> None:
    import json
    os.makedirs(os.path.dirname(path), exist_ok=True)
    body = {"arch": "gfx950", "key": "N,Ho,Wo,KH,KW,tap_len,Cout,stride,epilogue,pool2,out_px_stride,in_px_stride", "note": note,
            "plans": {_key_str(k): list(v) for k, v in sorted(_TUNED.items())}}
    with open(path, "w") as f:
        json.dump(body, f, indent=0, separators=(",", ":"))
        f.write("\n")


def _flush_caches(dev):
    """evict L2 / Infinity Cache between tuning runs (512 MiB write): in the network every layer meets its
    weights cold, which is what decides e.g. the tile order of the 1024-channel layers"""
    global _FLUSH
    if _FLUSH is None or _FLUSH.device != dev:
        _FLUSH = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    _FLUSH.fill_(1)


# tile edge (co, px slots) of the configurations the tuner may combine
_TILE = {1: (128, 128), 3: (128, 64), 4: (64, 128), 5: (128, 128), 10: (64, 128), 11: (256, 128), 12: (256, 256), 13: (256, 128), 14: (256, 208),
         15: (256, 208), 16: (256, 224), 17: (128, 208), 18: (128, 224), 19: (64, 16),       # (19: the streaming 1x1 kernel works in 16-pixel groups)
         20: (256, 208), 21: (256, 224)}                                                     # persistent kernels (igemm_persist.hip)
_TAIL_CANDIDATES = (5, 3, 4)
# (workgroup slots of the chip, relative time of one tile) per configuration, for _default_plan: 8-wave configurations run
# one workgroup per CU, the 4-wave ones two; times are relative to a 256x256 tile and follow the measured in-tile rates
_TILE_COST = {12: (256, 1.00), 14: (256, 0.80), 11: (256, 0.54), 5: (512, 0.36), 3: (512, 0.20), 4: (512, 0.20)}

_SPLITK_SCRATCH: dict = {}


def _splitk_scratch(n: int, zero: bool) -> torch.Tensor:
    """fp32 scratch of n elements on the current device for a split-K launch (allocated once per device and grown on
    demand; the atomics form needs it zero-filled, the slab form does not)"""
    dev = torch.cuda.current_device()
    buf = _SPLITK_SCRATCH.get(dev)
    if buf is None or buf.numel() < n:
        buf = _SPLITK_SCRATCH[dev] = torch.empty(n, dtype=torch.float32, device=torch.device("cuda", dev))
    v = buf[:n]
    if zero:
        v.zero_()
    return v


_WGRAD_SLAB_BUF: dict = {}


def _attach_wgrad_slabs(L_, wd: WgradDesc, dev) -> None:
    """slab mode of the pipelined weight-gradient kernel (yolo_wgrad_desc.slabs): partial tiles as plain stores + a fixed-order sum
    instead of fp32 atomics.  One scratch per device, grown on demand (the launches of one stream use it one after the other)."""
    need = ctypes.c_long(0)
    check(L_.yolo_wgrad_slab_floats(ctypes.byref(wd), ctypes.byref(need)), "wgrad_slab_floats")
    if need.value <= 0:
        return
    key = torch.device(dev).index
    buf = _WGRAD_SLAB_BUF.get(key)
    if buf is None or buf.numel() < need.value:
        buf = _WGRAD_SLAB_BUF[key] = torch.empty(need.value, dtype=torch.float32, device=dev)
    wd.slabs, wd.slab_floats = buf.data_ptr(), buf.numel()


def _pipe_ok(d: IgemmDesc) -> bool:
    """the register-pipelined kernels (tile_hint 15 .. 18) take the problem: an even number >= 4 of 32-deep K steps, no
    BatchNorm statistics"""
    nk = d.KH * d.KW * d.tap_len // 32
    return d.tap_len % 32 == 0 and nk % 2 == 0 and nk >= 4 and not d.bn_stats and not d.w_blocked


def _persist_ok(d: IgemmDesc) -> bool:
    """the persistent kernels (tile_hint 20 / 21: one software pipeline over all tiles of a workgroup, epilogue out of the accumulator
    registers) take the problem: an even number >= 6 of 32-deep K steps, whole 256-channel tiles (or one ragged tile), bf16 output"""
    nk = d.KH * d.KW * d.tap_len // 32
    return (d.tap_len % 32 == 0 and nk % 2 == 0 and nk >= 6 and not d.bn_stats and not d.w_blocked and not d.out_fp32 and d.split_k <= 1
            and (d.Cout <= 256 or d.Cout % 256 == 0) and d.Cout % 16 == 0 and d.epilogue in (EPI_NONE, EPI_BIAS, EPI_BIAS_LRELU, EPI_MUL_DLRELU, EPI_BIAS_ADD_LRELU))


def _pipe_pool_ok(d: IgemmDesc) -> bool:
    """... and their pooled epilogue (224-pixel tiles of whole row pairs) the output geometry"""
    return _pipe_ok(d) and d.Wo in (112, 56, 28) and d.Ho % 2 == 0 and (d.Ho * d.Wo) % 112 == 0 and d.Cout % 8 == 0


def _default_plan(d: IgemmDesc):
    """launch plan of a problem without a measured entry: the configuration with the smallest predicted time =
    rounds over the chip's workgroup slots x relative tile time (deterministic, no timing)."""
    M = d.N * d.Ho * d.Wo
    if d.pool2:
        # the library's own pooled epilogue tiles 8 x 16 pixel patches; other maps go through the 224-pixel pipelined / persistent tiles
        if d.Ho % 8 == 0 and d.Wo % 16 == 0 and not (_pipe_pool_ok(d) and d.Cout >= 192):
            return (0, 0)
        if _persist_ok(d) and d.pool2 in (1, 3) and M % 224 == 0 and d.Cout > 128:
            return ("tile", 21, 1, 0)
        return ("tile", 16 if d.Cout > 128 else 18, 1, 0)
    if M < 2048:
        return (0, 0)
    if _persist_ok(d) and d.Cout >= 192 and M * ((d.Cout + 255) // 256) >= 96 * 208:
        # the persistent kernel (one software pipeline over a workgroup's tiles, epilogue out of the registers) won 42 of the 90 problems
        # measured at batch 64, every one with >= 192 output channels and enough tiles for half the chip; 196-pixel tiles where they
        # divide the pixels (this network's maps are 49 * 4^k pixels)
        return ("tile", 20, 1, 196 if M % 196 == 0 else 208)
    best, best_t = (0, 0), None
    for hint, (slots, cost) in _TILE_COST.items():
        tco, tpx = _TILE[hint]
        forms = [((hint, 1), tpx)]
        if hint == 14:
            h = 15 if _pipe_ok(d) else 14     # the pipelined loop where it applies
            forms = [(("tile", h, 1, 196), 196)] if M % 196 == 0 else [(("tile", h, 1, 208), 208)]
        for plan, px in forms:
            tiles = ((d.Cout + tco - 1) // tco) * ((M + px - 1) // px)
            t = ((tiles + slots - 1) // slots) * cost
            if best_t is None or t < best_t - 1e-9:
                best, best_t = plan, t
    return best


def _run_plan_igemm(L_, d: IgemmDesc, plan, inp, w, bias, aux, out, st, what):
    """run one yolo_igemm problem with a launch plan (forms: see above)"""
    if d.bn_stats and (plan[0] in ("splitk", "slabs") or (isinstance(plan[0], int) and len(plan) == 4)):
        # a launch that also accumulates BatchNorm statistics must be one plain launch: keep the plan's main configuration
        plan = (plan[1], 1) if plan[0] in ("splitk", "slabs") else (plan[0], plan[1])
    if plan[0] == "tile" and plan[1] in (20, 21) and (d.pool2 == 2 or not PERSIST or not _persist_ok(d)):
        # the persistent kernels pool with pool2 = 1 (inference) and 3 (training: pooled map + arg-max codes; a plan measured for
        # pool2 = 1 also serves pool2 = 3, see _tune_key); pooled map + un-pooled activation (pool2 = 2), and anything else they do not
        # take, run the pipelined kernels
        plan = ("tile", 16 if (d.pool2 or plan[1] == 21) else 15, plan[2], 0 if d.pool2 else plan[3]) + tuple(plan[4:])
    if d.bn_stats:      # the pipelined kernels (15 .. 18) have no statistics epilogue: the staggered 256 x 208 loop takes their place
        if plan[0] == "tile" and plan[1] in (15, 16, 17, 18):
            plan = ("tile", 14, plan[2], min(plan[3], 208)) + tuple(plan[4:])
        elif plan[0] == "skew" and plan[1] in (15, 16, 17, 18):
            plan = ("skew", 14) + tuple(plan[2:])
        elif isinstance(plan[0], int) and plan[0] in (15, 16, 17, 18):
            plan = (14, plan[1])
        elif isinstance(plan[0], int) and plan[0] == 19 and d.epilogue != EPI_NONE:
            plan = (10, plan[1])           # the streaming 1x1 kernel accumulates statistics of raw conv outputs only
    d.tile_px, d.px_begin, d.px_end, d.skew_phases, d.skew_step = 0, 0, 0, 0, 0
    if plan[0] in ("splitk", "slabs"):
        # few-pixel deep-K layer: S workgroups per output tile into a dense fp32 scratch (atomics, or one slab per split),
        # then the epilogue as a separate pass (yolo_igemm_finish)
        slabs = plan[0] == "slabs"
        S = plan[2]
        M = d.N * d.Ho * d.Wo
        acc = _splitk_scratch(M * d.Cout * (S if slabs else 1), zero=not slabs)
        d2 = IgemmDesc.from_buffer_copy(d)
        d2.out_fp32, d2.epilogue, d2.split_k, d2.tile_hint, d2.tile_order = 1, EPI_NONE, S, plan[1], 1
        d2.out_img_stride, d2.out_row_stride, d2.out_px_stride, d2.out_off = d.Ho * d.Wo * d.Cout, d.Wo * d.Cout, d.Cout, 0
        d2.split_slabs, d2.tile_px = (1, plan[3]) if slabs else (0, 0)
        _igemm(L_, d2, inp, w, None, None, ptr(acc), st, what)
        d.split_k, d.split_slabs = (S, 1) if slabs else (1, 0)
        try:
            check(L_.yolo_igemm_finish(ctypes.byref(d), ptr(acc), bias, aux, out, st), what + " (finish)")
        finally:
            d.split_k, d.split_slabs = 1, 0
        return
    if plan[0] in ("skew", "tile"):
        # one launch of an 8-wave configuration: ("skew", hint, order, phases, step) staggers the first-round workgroups
        # (yolo_igemm_desc.skew_phases); ("tile", hint, order, tile_px[, phases, step]) sets the pixels per tile as well
        if plan[0] == "skew":
            d.tile_hint, d.tile_order, d.skew_phases, d.skew_step = plan[1], plan[2], plan[3], plan[4]
        else:
            d.tile_hint, d.tile_order, d.tile_px = plan[1], plan[2], plan[3]
            if len(plan) > 4:
                d.skew_phases, d.skew_step = plan[4], plan[5]
        try:
            _igemm(L_, d, inp, w, bias, aux, out, st, what)
        finally:
            d.skew_phases, d.skew_step, d.tile_px = 0, 0, 0
        return
    d.tile_hint, d.tile_order = plan[0], plan[1]
    if len(plan) == 2:
        _igemm(L_, d, inp, w, bias, aux, out, st, what)
        return
    d.px_begin, d.px_end = 0, plan[2]
    _igemm(L_, d, inp, w, bias, aux, out, st, what)
    d.tile_hint, d.px_begin, d.px_end = plan[3], plan[2], 0
    _igemm(L_, d, inp, w, bias, aux, out, st, what)
    d.px_begin, d.px_end = 0, 0


def _tune(L_, d: IgemmDesc, inp, w, bias, aux, out, st, what):
    """time the candidate plans of one problem (events on the launch stream behind a cache flush, min of 3) and return the
    fastest.  Only reached with AUTOTUNE = True (tools/tune_plans.py)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    stats_ptr, d.bn_stats = d.bn_stats, None        # tuning repeats the launch: keep it idempotent
    M = d.N * d.Ho * d.Wo

    def timed(plan):
        try:
            _run_plan_igemm(L_, d, plan, inp, w, bias, aux, out, st, what)
        except _hip.HipUnsupported:
            return None          # this configuration does not take the shape; every other error is a real failure
        ts = []
        for _ in range(3):
            _flush_caches(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _run_plan_igemm(L_, d, plan, inp, w, bias, aux, out, st, what)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts)

    times = {}

    def consider(plan):
        t = timed(plan)
        if t is not None:
            times[plan] = t

    cands = [c for c in _TUNE_CANDIDATES if not (d.pool2 and c in (3, 12))]
    if d.pool2 and not (d.Ho % 8 == 0 and d.Wo % 16 == 0):
        cands = []                # the 8 x 16-patch pooled epilogue does not tile this map: pipelined 224-pixel tiles only
    if d.KH == 1 and d.KW == 1 and d.tap_len <= 256 and not d.pool2:
        cands.append(10)      # thin-K 1x1 layers stream: 64x128x32, 28 KB of LDS, five workgroups per CU
    if d.KH == 1 and d.KW == 1 and not d.pool2:
        if d.tap_len in (64, 128, 192, 256, 512) and d.Cout % 64 == 0 and (d.N * d.Ho * d.Wo) % 16 == 0 and not d.out_fp32 and d.split_k <= 1:
            cands.append(19)  # ... or the streaming 1x1 kernel (igemm_stream.hip): weight panel in LDS, activations straight into MFMA fragments
    orders = (1, 2) if (d.Cout * d.KH * d.KW * d.tap_len * 2 > (4 << 20) and d.Cout >= 1024) else (1,)
    for c in cands:
        for o in orders:
            consider((c, o))
    # tiles of 49 * 4 pixels: this network's layers have N * 49 * 4^k output pixels -- whole rounds of 256 CUs at batch 64
    tile_px = 196 if M % 196 == 0 else 208
    if not d.pool2:
        for o in orders:
            consider(("tile", 14, o, tile_px))
            if _pipe_ok(d):
                consider(("tile", 15, o, tile_px))      # the same tile, register-pipelined one-barrier loop
                consider(("tile", 17, o, tile_px))      # 128 channels x 208, two workgroups per CU (short-K layers)
                if (d.Ho * d.Wo) % 112 == 0:
                    consider(("tile", 16, o, 0))        # 224-pixel tiles
                    consider(("tile", 18, o, 0))
            if _persist_ok(d):
                for tp in sorted({tile_px, 208, 0 if M % 224 else 224} - {0}):      # persistent loop: whole rounds matter less, full tiles more
                    consider(("tile", 21 if tp == 224 else 20, o, tp))
    elif _pipe_pool_ok(d):
        consider(("tile", 16, 1, 0))                    # MaxPool2d(2,2) fused into the pipelined kernels' epilogue
        consider(("tile", 18, 1, 0))
        if _persist_ok(d) and d.pool2 in (1, 3) and M % 224 == 0:
            consider(("tile", 21, 1, 0))                # ... and into the persistent kernel's register epilogue (quad-permute max)
    # tail plans: the two fastest large-tile configurations, cut where their tiles stop filling whole rounds of
    # 256 (one workgroup per CU) or 512 slots, remainder with a small tile
    if times and not d.pool2:
        singles = sorted((pl for pl in times if isinstance(pl[0], int)), key=times.get)
        for (c, o) in [pl for pl in singles if _TILE[pl[0]][0] * _TILE[pl[0]][1] >= 128 * 128][:2]:
            tco, tpx = _TILE[c]
            n_co = (d.Cout + tco - 1) // tco
            tiles = n_co * ((M + tpx - 1) // tpx)
            cuts = set()
            for slots in (256, 512):
                full = tiles // slots * slots
                cut = full // n_co * tpx
                if 0 < cut < M and tiles - full < 0.9 * slots:
                    cuts.add(cut)
            for cut in sorted(cuts):
                for tc in _TAIL_CANDIDATES:
                    consider((c, o, cut, tc))
    # start-skew plans: the 8-wave configurations run one workgroup per CU, all in the same phase; over several rounds
    # it pays to start the CUs a fraction of a tile time apart (see igemm.hip)
    if times and not d.pool2:
        eight = [pl for pl in sorted(times, key=times.get) if (len(pl) == 2 and pl[0] in (11, 12)) or pl[0] == "tile"][:2]
        for pl in eight:
            c, o, tpv = (pl[0], pl[1], 0) if isinstance(pl[0], int) else (pl[1], pl[2], pl[3])
            tco, tpx = _TILE[c]
            tiles = ((d.Cout + tco - 1) // tco) * ((M + (tpv or tpx) - 1) // (tpv or tpx))
            if tiles < 400:
                continue
            tile_cycles = times[pl] * 1e-3 / ((tiles + 255) // 256) * 2.1e9
            for ph, frac in ((3, 0.3), (5, 0.2), (3, 0.2), (5, 0.3)):
                consider(("tile", c, o, tpv, ph, int(frac * tile_cycles)) if tpv else ("skew", c, o, ph, int(frac * tile_cycles)))
    # split-K plans for few-pixel, deep-K layers (7x7x1024: 64 output tiles of 256 x 196 for 256 CUs): slabs summed in
    # fixed order, so any split count stays bit-reproducible
    if (times and not d.pool2 and not d.out_fp32 and M <= 8192 and d.KH * d.KW * d.tap_len >= 2304 and d.Cout % 8 == 0
            and d.epilogue in (EPI_NONE, EPI_BIAS, EPI_BIAS_LRELU, EPI_MUL_DLRELU)):
        for c, S, tpv in ((11, 2, 0), (5, 2, 0), (3, 2, 0), (14, 4, tile_px), (15, 4, tile_px), (15, 2, tile_px), (12, 4, 0), (11, 4, 0)):
            consider(("slabs", c, S, tpv))
    d.bn_stats = stats_ptr
    if not times:
        return (0, 0)
    best = min(times, key=times.get)
    if TUNE_LOG is not None:
        TUNE_LOG.append((_tune_key(d), best, sorted(times.items(), key=lambda kv: kv[1])[:6]))
    return best


TUNE_LOG: list | None = None      # tools/tune_plans.py: (key, winner, six fastest candidates with their times)


def igemm_call(d: IgemmDesc, inp, w, bias, aux, out, st, what: str):
    """yolo_igemm with the problem's launch plan (shipped table, else the deterministic default) -- only for plain,
    idempotent launches."""
    L_ = lib()
    plain = TILE_HINT == 0 and d.split_k <= 1 and not d.w_blocked and d.tap_len % 64 == 0
    if not plain:
        d.tile_hint, d.tile_px = TILE_HINT, TILE_PX
        _igemm(L_, d, inp, w, bias, aux, out, st, what)
        return
    key = _tune_key(d)
    best = _TUNED.get(key)
    if best is None:
        if AUTOTUNE and TIMERS is None and d.N * d.Ho * d.Wo >= 2048:
            best = _tune(L_, d, inp, w, bias, aux, out, st, what)
        else:
            best = _default_plan(d)
        _TUNED[key] = best
    _run_plan_igemm(L_, d, best, inp, w, bias, aux, out, st, what)


def _tune_key(d: IgemmDesc):
    # pool2 = 3 (pooled map + arg-max codes, training) runs the launch plan measured for pool2 = 1 (pooled map only, inference)
    return (d.N, d.Ho, d.Wo, d.KH, d.KW, d.tap_len, d.Cout, d.stride, d.epilogue, 1 if d.pool2 == 3 else d.pool2, d.out_px_stride, d.in_px_stride)


load_plans()


class _EventSlot:
    """holder of the event behind a background optimizer launch (yolo.optim.Adam.attach_plan(overlap=True)).  It lives on the plan
    object (not in a table keyed by id(plan), which outlives garbage collection); a deep copy of a plan starts with an empty slot --
    events do not copy, and the copy's parameters are new tensors nobody updates in the background."""

    def __init__(self):
        self.event = None

    def __deepcopy__(self, memo):
        return _EventSlot()

    def __reduce__(self):
        return (_EventSlot, ())

    def wait(self, dev=None, keep: bool = False):
        """the current stream waits for the pending update; ``keep``: leave the event in place for later readers on other streams"""
        ev = self.event
        if not keep:
            self.event = None
        if ev is not None:
            torch.cuda.current_stream(dev).wait_event(ev)


class Act:
    """Zero-haloed NHWC bf16 activation: [N][H+2h][W+2h][C] plus guard bands of zeros."""

    def __init__(self, N, H, W, C, halo, device, halo_hi=None):
        self.N, self.H, self.W, self.C = N, H, W, C
        self.halo = halo
        self.halo_hi = halo if halo_hi is None else halo_hi
        self.Hp = H + self.halo + self.halo_hi
        self.Wp = W + self.halo + self.halo_hi
        self.px_stride = C
        self.row_stride = self.Wp * C
        self.img_stride = self.Hp * self.Wp * C
        self.slots = N * self.Hp * self.Wp
        guard = _round_up((self.Wp + 2) * C + 64 * 8, 128)
        self.store = torch.zeros(guard + self.slots * C + guard, dtype=torch.bfloat16, device=device)
        self.t = self.store[guard: guard + self.slots * C]

    @property
    def p(self):
        return ctypes.c_void_p(self.t.data_ptr())

    def interior_off(self, shift=0):
        """element offset of logical pixel (-shift, -shift) inside an image"""
        h = self.halo - shift
        return (h * self.Wp + h) * self.C

    def view(self):
        return self.t.view(self.N, self.Hp, self.Wp, self.C)

    def interior(self):
        h = self.halo
        return self.view()[:, h: h + self.H, h: h + self.W, :]


@dataclass
class Layer:
    kind: str                      # conv | pool | flatten | fc
    name: str = ""
    Cout: int = 0
    Cin: int = 0
    K: int = 1
    stride: int = 1
    pad: int = 0
    lrelu: bool = False
    dropout: float = 0.0
    weight: nn.Parameter | None = None
    bias: nn.Parameter | None = None
    first: bool = False            # the 3-channel 7x7/s2 stem (NHWC4 input, row-segment taps)
    # geometry, filled by Plan._shape
    Hin: int = 0
    Win: int = 0
    Hout: int = 0
    Wout: int = 0


class Plan:
    """Executable plan for [conv|pool]* [flatten fc*]? ."""

    SLOPE = 0.1

    def __init__(self, layers: list[Layer], in_channels: int, input_is_image: bool, S: int | None = None):
        self.layers = layers
        self.in_channels = in_channels
        self.input_is_image = input_is_image
        self.params: list[nn.Parameter] = []
        for L in layers:
            if L.kind in ("conv", "fc"):
                self.params += [L.weight, L.bias]
        self._pf: dict[int, tuple] = {}
        self._pd: dict[int, tuple] = {}
        self._pfb: dict[int, tuple] = {}
        self._pd2: dict[int, tuple] = {}
        self._ws: dict[tuple, list] = {}
        self.grad_norm_sq: dict = {}   # id(weight) -> ((data_ptr, shape) of the gradient, its version, device double |g|^2) left by the last backward pass
        self.debug_keep = False      # tests: True = keep the last workspace (activations + gradients) for inspection AND store the
                                     # un-pooled activations; "codes" = keep the workspace of the product path (pooled maps + arg-max codes)
        self.last = None
        # optional persistent gradient arena (data-parallel training): one flat fp32 buffer holding every
        # parameter gradient in the order backward PRODUCES them (last layer first), so that finished
        # gradients form a growing contiguous prefix that can be all-reduced while backward continues
        self.arena = None
        self.arena_views: dict[int, tuple] = {}
        self.on_grad_ready = None    # callback(lo, hi): arena[lo:hi] (elements) is final -- called with the PRODUCING stream current
        self.on_backward_done = None # callback(): every gradient is final and the current stream has waited for all of them
        self.on_stream_wait = None   # callback(waiter, waited): hipStream_t handles; `waiter` now waits for everything queued on `waited`
        self.params_ready = _EventSlot()      # event behind a background update of the Linear layers (forward waits in front of them)
        self.owner = None            # weakref to the nn.Module whose layers this plan runs (models.*.hip_plan sets it)

    def attach_grad_arena(self, device) -> torch.Tensor:
        """Allocate the gradient arena; backward then writes gradients into it, assigns ``p.grad`` views and
        returns no gradients to autograd (gradients are OVERWRITTEN each backward: no accumulation)."""
        order = [li for li in reversed(range(len(self.layers))) if self.layers[li].kind in ("conv", "fc")]
        off = 0
        wv, bv = {}, {}
        for li in order:                          # every view starts on a 256-B boundary (float4 kernels)
            n = self.layers[li].weight.numel()
            wv[li] = (off, off + n, _round_up(off + n, 64))
            off = _round_up(off + n, 64)
        self._arena_w_end = off
        for li in order:
            n = self.layers[li].bias.numel()
            bv[li] = (off, off + n)
            off = _round_up(off + n, 64)
        self.arena = torch.zeros(off, dtype=torch.float32, device=device)
        self.arena_views = {li: (self.arena[wv[li][0]:wv[li][1]].view_as(self.layers[li].weight),
                                 self.arena[bv[li][0]:bv[li][1]].view_as(self.layers[li].bias), wv[li][0], wv[li][2])
                            for li in order}
        return self.arena

    @staticmethod
    def _side_stream(dev) -> "torch.cuda.Stream":
        """the second stream of the backward pass (weight gradients): one per device, shared by all plans (kept outside the plan
        objects, which are deep-copied with their modules)"""
        key = torch.device(dev).index
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = STREAMS.side(dev, SIDE_LOW)
        return _SIDE_STREAMS[key]

    def _layer_done(self, li: int):
        if self.arena is not None and self.on_grad_ready is not None:
            _, _, a, b = self.arena_views[li]
            self.on_grad_ready(a, b)

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def from_modules(mods, in_channels: int, input_is_image: bool) -> "Plan":
        """mods: flat list of nn.Module (Conv2d, LeakyReLU, MaxPool2d, Flatten, Linear, Dropout)."""
        layers: list[Layer] = []
        i = 0
        mods = list(mods)
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, nn.Conv2d):
                k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
                if m.kernel_size[0] != m.kernel_size[1] or m.groups != 1 or m.dilation != (1, 1) or m.bias is None:
                    raise ValueError(f"unsupported conv {m}")
                act = isinstance(nxt, nn.LeakyReLU)
                if act and abs(nxt.negative_slope - Plan.SLOPE) > 1e-12:
                    raise ValueError("only LeakyReLU(0.1) is fused")
                first = (m.in_channels == 3 and k == 7 and s == 2 and p == 3)
                if not first and not ((k == 3 and p == 1) or (k == 1 and p == 0)) or (not first and s not in (1, 2)):
                    raise ValueError(f"unsupported conv geometry {m}")
                if not first and (m.in_channels % 32 or m.out_channels % 8):
                    raise ValueError(f"unsupported channel counts {m}")
                layers.append(Layer("conv", Cout=m.out_channels, Cin=m.in_channels, K=k, stride=s, pad=p, lrelu=act,
                                    weight=m.weight, bias=m.bias, first=first))
                i += 2 if act else 1
            elif isinstance(m, nn.MaxPool2d):
                ks = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
                st = m.stride if isinstance(m.stride, int) else m.stride[0]
                if ks != 2 or st != 2:
                    raise ValueError("only MaxPool2d(2,2)")
                layers.append(Layer("pool"))
                i += 1
            elif isinstance(m, nn.Flatten):
                layers.append(Layer("flatten"))
                i += 1
            elif isinstance(m, nn.Linear):
                if m.in_features % 64 or m.bias is None:
                    raise ValueError(f"unsupported Linear {m}: in_features must be a multiple of 64")
                act = isinstance(nxt, nn.LeakyReLU)
                j = i + (2 if act else 1)
                drop = 0.0
                if j < len(mods) and isinstance(mods[j], nn.Dropout):
                    drop = mods[j].p
                    j += 1
                layers.append(Layer("fc", Cout=m.out_features, Cin=m.in_features, lrelu=act, dropout=drop, weight=m.weight, bias=m.bias))
                i = j
            else:
                raise ValueError(f"unsupported module in plan: {m}")
        return Plan(layers, in_channels, input_is_image)

    # ------------------------------------------------------------------ weights
    # bf16 operand copies of the fp32 masters, cached per layer and keyed on (tensor version, storage):
    #   _pf[li] = (key, forward operand)        conv: [Cout][KH][KW][Cin]      Linear: [O][K] (the master's layout)
    #   _pd[li] = (key, data-gradient operand)  conv: [Cin][KH][KW][Cout] flipped   Linear: [K][ld(O)] (only the
    #             small Linear layers; the big one in front of nn.Flatten uses the forward copy, see backward)
    @staticmethod
    def _wkey(w):
        return (w._version, w.data_ptr())

    def _multi_ok(self, L: Layer) -> bool:
        return L.kind == "conv" and not L.first and L.Cout % 64 == 0 and L.Cin % 64 == 0 and L.K * L.K <= 9

    def _src(self, L: Layer):
        wsrc = L.weight.detach()
        if wsrc.dtype != torch.float32 or not wsrc.is_contiguous():
            wsrc = wsrc.float().contiguous()
        return wsrc

    def _pack_all(self, need_dgrad: bool):
        """Refresh every stale conv operand of the plan in ONE launch (yolo_pack_conv_weights_multi)."""
        items, keep, done = [], [], []
        for li, L in enumerate(self.layers):
            if not self._multi_ok(L):
                continue
            key = self._wkey(L.weight)
            f, d = self._pf.get(li), self._pd.get(li)
            want_f = f is None or f[0] != key
            want_d = need_dgrad and li > 0 and (d is None or d[0] != key)
            if not (want_f or want_d):
                continue
            dev = L.weight.device
            wsrc = self._src(L)
            keep.append(wsrc)
            wf = wd = None
            if want_f:
                wf = f[1] if f is not None else torch.empty((L.Cout, L.K, L.K, L.Cin), dtype=torch.bfloat16, device=dev)
            if want_d:
                wd = d[1] if d is not None else torch.empty((L.Cin, L.K, L.K, L.Cout), dtype=torch.bfloat16, device=dev)
            items.append(ConvPackItem(wsrc.data_ptr(), wf.data_ptr() if wf is not None else None, wd.data_ptr() if wd is not None else None,
                                      L.Cout, L.Cin, L.K, L.K))
            done.append((li, key, wf, wd))
        if items:
            tab = (ConvPackItem * len(items))(*items)
            check(lib().yolo_pack_conv_weights_multi(tab, len(items), stream()), "pack_conv_weights_multi")
            for li, key, wf, wd in done:
                if wf is not None:
                    self._pf[li] = (key, wf)
                if wd is not None:
                    self._pd[li] = (key, wd)

    def _pack(self, li: int, need_dgrad: bool):
        """(forward operand, data-gradient operand | None) of layer li, refreshed if the master changed."""
        L = self.layers[li]
        key = self._wkey(L.weight)
        f, d = self._pf.get(li), self._pd.get(li)
        ok_f = f is not None and f[0] == key
        ok_d = d is not None and d[0] == key
        if ok_f and (ok_d or not need_dgrad):
            return f[1], (d[1] if ok_d else None)
        dev = L.weight.device
        wsrc = self._src(L)
        wf = f[1] if f is not None else None
        wd = d[1] if d is not None else None
        if L.kind == "conv":
            if L.first:
                if wf is None:
                    wf = torch.empty((L.Cout, 7, 8, 4), dtype=torch.bfloat16, device=dev)
                check(lib().yolo_pack_conv_weight(ptr(wsrc), L.Cout, 3, 7, 7, 4, 8, ptr(wf), None, stream()), "pack_conv_weight")
                self._pf[li] = (key, wf)
                return wf, None
            if wf is None:
                wf = torch.empty((L.Cout, L.K, L.K, L.Cin), dtype=torch.bfloat16, device=dev)
            if need_dgrad and wd is None:
                wd = torch.empty((L.Cin, L.K, L.K, L.Cout), dtype=torch.bfloat16, device=dev)
            check(lib().yolo_pack_conv_weight(ptr(wsrc), L.Cout, L.Cin, L.K, L.K, L.Cin, L.K, None if ok_f else ptr(wf),
                                              ptr(wd) if (need_dgrad and not ok_d) else None, stream()), "pack_conv_weight")
        else:
            if not ok_f:
                if wf is None:
                    wf = torch.empty((L.Cout, L.Cin), dtype=torch.bfloat16, device=dev)
                check(lib().yolo_cast_f32_to_bf16(ptr(wsrc), wsrc.numel(), ptr(wf), stream()), "cast fc weight")
            if need_dgrad and not ok_d:
                ld = _round_up(L.Cout, 32)
                if wd is None:
                    wd = torch.zeros((L.Cin, ld), dtype=torch.bfloat16, device=dev)
                check(lib().yolo_transpose_f32_to_bf16(ptr(wsrc), L.Cout, L.Cin, ptr(wd), ld, stream()), "transpose")
        self._pf[li] = (key, wf)
        if need_dgrad:
            self._pd[li] = (key, wd)
            return wf, wd
        return wf, (wd if ok_d else None)

    def _stride2_panels(self, li: int, wdg: torch.Tensor) -> dict:
        """data-gradient operands of a stride-2 3x3 conv by input-pixel parity: slices of the flipped panel
        wd[ci][ky'][kx'][co] (ky' = 2 - ky): parity 0 uses ky' = 1, parity 1 uses ky' = 0 (tap offset 0) and 2 (offset 1)."""
        L = self.layers[li]
        key = self._wkey(L.weight)
        hit = self._pd2.get(li)
        if hit is not None and hit[0] == key:
            return hit[1]
        sel = {0: slice(1, 2), 1: slice(0, 3, 2)}      # basic slices: one strided copy per class, no gather kernels
        panels = {}
        for py in (0, 1):
            for px in (0, 1):
                panels[(py, px)] = wdg[:, sel[py], sel[px], :].contiguous()
        self._pd2[li] = (key, panels)
        return panels

    def _pack_fc_blocked(self, li: int, hwc=None):
        """inference operand of a Linear layer: bf16 [O/128][K/64][128][64] panels (contiguous 16-KB stage reads; the
        plain [O][K] copy that training shares with the optimizer streams ~15 % slower).  ``hwc = (C, HW)``: K axis permuted from
        nn.Flatten's (c, hw) order to (hw, c) -- the layer then reads the dense NHWC conv output directly (FLATTEN_FREE)."""
        L = self.layers[li]
        key = (self._wkey(L.weight), hwc)
        hit = self._pfb.get(li)
        if hit is not None and hit[0] == key:
            return hit[1]
        wsrc = self._src(L)
        wb = hit[1] if hit is not None else torch.empty((_round_up(L.Cout, 128) * L.Cin,), dtype=torch.bfloat16, device=L.weight.device)
        if hwc is not None:
            check(lib().yolo_pack_fc_weight_blocked_hwc(ptr(wsrc), L.Cout, hwc[0], hwc[1], ptr(wb), stream()), "pack_fc_blocked_hwc")
        else:
            check(lib().yolo_pack_fc_weight_blocked(ptr(wsrc), L.Cout, L.Cin, ptr(wb), stream()), "pack_fc_blocked")
        self._pfb[li] = (key, wb)
        return wb

    def fc_biases(self):
        """bias parameters of the Linear layers on the device (updated together with their weights, yolo.optim.Adam.attach_plan)"""
        return [L.bias for L in self.layers if L.kind == "fc" and L.bias is not None and L.bias.is_cuda]

    def bf16_shadows(self):
        """[(param, bf16 forward operand with the master's layout, callback)] for the Linear layers: an optimizer
        that writes bf16(p) into the operand while it updates p calls ``callback(p)`` afterwards
        (yolo.optim.Adam.attach_plan), which saves the 822 MB + 411 MB re-cast of the big Linear per step."""
        out = []
        for li, L in enumerate(self.layers):
            if L.kind != "fc" or not L.weight.is_cuda:
                continue
            wf, _ = self._pack(li, False)

            def fresh(p, li=li, wf=wf):
                self._pf[li] = (self._wkey(p), wf)
            out.append((L.weight, wf, fresh))
        return out

    # ------------------------------------------------------------------ workspace
    def _workspace(self, N: int, x_shape, device, train: bool):
        key = (N, tuple(x_shape[1:]), str(device), train)
        pool = self._ws.setdefault(key, [])
        if pool:
            ws = pool.pop()
            self._apply_geom(ws)
            return key, ws
        ws = {"acts": [], "grads": {}, "misc": {}, "geom": {}}
        C, H, W = x_shape[1], x_shape[2], x_shape[3]
        if self.layers and self.layers[0].kind == "conv" and self.layers[0].first:
            a = Act(N, H, W, 4, 3, device)
        else:
            a = Act(N, H, W, C, 1, device)
        ws["in"] = a
        cur = a
        flat = None
        for li, L in enumerate(self.layers):
            if L.kind == "conv":
                L.Hin, L.Win = cur.H, cur.W
                L.Hout = (cur.H + 2 * L.pad - L.K) // L.stride + 1
                L.Wout = (cur.W + 2 * L.pad - L.K) // L.stride + 1
                # inference, conv -> nn.Flatten -> Linear: the conv writes a dense NHWC map (no halo) that the Linear layer reads as it
                # lies, through weight panels with a permuted K axis -- no flatten pass (FLATTEN_FREE)
                dense = (not train and FLATTEN_FREE and li + 2 < len(self.layers) and self.layers[li + 1].kind == "flatten"
                         and self.layers[li + 2].kind == "fc" and L.Cout % 8 == 0 and (L.Cout * L.Hout * L.Wout) % 64 == 0)
                cur = Act(N, L.Hout, L.Wout, L.Cout, 0 if dense else 1, device)
            elif L.kind == "pool":
                L.Hin, L.Win = cur.H, cur.W
                cur = Act(N, cur.H // 2, cur.W // 2, cur.C, 1, device)
            elif L.kind == "flatten":
                flat = torch.empty((N, cur.C * cur.H * cur.W), dtype=torch.bfloat16, device=device)
                cur = flat
            elif L.kind == "fc":
                feat = flat.shape[1] if (flat is not None and cur is flat) else None
                if feat is not None and feat != L.Cin:
                    # the reference raises here too (stock nn.Linear): e.g. a 224x224 batch into the 448x448 head
                    raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({N}x{feat} and {L.Cin}x{L.Cout}): input of "
                                       f"{tuple(x_shape[2:])} pixels does not match the Linear layer behind nn.Flatten")
                cur = None  # allocated per call (tiny)
            ws["acts"].append(cur)
            ws["geom"][li] = (L.Hin, L.Win, L.Hout, L.Wout)
        return key, ws

    def _apply_geom(self, ws):
        """the layer geometry (Hin, Win, Hout, Wout) belongs to a workspace, not to the plan: a plan may serve several input
        sizes, and a forward at another size may run between a training forward and its backward.  Every entry point that
        reads ``L.Hin`` .. ``L.Wout`` calls this first with the workspace it is about to use."""
        for li, g in ws["geom"].items():
            L = self.layers[li]
            L.Hin, L.Win, L.Hout, L.Wout = g

    def _release(self, key, ws):
        self._ws.setdefault(key, []).append(ws)

    # ------------------------------------------------------------------ descriptors
    def _conv_desc(self, L: Layer, a_in: Act, a_out: Act) -> IgemmDesc:
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = a_in.N, L.Hout, L.Wout
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.stride = L.stride
        if L.first:
            d.in_off = 0
            d.KH, d.KW, d.tap_len = 7, 1, 32
        else:
            d.in_off = a_in.interior_off(L.pad)
            d.KH, d.KW, d.tap_len = L.K, L.K, L.Cin
        d.Cout = L.Cout
        d.out_img_stride, d.out_row_stride, d.out_px_stride = a_out.img_stride, a_out.row_stride, a_out.px_stride
        d.out_off = a_out.interior_off()
        d.epilogue = EPI_BIAS_LRELU if L.lrelu else EPI_BIAS
        d.slope = self.SLOPE
        d.out_fp32 = 0
        d.split_k = 1
        d.tile_hint, d.tile_px = TILE_HINT, TILE_PX
        return d

    @staticmethod
    def _pool_fusable(L: Layer) -> bool:
        """conv -> LeakyReLU -> MaxPool2d(2,2) as one launch: the library's 8 x 16-patch pooled epilogue (224^2 and 112^2
        maps), or the pipelined kernels' 224-pixel tiles of whole row pairs (rows of 112, 56 or 28 pixels)"""
        if L.Cout % 8:
            return False
        if L.Hout % 8 == 0 and L.Wout % 16 == 0:
            return True
        nk = L.K * L.K * L.Cin // 32
        return (not L.first and L.Cin % 32 == 0 and nk % 2 == 0 and nk >= 4 and L.Wout in (112, 56, 28) and L.Hout % 2 == 0
                and (L.Hout * L.Wout) % 112 == 0)

    # ------------------------------------------------------------------ forward
    @_hip.device_guard
    def forward(self, x: torch.Tensor, train: bool, drop_training: bool, u8_size=None):
        """x: NCHW fp32 device tensor -- or, with ``u8_size = (H, W)``, decoded uint8 images [N][h][w][3] that
        yolo_preprocess_u8 resizes + normalises straight into the stem's NHWC4 input buffer (no fp32 NCHW round trip).
        Returns (out, saved) -- out is (N, O) fp32 if the plan ends with an fc layer, else NCHW fp32 features."""
        L_ = lib()
        st = stream()
        N = x.shape[0]
        dev = x.device
        x = x.detach()
        stem_f32 = False
        if u8_size is not None:
            from . import preprocess as _pp
            key, ws = self._workspace(N, (N, 3, u8_size[0], u8_size[1]), dev, train)
            self._pack_all(train)
            a = ws["in"]
            if not (a.C == 4 and a.halo == 3):
                raise ValueError("uint8 input needs a plan that starts with the 7x7/s2 stem")
            _pp.preprocess_u8_into(x, u8_size, a)
        else:
            if x.dim() != 4 or x.shape[1] != self.in_channels:
                raise RuntimeError(f"expected input of shape (N, {self.in_channels}, H, W), got {tuple(x.shape)}")
            if x.dtype != torch.float32 or not x.is_contiguous():
                x = x.float().contiguous()
            key, ws = self._workspace(N, x.shape, dev, train)
            self._pack_all(train)
            a = ws["in"]
            L0 = self.layers[0]
            # inference: the stem kernel reads the caller's NCHW fp32 batch itself (the patch is converted on its way into LDS); training
            # keeps the NHWC4 copy, which the stem's weight gradient reads
            stem_f32 = (not train and STEM_F32_INPUT and STEM_KERNEL and a.C == 4 and a.halo == 3 and L0.kind == "conv" and L0.first and L0.Cout == 64
                        and L0.Hout % 8 == 0 and L0.Wout % 16 == 0 and 2 * L0.Hout == x.shape[2] and 2 * L0.Wout == x.shape[3]
                        and ws["acts"][0].C == 64 and L0.bias.dtype == torch.float32)      # = the conditions of the stem-kernel branch below
            if stem_f32:
                pass
            elif a.C == 4 and a.halo == 3:
                check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, x.shape[1], x.shape[2], x.shape[3], a.p, 4, 3, 3, st), "nchw->nhwc4")
            else:
                check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, x.shape[1], x.shape[2], x.shape[3], a.p, a.C, 1, 1, st), "nchw->nhwc")
        cur = a
        fc_saved = {}
        out = None
        skip_pool = False
        # training, conv -> LeakyReLU -> MaxPool2d(2,2): the fused epilogue stores the pooled map and, per pooled element, the 2-bit
        # window position of the maximum; the backward pass needs nothing else of the un-pooled activation (debug_keep: the tests'
        # teacher-forced checks read that activation, so it is written instead)
        self.grad_norm_sq.clear()
        hwc = None
        codes_mode = train and POOL_CODES and self.debug_keep is not True      # (debug_keep = "codes": keep the workspace of the product path)
        ws["codes"] = set()
        for li, L in enumerate(self.layers):
            nxt = ws["acts"][li]
            if L.kind == "conv":
                wf, _ = self._pack(li, train)
                # inference: conv -> LeakyReLU -> MaxPool2d(2,2) as ONE launch when the conv output tiles into
                # 8 x 16 pixel patches (the first two layers: 224^2 and 112^2); training keeps the un-pooled
                # activation, which the backward pass needs
                fuse = (not train and FUSE_POOL and li + 1 < len(self.layers) and self.layers[li + 1].kind == "pool"
                        and self._pool_fusable(L))
                if fuse:
                    nxt = ws["acts"][li + 1]
                d = self._conv_desc(L, cur, nxt)
                d.pool2 = 1 if fuse else 0
                b = L.bias.detach()
                if L.first and STEM_KERNEL and L.Cout == 64 and L.Hout % 8 == 0 and L.Wout % 16 == 0 and nxt.C == 64 and b.dtype == torch.float32:
                    # dedicated stem kernel: input patch staged once per 8x16 tile, weights in registers; in training the
                    # following MaxPool2d is fused as well, with the un-pooled activation written next to the pooled one
                    dual = (train and FUSE_POOL and li + 1 < len(self.layers) and self.layers[li + 1].kind == "pool")
                    full = nxt if dual else None
                    dst = ws["acts"][li + 1] if dual else nxt
                    codes = self._codes(ws, li, dst) if (dual and codes_mode) else None
                    with _timed(f"conv{li}" + ("+pool" if (fuse or dual) else ""), "stem", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        if stem_f32 and li == 0:
                            check(L_.yolo_conv_stem7_fwd_f32(ptr(x), ptr(wf), ptr(b), N, x.shape[2], x.shape[3], self.SLOPE if L.lrelu else 1.0,
                                                             1 if fuse else 0, dst.p, dst.img_stride, dst.row_stride, dst.interior_off(), None, 0, 0, 0, st),
                                  "conv_stem7_fwd_f32")
                        elif codes is not None:       # pooled map + arg-max codes: the 411 MB un-pooled activation (batch 64) is never written
                            check(L_.yolo_conv_stem7_fwd(cur.p, ptr(wf), ptr(b), N, L.Hout, L.Wout, cur.img_stride, cur.row_stride,
                                                         self.SLOPE if L.lrelu else 1.0, 3, dst.p, dst.img_stride, dst.row_stride,
                                                         dst.interior_off(), ptr(codes), 0, 0, 0, st), "conv_stem7_fwd")
                        else:
                            check(L_.yolo_conv_stem7_fwd(cur.p, ptr(wf), ptr(b), N, L.Hout, L.Wout, cur.img_stride, cur.row_stride,
                                                         self.SLOPE if L.lrelu else 1.0, 1 if (fuse or dual) else 0, dst.p, dst.img_stride, dst.row_stride,
                                                         dst.interior_off(), full.p if dual else None, full.img_stride if dual else 0,
                                                         full.row_stride if dual else 0, full.interior_off() if dual else 0, st), "conv_stem7_fwd")
                    cur = dst
                    skip_pool = fuse or dual
                    continue
                # training: the same fused pool, with the un-pooled activation written too (pool2 = 2)
                dual = (train and FUSE_POOL and not fuse and li + 1 < len(self.layers) and self.layers[li + 1].kind == "pool"
                        and self._pool_fusable(L) and not L.first)
                if dual:
                    full, pooled = nxt, ws["acts"][li + 1]
                    d = self._conv_desc(L, cur, pooled)
                    codes = self._codes(ws, li, pooled) if codes_mode else None
                    if codes is not None:
                        d.pool2 = 3
                        auxp = ptr(codes)
                    else:
                        d.pool2 = 2
                        d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = full.img_stride, full.row_stride, full.px_stride, full.interior_off()
                        auxp = full.p
                    with _timed(f"conv{li}+pool", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        igemm_call(d, cur.p, ptr(wf), ptr(b), auxp, pooled.p, st, f"igemm conv{li}")
                    cur = pooled
                    skip_pool = True
                    continue
                with _timed(f"conv{li}" + ("+pool" if fuse else ""), "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                    igemm_call(d, cur.p, ptr(wf), ptr(b), None, nxt.p, st, f"igemm conv{li}")
                cur = nxt
                skip_pool = fuse
            elif L.kind == "pool" and skip_pool:
                skip_pool = False
            elif L.kind == "pool":
                pd = PoolDesc(N, cur.H, cur.W, cur.C, cur.halo, nxt.halo)
                with _timed(f"pool{li}", "maxpool2_fwd"):
                    check(L_.yolo_maxpool2_fwd(ctypes.byref(pd), cur.p, nxt.p, st), "maxpool")
                cur = nxt
            elif L.kind == "flatten":
                if not train and isinstance(cur, Act) and cur.halo == 0 and cur.halo_hi == 0 and FLATTEN_FREE:
                    hwc = (cur.C, cur.H * cur.W)            # the next Linear layer takes (hw, c)-ordered panels
                    cur = cur.t.view(N, -1)
                    continue
                check(L_.yolo_nhwc_bf16_to_nchw_bf16(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(nxt), st), "flatten")
                cur = nxt
            elif L.kind == "fc":
                self.params_ready.wait(dev)      # yolo.optim.Adam(overlap): the Linear layers' update of the last step runs on a second stream
                if train:
                    wf, _ = self._pack(li, False)
                else:
                    wf = self._pack_fc_blocked(li, hwc)
                    hwc = None
                xin = cur  # (N, K) bf16
                K = L.Cin
                d = IgemmDesc()
                d.N, d.Ho, d.Wo = N, 1, 1
                d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = xin.shape[1], 0, xin.shape[1], 0
                d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 1, 1, K, L.Cout
                d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = L.Cout, 0, L.Cout, 0
                d.slope = self.SLOPE
                d.out_fp32 = 1
                d.w_blocked = 0 if train else 1
                last = (li == len(self.layers) - 1)
                nk = K // 64
                # blocked panels (inference) run the 3-stage weight-stream kernel: 32 co-tiles x 32 splits = two full rounds of 512 slots
                splits = max(1, min(32 if d.w_blocked else 48, nk // 16)) if K >= 4096 else 1
                b = L.bias.detach()
                if splits > 1:
                    # every K split STORES its partial [N][Cout] result as a slab; the finishing pass adds the slabs in fixed
                    # order -> the forward is bit-reproducible (fp32 atomics of 32 splits were not) and needs no zero fill
                    acc = _splitk_scratch(splits * N * L.Cout, zero=False)
                    d.epilogue, d.split_k, d.split_slabs = EPI_NONE, splits, 1
                    with _timed(f"fc{li}", "igemm", 2.0 * N * L.Cout * L.Cin):
                        _igemm(L_, d, ptr(xin), ptr(wf), None, None, ptr(acc), st, f"igemm fc{li}")
                    yb = torch.empty((N, L.Cout), dtype=torch.bfloat16, device=dev) if not last else None
                    yf = torch.empty((N, L.Cout), dtype=torch.float32, device=dev) if last else None
                    check(L_.yolo_bias_lrelu_rows_slabs(ptr(acc), splits, ptr(b), N, L.Cout, self.SLOPE if L.lrelu else 1.0, ptr(yb), ptr(yf), st),
                          "bias_lrelu_rows")
                else:
                    yf = torch.empty((N, L.Cout), dtype=torch.float32, device=dev)
                    d.epilogue, d.split_k = (EPI_BIAS_LRELU if L.lrelu else EPI_BIAS), 1
                    with _timed(f"fc{li}", "igemm", 2.0 * N * L.Cout * L.Cin):
                        _igemm(L_, d, ptr(xin), ptr(wf), ptr(b), None, ptr(yf), st, f"igemm fc{li}")
                    yb = None
                    if not last:
                        yb = torch.empty((N, L.Cout), dtype=torch.bfloat16, device=dev)
                        check(L_.yolo_cast_f32_to_bf16(ptr(yf), yf.numel(), ptr(yb), st), "cast")
                mask = None
                y_act = yb
                if not last and L.dropout > 0 and drop_training:
                    mask = (torch.rand((N, L.Cout), device=dev) >= L.dropout).to(torch.uint8)
                    yd = torch.empty_like(yb)
                    check(L_.yolo_dropout_bf16(ptr(yb), ptr(mask), 1.0 / (1.0 - L.dropout), yb.numel(), ptr(yd), st), "dropout")
                    cur = yd
                else:
                    cur = yb
                fc_saved[li] = (xin, y_act, mask)
                if last:
                    out = yf
        if out is None:
            out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
            check(L_.yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        saved = (key, ws, fc_saved, N, dev) if train else None
        if not train:
            self._release(key, ws)
        return out, saved

    # ------------------------------------------------------------------ backward
    @staticmethod
    def _codes(ws, li: int, pooled: Act) -> torch.Tensor:
        """arg-max codes of the pool behind conv layer li: uint16 per (pooled pixel, 8 channels), indexed like the pooled map / 8"""
        c = ws["misc"].get(("codes", li))
        if c is None:
            c = torch.empty(pooled.t.numel() // 8, dtype=torch.int16, device=pooled.t.device)
            ws["misc"][("codes", li)] = c
        ws["codes"].add(li)
        return c

    def _grad_buf(self, ws, li: int, N, dev) -> Act:
        """gradient wrt the (post-activation-derivative) output of conv layer li, in the geometry
        yolo_wgrad's flat indexing needs (= the layer's INPUT geometry; zero-stuffed for stride 2)."""
        g = ws["grads"].get(li)
        if g is None:
            L = self.layers[li]
            if L.stride == 1 or L.first:
                g = Act(N, L.Hout, L.Wout, L.Cout, 1, dev)
            else:
                g = Act(N, L.Hin, L.Win, L.Cout, 1, dev)
            ws["grads"][li] = g
        return g

    def _grad_out_strides(self, L: Layer, g: Act):
        """(img, row, px, off) strides a producer uses to write layer L's output gradient into g."""
        if L.stride == 1 or L.first:
            return g.img_stride, g.row_stride, g.px_stride, g.interior_off()
        return g.img_stride, 2 * g.row_stride, 2 * g.px_stride, g.interior_off()

    @staticmethod
    def _wgrad_desc(L: Layer, g: Act, xin: Act, N: int) -> WgradDesc:
        """yolo_wgrad problem of conv layer L over N images of the gradient buffer g / the input buffer xin"""
        # kernel variant: the 256 x 256 pipelined kernel (5) on the big deep layers, where the in-process A/B measured it 10-19 %
        # faster (56x56 256 -> 512, 28x28 512 -> 1024, 14x14 1024 -> 1024: tools/time_wgrad.py); the 128 x 128 kernel (0) elsewhere
        deep = (L.Cout >= 512 and L.Cin >= 256 and N * L.Hout * L.Wout >= 40000) or (L.Cout >= 1024 and L.Cin >= 1024 and N * L.Hout * L.Wout >= 12000)
        variant = 5 if (WGRAD_PIPE and L.K == 3 and L.stride == 1 and deep) else 0
        if variant == 5 or (L.Hout >= 2 and L.Wout >= 2 and (L.stride > 1 or g.Hp * g.Wp >= 1.12 * L.Hout * L.Wout)):
            return WgradDesc(N * L.Hout * L.Wout, g.px_stride, xin.px_stride, L.Cout, L.Cin, L.K, L.K, L.pad, xin.row_stride, 0, 0, variant,
                             L.Wout, L.Hout, g.Hp * g.Wp, g.Wp * L.stride, L.stride, g.halo * g.Wp + g.halo)
        return WgradDesc(N * g.Hp * g.Wp, g.px_stride, xin.px_stride, L.Cout, L.Cin, L.K, L.K, L.pad, xin.row_stride, 0, 0, variant)

    def backward(self, saved, gout: torch.Tensor, need_gx: bool):
        """gout: gradient of the plan output (same shape as forward's out).  Returns (gx or None, [param grads])."""
        L_ = lib()
        st = stream()
        key, ws, fc_saved, N, dev = saved
        self._apply_geom(ws)
        if self.arena is not None:
            self.arena[self._arena_w_end:].zero_()      # bias gradients are accumulated with atomics
        grads: dict[int, tuple] = {}
        nl = len(self.layers)
        # one zero-filled fp32 scratch for the whole pass: the packed conv weight gradients (targets of
        # yolo_wgrad's atomics) and, without an arena, the bias gradients -- one fill instead of ~50
        offs, tot = {}, 0
        for i, L in enumerate(self.layers):
            if L.kind == "conv":
                offs[("w", i)] = tot
                tot += _round_up(L.Cout * 7 * 8 * 4 if L.first else L.Cout * L.K * L.K * L.Cin, 64)
        if self.arena is None:
            for i, L in enumerate(self.layers):
                if L.kind in ("conv", "fc"):
                    offs[("b", i)] = tot
                    tot += _round_up(L.Cout, 64)
        scratch = torch.zeros(tot, dtype=torch.float32, device=dev)

        def grad_tensors(i):
            L = self.layers[i]
            if self.arena is not None:
                dw, db, _, _ = self.arena_views[i]
                return dw, db
            o = offs[("b", i)]
            return torch.empty_like(L.weight, dtype=torch.float32), scratch[o: o + L.Cout]

        # packed -> OIHW conversion of finished conv gradients is deferred and done for several layers per
        # launch (yolo_unpack_conv_wgrads_multi); gradients become final (and are announced) at the flush
        pending: list[tuple] = []
        stem_dpool = None            # pooled gradient handed straight to the stem's weight-gradient kernel (pool backward fused there)

        def flush():
            items = [ConvUnpackItem(dwp.data_ptr(), dw.data_ptr(), L.Cout, L.Cin, L.K, L.K) for (i, L, dwp, dw) in pending if self._multi_ok(L)]
            if items:
                check(L_.yolo_unpack_conv_wgrads_multi((ConvUnpackItem * len(items))(*items), len(items), stream()), "unpack_conv_wgrads_multi")
            for (i, L, dwp, dw) in pending:
                if not self._multi_ok(L):
                    if L.first:
                        check(L_.yolo_unpack_conv_wgrad(ptr(dwp), L.Cout, 3, 7, 7, 4, 8, ptr(dw), 0, stream()), "unpack")
                    else:
                        check(L_.yolo_unpack_conv_wgrad(ptr(dwp), L.Cout, L.Cin, L.K, L.K, L.Cin, L.K, ptr(dw), 0, stream()), "unpack")
                self._layer_done(i)
            pending.clear()

        gout = gout.detach()
        if gout.dtype != torch.float32 or not gout.is_contiguous():
            gout = gout.float().contiguous()

        # The data gradients form the chain every later layer waits for; a layer's weight gradient only needs that layer's output
        # gradient and is first read by the optimizer.  The conv weight gradients (and their unpack passes / gradient-ready
        # callbacks) therefore go to a second stream: their atomic epilogues, partial last rounds and prologues -- phases in which a
        # kernel leaves the matrix cores idle -- overlap with the data-gradient kernels of the layers below, workgroup by workgroup.
        main_t = STREAMS.current(dev)
        side_t = self._side_stream(dev) if (WGRAD_STREAM and TIMERS is None) else None

        def _on_side():
            return _on_side_stream(main_t, side_t, self.on_stream_wait if self.arena is not None else None)

        # what each layer's input activation is
        def input_of(li):
            return ws["in"] if li == 0 else ws["acts"][li - 1]

        # g_cur: gradient flowing into the output of layer li (representation depends on kind)
        g_flat = None       # fp32 (N, K) gradient wrt an fc layer's output / flatten output
        g_act: Act | None = None   # Act gradient wrt a conv/pool output (already through LeakyReLU')
        li = nl - 1
        if self.layers[li].kind == "fc":
            g_flat = gout.reshape(N, -1)
        else:
            # plan ends with feature maps (NCHW fp32 gradient): last layer is a conv(+lrelu) or a pool
            L = self.layers[li]
            y = ws["acts"][li]
            graw = Act(N, y.H, y.W, y.C, 1, dev)
            check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(gout), N, y.C, y.H, y.W, graw.p, y.C, 1, 1, st), "gout->nhwc")
            if L.kind == "conv":
                g = self._grad_buf(ws, li, N, dev)
                self._apply_dlrelu_into(graw, y, L, g, st)
                g_act = g
            else:
                assert L.kind == "pool", "plans end with fc, conv or pool"
                g_act = graw

        while li >= 0:
            L = self.layers[li]
            if L.kind == "fc":
                xin, y_act, mask = fc_saved[li]
                last = (li == nl - 1)
                ldg = _round_up(L.Cout, 32)
                gb = torch.empty((N, ldg), dtype=torch.bfloat16, device=dev)
                # through dropout + LeakyReLU of THIS layer's output (none for the last layer)
                check(L_.yolo_scale_rows_to_bf16(ptr(g_flat), ptr(mask), (1.0 / (1.0 - L.dropout)) if mask is not None else 1.0,
                                                 ptr(y_act) if (L.lrelu and not last) else None, self.SLOPE, N, L.Cout, ldg, ptr(gb), st), "scale_rows")
                # weight / bias gradient, native [O][K] layout
                dw, db = grad_tensors(li)
                wd = WgradDesc(N, ldg, L.Cin, L.Cout, L.Cin, 1, 1, 0, 0, 1, 0)
                nsq = None
                if L.Cout * L.Cin >= FC_NORM_IN_WGRAD and L.Cin % 4 == 0:
                    # the kernel that stores this gradient also sums its squares: the optimizer's global-norm pass (clip_grad_norm_) then
                    # need not read the 822 MB of the Linear behind nn.Flatten again (yolo.optim.grad_norm_sq, `known`)
                    nsq = torch.zeros((), dtype=torch.float64, device=dev)
                    wd.dw_sumsq = nsq.data_ptr()
                with _timed(f"fc{li}.wgrad", "wgrad", 2.0 * N * L.Cout * L.Cin):
                    check(L_.yolo_wgrad(ctypes.byref(wd), ptr(xin), ptr(gb), ptr(dw), ptr(db), st), f"wgrad fc{li}")
                if nsq is not None:
                    # (no reference to dw itself: autograd takes the gradient over without a copy only while nobody else holds it)
                    self.grad_norm_sq[id(L.weight)] = ((dw.data_ptr(), tuple(dw.shape)), dw._version, nsq)
                grads[li] = (dw, db)
                self._layer_done(li)
                # data gradient
                need_prev = li > 0 or need_gx
                behind_flatten = li >= 2 and self.layers[li - 1].kind == "flatten" and self.layers[li - 2].kind in ("conv", "pool")
                if need_prev and behind_flatten:
                    # the Linear behind nn.Flatten (205 M weights): reduce over the OUTPUT features with the
                    # weight-gradient kernel -- both operands are strided along the reduction axis there, which
                    # is exactly how W[o][k] and g^T[o][n] lie in memory -- and read the forward bf16 copy of W:
                    #   dxT[k][n] = sum_o W[o][k] * gT[o][n]
                    Lc = self.layers[li - 2]
                    y = ws["acts"][li - 2]
                    wf, _ = self._pack(li, False)
                    ldn = _round_up(N, 8)
                    gT = torch.zeros((L.Cout, ldn), dtype=torch.bfloat16, device=dev)
                    check(L_.yolo_transpose_bf16(ptr(gb), N, L.Cout, ldg, ptr(gT), ldn, st), "transpose g")
                    dxT = torch.zeros((L.Cin, N), dtype=torch.float32, device=dev)
                    wd = WgradDesc(L.Cout, L.Cin, ldn, L.Cin, N, 1, 1, 0, 0, 0, 1)   # split 0: library's schedule (0.095 vs 0.135 ms with 3 ranges)
                    with _timed(f"fc{li}.dgrad", "wgrad", 2.0 * N * L.Cout * L.Cin):
                        check(L_.yolo_wgrad(ctypes.byref(wd), ptr(gT), ptr(wf), ptr(dxT), None, st), f"dgrad fc{li}")
                    if Lc.kind == "conv":
                        assert Lc.stride == 1, "nn.Flatten is expected after a stride-1 conv or a pool"
                        g = self._grad_buf(ws, li - 2, N, dev)
                        yact = y.p if Lc.lrelu else None
                    else:
                        g = ws["misc"].get("graw_flat")
                        if g is None:
                            g = Act(N, y.H, y.W, y.C, 1, dev)
                            ws["misc"]["graw_flat"] = g
                        yact = None
                    check(L_.yolo_fc_dgrad_to_nhwc(ptr(dxT), N, y.C, y.H, y.W, 1, yact, self.SLOPE, g.p, st), "fc_dgrad_to_nhwc")
                    g_act = g
                    g_flat = None
                    li -= 2          # nn.Flatten is done as well
                    continue
                if need_prev:
                    _, wt = self._pack(li, True)
                    d = IgemmDesc()
                    d.N, d.Ho, d.Wo = N, 1, 1
                    d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = ldg, 0, ldg, 0
                    d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 1, 1, ldg, L.Cin
                    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = L.Cin, 0, L.Cin, 0
                    d.epilogue, d.slope, d.out_fp32, d.split_k = EPI_NONE, self.SLOPE, 1, 1
                    gprev = torch.empty((N, L.Cin), dtype=torch.float32, device=dev)
                    with _timed(f"fc{li}.dgrad", "igemm", 2.0 * N * L.Cout * L.Cin):
                        _igemm(L_, d, ptr(gb), ptr(wt), None, None, ptr(gprev), st, f"dgrad fc{li}")
                    g_flat = gprev
                li -= 1
            elif L.kind == "flatten":
                raise AssertionError("nn.Flatten is handled together with the Linear layer behind it")
            elif L.kind == "pool":
                # g_act = gradient wrt the pooled output; produce gradient wrt the conv in front
                lc = li - 1
                Lc = self.layers[lc]
                assert Lc.kind == "conv" and Lc.lrelu, "MaxPool2d is expected right after conv+LeakyReLU"
                yfull = ws["acts"][lc]
                if (lc == 0 and Lc.first and STEM_POOL_BWD_FUSED and Lc.Cout == 64 and Lc.Hout % 8 == 0 and Lc.Wout % 16 == 0 and g_act.halo == 1
                        and not need_gx):      # (a gradient wrt the input image needs the stem's output gradient as a tensor)
                    # the stem's weight-gradient kernel rebuilds this pool's (+ LeakyReLU's) backward per tile from the
                    # activation and the pooled gradient: the 224x224x64 gradient buffer is never written or read
                    stem_dpool = g_act
                    li -= 1
                    continue
                g = self._grad_buf(ws, lc, N, dev)
                pd = PoolDesc(N, yfull.H, yfull.W, yfull.C, 1, 1)
                with _timed(f"pool{li}.bwd", "maxpool2_bwd"):
                    if lc in ws.get("codes", ()):
                        ypool = ws["acts"][li]
                        assert (ypool.Hp, ypool.Wp, ypool.C, ypool.halo) == (g_act.Hp, g_act.Wp, g_act.C, g_act.halo)
                        check(L_.yolo_maxpool2_bwd_codes(ctypes.byref(pd), ypool.p, ptr(ws["misc"][("codes", lc)]), g_act.p, self.SLOPE, g.p, st), "maxpool_bwd_codes")
                    else:
                        check(L_.yolo_maxpool2_bwd_lrelu(ctypes.byref(pd), yfull.p, g_act.p, self.SLOPE, g.p, st), "maxpool_bwd")
                g_act = g
                li -= 1
            elif L.kind == "conv":
                g = g_act  # dZ of this layer, flat-geometry buffer
                xin = input_of(li)
                # ---- weight + bias gradient
                dw, db = grad_tensors(li)
                with _on_side() as wst:
                    o = offs[("w", li)]
                    stem_direct = L.first and L.Cout == 64 and L.Hout % 8 == 0 and L.Wout % 16 == 0
                    if stem_direct:
                        part = ws["misc"].get("stem_part")
                        if part is None:
                            part = torch.empty((768 * 14400,), dtype=torch.float32, device=dev)
                            ws["misc"]["stem_part"] = part
                        with _timed(f"conv{li}.wgrad", "wgrad", 2.0 * N * L.Hout * L.Wout * L.Cout * 147):
                            if stem_dpool is not None and 0 in ws.get("codes", ()):
                                yp = ws["acts"][1]
                                assert (yp.Hp, yp.Wp, yp.C, yp.halo) == (stem_dpool.Hp, stem_dpool.Wp, stem_dpool.C, stem_dpool.halo)
                                check(L_.yolo_wgrad_stem7_codes(xin.p, yp.p, ptr(ws["misc"][("codes", 0)]), N, L.Hout, L.Wout, xin.img_stride, xin.row_stride,
                                                                stem_dpool.p, stem_dpool.img_stride, stem_dpool.row_stride, stem_dpool.interior_off(),
                                                                self.SLOPE if L.lrelu else 1.0, ptr(dw), ptr(db), ptr(part), part.numel(), wst), "wgrad_stem7_codes")
                            elif stem_dpool is not None:
                                yf = ws["acts"][0]
                                check(L_.yolo_wgrad_stem7_pooled(xin.p, yf.p, N, L.Hout, L.Wout, xin.img_stride, xin.row_stride, yf.img_stride, yf.row_stride,
                                                                 yf.interior_off(), stem_dpool.p, stem_dpool.img_stride, stem_dpool.row_stride,
                                                                 stem_dpool.interior_off(), self.SLOPE if L.lrelu else 1.0, ptr(dw), ptr(db), ptr(part),
                                                                 part.numel(), wst), "wgrad_stem7_pooled")
                            else:
                                check(L_.yolo_wgrad_stem7(xin.p, g.p, N, L.Hout, L.Wout, xin.img_stride, xin.row_stride, g.img_stride, g.row_stride,
                                                          g.interior_off(), ptr(dw), ptr(db), ptr(part), part.numel(), wst), "wgrad_stem7")
                        grads[li] = (dw, db)
                        flush()
                        self._layer_done(li)
                    elif L.first:
                        xcol = ws["misc"].get("xcol")
                        if xcol is None:
                            xcol = Act(N, L.Hout, L.Wout, 7 * 32, 1, dev)
                            ws["misc"]["xcol"] = xcol
                        check(L_.yolo_im2col_rows(xin.p, xin.img_stride, xin.row_stride, xin.px_stride, 2, 7, 32, N, L.Hout, L.Wout, 1, xcol.p, wst), "im2col_rows")
                        dwp = scratch[o: o + L.Cout * 7 * 8 * 4]
                        split = max(1, min(1024, g.slots // 4096))
                        wd = WgradDesc(g.slots, g.px_stride, xcol.px_stride, L.Cout, 7 * 32, 1, 1, 0, xcol.row_stride, split, 0)
                        with _timed(f"conv{li}.wgrad", "wgrad", 2.0 * N * L.Hout * L.Wout * L.Cout * 147):
                            check(L_.yolo_wgrad(ctypes.byref(wd), xcol.p, g.p, ptr(dwp), ptr(db), wst), "wgrad conv0")
                    else:
                        dwp = scratch[o: o + L.Cout * L.K * L.K * L.Cin]
                        # reduce over the layer's OUTPUT pixels only (not over every slot of the zero-haloed -- for stride 2
                        # zero-stuffed -- gradient buffer, whose geometry the input buffer shares slot for slot)
                        # (measured: worth it from 28x28 down and for stride 2; at 56x56 and above the halo is < 8 % of the slots
                        # and the per-row coordinate arithmetic costs more than it saves)
                        wd = self._wgrad_desc(L, g, xin, N)
                        if WGRAD_SLABS and wd.variant == 5:
                            _attach_wgrad_slabs(L_, wd, dev)
                        with _timed(f"conv{li}.wgrad", "wgrad", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                            check(L_.yolo_wgrad(ctypes.byref(wd), xin.p, g.p, ptr(dwp), ptr(db), wst), f"wgrad conv{li}")
                    if not stem_direct:
                        grads[li] = (dw, db)
                        pending.append((li, L, dwp, dw))
                        if li == 0 or sum(t[2].numel() for t in pending) >= (16 << 20):
                            flush()
                # ---- data gradient
                if li == 0:
                    if side_t is not None:
                        main_t.wait_stream(side_t)       # every weight gradient is final before anything that follows the backward pass
                        if self.arena is not None and self.on_stream_wait is not None:
                            self.on_stream_wait(main_t.cuda_stream, side_t.cuda_stream)
                    gx = None
                    if need_gx:
                        gx = self._stem_dgrad(li, g, N, dev, st) if L.first else self._dgrad_to_input(li, g, N, dev, st)
                    if self.debug_keep:
                        self.last = (ws, fc_saved)
                    else:
                        self._release(key, ws)
                    if self.arena is not None:
                        if self.on_backward_done is not None:
                            self.on_backward_done()
                        for i in grads:          # hand the views to the optimizer without going through autograd
                            L2 = self.layers[i]
                            if L2.weight.grad is not grads[i][0]:
                                L2.weight.grad = grads[i][0]
                            if L2.bias.grad is not grads[i][1]:
                                L2.bias.grad = grads[i][1]
                        return gx, [None] * (2 * len(grads))
                    return gx, [grads[i][j] for i in sorted(grads) for j in (0, 1)]
                _, wdg = self._pack(li, True)
                prev = self.layers[li - 1]
                d = IgemmDesc()
                d.N, d.Ho, d.Wo = N, L.Hin, L.Win            # gradient grid = this layer's input grid
                d.in_img_stride, d.in_row_stride, d.in_px_stride = g.img_stride, g.row_stride, g.px_stride
                d.in_off = g.interior_off(L.K - 1 - L.pad)
                d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, L.K, L.K, L.Cout, L.Cin
                d.slope, d.out_fp32, d.split_k = self.SLOPE, 0, 1
                d.tile_hint = TILE_HINT
                if (prev.kind == "conv" and STRIDE2_CLASSES and L.stride == 2 and L.K == 3 and L.pad == 1 and prev.stride == 1
                        and L.Hin % 2 == 0 and L.Win % 2 == 0):
                    # stride-2 3x3 conv: the gradient buffer g holds dy zero-stuffed to the input grid, and the plain data gradient
                    # spends 3/4 of its MACs on those zeros.  By input-pixel parity (py, px) only the taps ky = 1 (py even) or
                    # ky = 2, 0 (py odd; likewise kx) contribute: four small convs over the NON-ZERO slots (doubled input strides)
                    # with 1, 2, 2 and 4 taps -- 9 taps per 2x2 input pixels instead of 36 -- each writing its parity class of the
                    # previous layer's gradient (doubled output strides).
                    gp = self._grad_buf(ws, li - 1, N, dev)
                    yprev = ws["acts"][li - 1]
                    panels = self._stride2_panels(li, wdg)
                    with _timed(f"conv{li}.dgrad", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        for (py, px), wc in panels.items():
                            dc = IgemmDesc()
                            dc.N, dc.Ho, dc.Wo = N, L.Hin // 2, L.Win // 2
                            dc.in_img_stride, dc.in_row_stride, dc.in_px_stride, dc.in_off = g.img_stride, 2 * g.row_stride, 2 * g.px_stride, g.interior_off()
                            dc.stride, dc.KH, dc.KW, dc.tap_len, dc.Cout = 1, 1 + py, 1 + px, L.Cout, L.Cin
                            dc.slope, dc.out_fp32, dc.split_k, dc.tile_hint = self.SLOPE, 0, 1, TILE_HINT
                            dc.out_img_stride, dc.out_row_stride, dc.out_px_stride = gp.img_stride, 2 * gp.row_stride, 2 * gp.px_stride
                            dc.out_off = gp.interior_off() + py * gp.row_stride + px * gp.px_stride
                            aux = None
                            dc.epilogue = EPI_NONE
                            if prev.lrelu:
                                dc.epilogue = EPI_MUL_DLRELU
                                dc.aux_img_stride, dc.aux_row_stride, dc.aux_px_stride = yprev.img_stride, 2 * yprev.row_stride, 2 * yprev.px_stride
                                dc.aux_off = yprev.interior_off() + py * yprev.row_stride + px * yprev.px_stride
                                aux = yprev.p
                            igemm_call(dc, g.p, ptr(wc), None, aux, gp.p, st, f"dgrad conv{li} class {py}{px}")
                    g_act = gp
                elif prev.kind == "conv":
                    gp = self._grad_buf(ws, li - 1, N, dev)
                    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = self._grad_out_strides(prev, gp)
                    yprev = ws["acts"][li - 1]
                    if prev.lrelu:
                        d.epilogue = EPI_MUL_DLRELU
                        d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = yprev.img_stride, yprev.row_stride, yprev.px_stride, yprev.interior_off()
                        aux = yprev.p
                    else:
                        d.epilogue, aux = EPI_NONE, None
                    with _timed(f"conv{li}.dgrad", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        igemm_call(d, g.p, ptr(wdg), None, aux, gp.p, st, f"dgrad conv{li}")
                    g_act = gp
                elif prev.kind == "pool":
                    gp = ws["misc"].get(("gpool", li))
                    if gp is None:
                        gp = Act(N, L.Hin, L.Win, L.Cin, 1, dev)
                        ws["misc"][("gpool", li)] = gp
                    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = gp.img_stride, gp.row_stride, gp.px_stride, gp.interior_off()
                    d.epilogue = EPI_NONE
                    with _timed(f"conv{li}.dgrad", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        igemm_call(d, g.p, ptr(wdg), None, None, gp.p, st, f"dgrad conv{li}")
                    g_act = gp
                else:
                    raise AssertionError("conv after flatten/fc")
                li -= 1
        raise AssertionError("unreachable")

    def _apply_dlrelu_into(self, graw: Act, y: Act, L: Layer, g: Act, st):
        """g(interior, possibly zero-stuffed) = graw * lrelu'(y) -- used only at plan ends (rare path)."""
        gi = graw.interior().float()
        if L.lrelu:
            gi = gi * torch.where(y.interior().float() > 0, 1.0, self.SLOPE)
        gi = gi.to(torch.bfloat16)
        if L.stride == 1 or L.first:
            g.interior().copy_(gi)
        else:
            g.interior()[:, 0::2, 0::2, :][:, : gi.shape[1], : gi.shape[2], :].copy_(gi)

    def _stem_dgrad(self, li, g: Act, N, dev, st):
        """gradient wrt the input IMAGE through the 7x7 / stride-2 / pad-3 stem (the reference back-propagates to x in
        tests/test_backbone.py:187-196; training never asks for it, so the step's FLOP count skips this product).  By the parity
        (py, px) of the image pixel (y, x) = (2a + py, 2b + px) only the taps ky = py + 5 - 2 ty (ty = 0 .. 2 + py; likewise kx) meet an
        output pixel, (a + ty - 1, b + tx - 1): four stride-1 correlations over the stem's output gradient g with 3x3, 3x4, 4x3 and 4x4
        taps of 64 channels, each writing its parity class of the image (doubled output strides) -- the scheme of the stride-2 3x3
        layers' data gradient.  The three image channels ride in an 8-channel fp32 NHWC scratch; rows / columns a + 2 past the map fall
        on the zero halo of the next row / image (or the guard band)."""
        L = self.layers[li]
        assert L.first and L.K == 7 and L.stride == 2 and L.pad == 3 and g.halo == 1 and g.C == L.Cout and L.Cout % 64 == 0
        H, W = 2 * L.Hout, 2 * L.Wout
        w = L.weight.detach().float()                                   # [Cout][3][7][7]
        buf = torch.empty((N, H, W, 8), dtype=torch.float32, device=dev)
        L_ = lib()
        for py in (0, 1):
            for px in (0, 1):
                kys = [py + 5 - 2 * t for t in range(3 + py)]
                kxs = [px + 5 - 2 * t for t in range(3 + px)]
                panel = torch.zeros((8, len(kys), len(kxs), L.Cout), dtype=torch.bfloat16, device=dev)
                panel[:3] = w[:, :, kys][:, :, :, kxs].permute(1, 2, 3, 0).to(torch.bfloat16)      # [c][ty][tx][co]
                d = IgemmDesc()
                d.N, d.Ho, d.Wo = N, L.Hout, L.Wout
                d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = g.img_stride, g.row_stride, g.px_stride, g.interior_off(1)
                d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, len(kys), len(kxs), L.Cout, 8
                d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = H * W * 8, 2 * W * 8, 16, (py * W + px) * 8
                d.epilogue, d.slope, d.out_fp32, d.split_k, d.tile_hint = EPI_NONE, self.SLOPE, 1, 1, 4      # 64 x 128 tiles: 8 "channels"
                _igemm(L_, d, g.p, ptr(panel), None, None, ptr(buf), st, f"stem dgrad class {py}{px}")
        return buf[..., :3].permute(0, 3, 1, 2).contiguous()

    def _dgrad_to_input(self, li, g: Act, N, dev, st):
        """data gradient of the first conv of a plan whose input is a feature map (DetectionHead)."""
        L = self.layers[li]
        _, wdg = self._pack(li, True)
        gi = Act(N, L.Hin, L.Win, L.Cin, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, L.Hin, L.Win
        d.in_img_stride, d.in_row_stride, d.in_px_stride = g.img_stride, g.row_stride, g.px_stride
        d.in_off = g.interior_off(L.K - 1 - L.pad)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, L.K, L.K, L.Cout, L.Cin
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = gi.img_stride, gi.row_stride, gi.px_stride, gi.interior_off()
        d.epilogue, d.slope, d.out_fp32, d.split_k = EPI_NONE, self.SLOPE, 0, 1
        _igemm(lib(), d, g.p, ptr(wdg), None, None, gi.p, st, "dgrad input")
        gx = torch.empty((N, L.Cin, L.Hin, L.Win), dtype=torch.float32, device=dev)
        check(lib().yolo_nhwc_bf16_to_nchw_f32(gi.p, N, L.Cin, L.Hin, L.Win, 1, ptr(gx), st), "gx nhwc->nchw")
        return gx


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
    def forward(ctx, plan, x: torch.Tensor, *params):
        out, saved = plan.forward_train(x)
        ctx.plan, ctx.saved, ctx.params = plan, saved, params
        return out

    @staticmethod
    @_hip.device_guard
    def backward(ctx, gout):
        if ctx.saved is None:
            raise RuntimeError("backward through a ResNet trunk forward that was already consumed")
        grads = ctx.plan.backward_train(ctx.saved, gout)
        ctx.saved = None
        return (None, None) + tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)


@_hip.device_guard
def run_plan(plan: Plan, x: torch.Tensor, drop_training: bool) -> torch.Tensor:
    _hip.require_cuda(x)
    # grad mode must be sampled here: inside Function.forward it is always off
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in plan.params))
    return PlanFunction.apply(plan, drop_training, need, x, *plan.params)


# ====================================================================================================
# ResNet-50 trunk, inference only (BatchNorm folded into the conv that precedes it)
# ====================================================================================================
class ResNetPlan:
    """Inference executor for ``yolo.resnet.resnet50_trunk`` on the same kernels: every conv+BN(+ReLU) is
    one yolo_igemm (BN folded into the bf16 weights and an fp32 bias at pack time), the residual add + ReLU
    of a bottleneck is the epilogue of its last 1x1 conv (YOLO_EPI_BIAS_ADD_LRELU with slope 0), the stem's
    MaxPool2d(3,2,1) is yolo_maxpool3s2_fwd.  ``forward_batch_stats`` runs the same trunk with BatchNorm in training mode
    (batch statistics: conv with the raw weights, then yolo_batchnorm_train_fwd) for the FROZEN backbone of a training run;
    ``forward_train`` / ``backward_train`` are the trainable trunk of the reference's default run (src/train.py:144)."""

    def __init__(self, trunk: nn.Sequential):
        self.trunk = trunk
        self._packed = None
        self._raw = None
        self._bn_scratch = None
        self._bufs: dict = {}
        self.trace = None            # tests: a list that backward_train fills with per-block gradients

    # -- BN folding: y = gamma * (conv(x) - mean) / sqrt(var + eps) + beta
    @staticmethod
    def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d):
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = conv.weight.detach().float() * scale.view(-1, 1, 1, 1)
        b = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        if conv.bias is not None:
            b = b + conv.bias.detach().float() * scale
        return w.contiguous(), b.contiguous()

    def _pack_all(self):
        ver = tuple(int(p._version) for p in self.trunk.parameters()) + tuple(int(b._version) for b in self.trunk.buffers())
        if self._packed is not None and self._packed[0] == ver:
            return self._packed[1]
        st = stream()
        out = {}

        def pack(name, conv, bn, first=False):
            w, b = self._fold(conv, bn)
            co, ci, k, _ = w.shape
            if first:
                wf = torch.empty((co, 7, 8, 4), dtype=torch.bfloat16, device=w.device)
                check(lib().yolo_pack_conv_weight(ptr(w), co, 3, 7, 7, 4, 8, ptr(wf), None, st), "pack stem")
            else:
                wf = torch.empty((co, k, k, ci), dtype=torch.bfloat16, device=w.device)
                check(lib().yolo_pack_conv_weight(ptr(w), co, ci, k, k, ci, k, ptr(wf), None, st), "pack")
            out[name] = (wf, b, conv)

        pack("stem", self.trunk[0], self.trunk[1], first=True)
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                pack((li, bi, 1), blk.conv1, blk.bn1)
                pack((li, bi, 2), blk.conv2, blk.bn2)
                pack((li, bi, 3), blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    pack((li, bi, "d"), blk.downsample[0], blk.downsample[1])
        self._packed = (ver, out)
        return out

    def _pack_raw(self):
        """bf16 operands of the UN-folded conv weights (batch-statistics mode: BatchNorm cannot be folded)."""
        ver = tuple(int(p._version) for n, p in self.trunk.named_parameters() if p.dim() == 4)
        if self._raw is not None and self._raw[0] == ver:
            return self._raw[1]
        st = stream()
        out = {}

        def pack(name, conv, bn, first=False):
            w = conv.weight.detach().float().contiguous()
            co, ci, k, _ = w.shape
            if first:
                wf = torch.empty((co, 7, 8, 4), dtype=torch.bfloat16, device=w.device)
                check(lib().yolo_pack_conv_weight(ptr(w), co, 3, 7, 7, 4, 8, ptr(wf), None, st), "pack stem")
            else:
                wf = torch.empty((co, k, k, ci), dtype=torch.bfloat16, device=w.device)
                check(lib().yolo_pack_conv_weight(ptr(w), co, ci, k, k, ci, k, ptr(wf), None, st), "pack")
            out[name] = (wf, None, conv, bn)

        pack("stem", self.trunk[0], self.trunk[1], first=True)
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                pack((li, bi, 1), blk.conv1, blk.bn1)
                pack((li, bi, 2), blk.conv2, blk.bn2)
                pack((li, bi, 3), blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    pack((li, bi, "d"), blk.downsample[0], blk.downsample[1])
        self._raw = (ver, out)
        return out

    def _bn_train(self, a: Act, bn: nn.BatchNorm2d, relu: bool, residual: Act | None, dev, st, out: Act | None = None, save: torch.Tensor | None = None,
                  stats_ready: bool = False):
        """BatchNorm with batch statistics (+ residual, + ReLU) in place on the conv output, running statistics updated."""
        C = a.C
        acc, ss = self._scratch(dev)
        if C > 2048 or bn.weight is None or not bn.track_running_stats:
            raise NotImplementedError("batch-statistics BatchNorm: affine layers with running statistics and C <= 2048")
        mom = 0.1 if bn.momentum is None else bn.momentum
        check(lib().yolo_batchnorm_train_fwd(a.p, a.N, a.H, a.W, C, a.halo, ptr(bn.weight.detach()), ptr(bn.bias.detach()), float(bn.eps), float(mom),
                                             ptr(bn.running_mean), ptr(bn.running_var), residual.p if residual is not None else None,
                                             residual.halo if residual is not None else 0, 1 if relu else 0, ptr(acc), ptr(ss),
                                             out.p if out is not None else None, out.halo if out is not None else 0,
                                             ptr(save) if save is not None else None, 1 if stats_ready else 0, st), "batchnorm_train_fwd")
        bn.num_batches_tracked += 1

    def _scratch(self, dev):
        if self._bn_scratch is None or self._bn_scratch[0].device != dev:
            self._bn_scratch = (torch.zeros(_hip.BN_ACC_REPLICAS * 2 * 2048, dtype=torch.float64, device=dev), torch.empty(2 * 2048, dtype=torch.float32, device=dev))
        return self._bn_scratch

    def _conv_bn_train(self, tag, a_in: Act, packed, N, relu: bool, residual: Act | None, dev, st):
        wf, _, conv, bn = packed
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (a_in.H + 2 * p - k) // s + 1, (a_in.W + 2 * p - k) // s + 1
        a_out = self._act(tag, N, Ho, Wo, conv.out_channels, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.in_off = a_in.interior_off(p)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = s, k, k, conv.in_channels, conv.out_channels
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = a_out.img_stride, a_out.row_stride, a_out.px_stride, a_out.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        d.bn_stats = self._scratch(dev)[0].data_ptr() if BN_STATS_IN_CONV else None      # the conv's epilogue accumulates BatchNorm's sums
        with _timed(str(tag), "igemm", 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * k * k):
            igemm_call(d, a_in.p, ptr(wf), None, None, a_out.p, st, f"igemm {tag}")
        self._bn_train(a_out, bn, relu, residual, dev, st, stats_ready=BN_STATS_IN_CONV)
        return a_out

    @_hip.device_guard
    def forward_batch_stats(self, x: torch.Tensor) -> torch.Tensor:
        """the trunk with its BatchNorm layers in TRAINING mode (batch statistics, running statistics updated) -- the frozen
        backbone of the reference's default training run (trainer.py:49).  Forward only: no gradient flows into the trunk."""
        _hip.require_cuda(x)
        st = stream()
        pk = self._pack_raw()
        N, _, H, W = x.shape
        dev = x.device
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        a = self._act("in", N, H, W, 4, 3, dev)
        check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, 3, H, W, a.p, 4, 3, 3, st), "nchw->nhwc4")
        wf, _, conv, bn = pk["stem"]
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        s1 = self._act("stem", N, Ho, Wo, 64, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a.img_stride, a.row_stride, a.px_stride, 0
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 2, 7, 1, 32, 64
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = s1.img_stride, s1.row_stride, s1.px_stride, s1.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        _igemm(lib(), d, a.p, ptr(wf), None, None, s1.p, st, "igemm stem")
        self._bn_train(s1, bn, True, None, dev, st)
        Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        cur = self._act("pool", N, Hq, Wq, 64, 1, dev)
        pd = PoolDesc(N, Ho, Wo, 64, 1, 1)
        check(lib().yolo_maxpool3s2_fwd(ctypes.byref(pd), s1.p, cur.p, st), "maxpool3s2")
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                idn = cur if blk.downsample is None else self._conv_bn_train((li, bi, "d"), cur, pk[(li, bi, "d")], N, False, None, dev, st)
                t = self._conv_bn_train((li, bi, 1), cur, pk[(li, bi, 1)], N, True, None, dev, st)
                t = self._conv_bn_train((li, bi, 2), t, pk[(li, bi, 2)], N, True, None, dev, st)
                cur = self._conv_bn_train((li, bi, 3), t, pk[(li, bi, 3)], N, True, idn, dev, st)
        out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
        check(lib().yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        return out

    # ------------------------------------------------------------------ trainable trunk (forward keeps z, backward)
    def _pack_train(self):
        """bf16 forward AND data-gradient operands of the raw conv weights (the stem needs no data gradient); refreshed with
        yolo_pack_conv_weights_multi, 32 layers per launch, whenever a weight changed"""
        ver = tuple(int(p._version) for n, p in self.trunk.named_parameters() if p.dim() == 4)
        if getattr(self, "_train_pk", None) is not None and self._train_pk[0] == ver:
            return self._train_pk[1]
        st = stream()
        out = self._train_pk[1] if getattr(self, "_train_pk", None) is not None else {}
        items = []

        def pack(name, conv, bn, first=False):
            w = conv.weight.detach()
            if w.dtype != torch.float32 or not w.is_contiguous():
                w = w.float().contiguous()
            co, ci, k, _ = w.shape
            if name in out:
                wf, wd = out[name][0], out[name][1]
            elif first:
                wf, wd = torch.empty((co, 7, 8, 4), dtype=torch.bfloat16, device=w.device), None
            else:
                wf = torch.empty((co, k, k, ci), dtype=torch.bfloat16, device=w.device)
                wd = torch.empty((ci, k, k, co), dtype=torch.bfloat16, device=w.device)
            if first:
                check(lib().yolo_pack_conv_weight(ptr(w), co, 3, 7, 7, 4, 8, ptr(wf), None, st), "pack stem")
            elif co % 64 == 0 and ci % 64 == 0:
                items.append((ConvPackItem(w.data_ptr(), wf.data_ptr(), wd.data_ptr(), co, ci, k, k), w))
            else:
                check(lib().yolo_pack_conv_weight(ptr(w), co, ci, k, k, ci, k, ptr(wf), ptr(wd), st), "pack")
            out[name] = (wf, wd, conv, bn)

        pack("stem", self.trunk[0], self.trunk[1], first=True)
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                pack((li, bi, 1), blk.conv1, blk.bn1)
                pack((li, bi, 2), blk.conv2, blk.bn2)
                pack((li, bi, 3), blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    pack((li, bi, "d"), blk.downsample[0], blk.downsample[1])
        for i in range(0, len(items), 32):
            tab = (ConvPackItem * len(items[i: i + 32]))(*[it[0] for it in items[i: i + 32]])
            check(lib().yolo_pack_conv_weights_multi(tab, len(items[i: i + 32]), st), "pack_conv_weights_multi")
        self._train_pk = (ver, out)
        return out

    def _unit_fwd(self, tag, a_in: Act, packed, N, relu: bool, residual: Act | None, stats: torch.Tensor, dev, st):
        """conv -> z (kept) -> BatchNorm(batch statistics) [+ residual] [ReLU] -> y; returns the record the backward needs."""
        wf, wd, conv, bn = packed
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (a_in.H + 2 * p - k) // s + 1, (a_in.W + 2 * p - k) // s + 1
        z = self._act((tag, "z"), N, Ho, Wo, conv.out_channels, 1, dev)
        y = self._act((tag, "y"), N, Ho, Wo, conv.out_channels, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.in_off = a_in.interior_off(p)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = s, k, k, conv.in_channels, conv.out_channels
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = z.img_stride, z.row_stride, z.px_stride, z.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        d.bn_stats = self._scratch(dev)[0].data_ptr() if BN_STATS_IN_CONV else None      # the conv's epilogue accumulates BatchNorm's sums
        with _timed(str(tag), "igemm", 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * k * k):
            igemm_call(d, a_in.p, ptr(wf), None, None, z.p, st, f"igemm {tag}")
        self._bn_train(z, bn, relu, residual, dev, st, out=y, save=stats, stats_ready=BN_STATS_IN_CONV)
        return {"tag": tag, "conv": conv, "bn": bn, "x": a_in, "z": z, "y": y, "relu": relu, "res": residual is not None, "stats": stats, "wd": wd,
                "k": k, "s": s, "p": p}

    @_hip.device_guard
    def forward_train(self, x: torch.Tensor):
        """Training-mode forward of a TRAINABLE trunk (the reference's default run, src/train.py:144: ResNetBackbone(freeze=False)):
        as forward_batch_stats, but every unit keeps its conv output z, its activation y and the batch mean / invstd.
        Returns (out, saved).  One forward may be in flight per plan (the buffers are reused step to step)."""
        _hip.require_cuda(x)
        st = stream()
        pk = self._pack_train()
        N, _, H, W = x.shape
        dev = x.device
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        if Ho % 8 or Wo % 16:
            raise NotImplementedError("trainable ResNet trunk: the stem's weight-gradient kernel needs an input of (16k) x (32k) pixels")
        nstat = 4 * (64 + sum(u.num_features for u in self.trunk.modules() if isinstance(u, nn.BatchNorm2d)))
        if getattr(self, "_stats", None) is None or self._stats.numel() < nstat or self._stats.device != dev:
            self._stats = torch.empty(nstat, dtype=torch.float32, device=dev)
        cursor = [0]

        def stat(C):
            t = self._stats[cursor[0]: cursor[0] + 4 * C]
            cursor[0] += 4 * C
            return t

        a = self._act("in", N, H, W, 4, 3, dev)
        check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, 3, H, W, a.p, 4, 3, 3, st), "nchw->nhwc4")
        wf, _, conv, bn = pk["stem"]
        z0 = self._act(("stem", "z"), N, Ho, Wo, 64, 1, dev)
        y0 = self._act(("stem", "y"), N, Ho, Wo, 64, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a.img_stride, a.row_stride, a.px_stride, 0
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 2, 7, 1, 32, 64
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = z0.img_stride, z0.row_stride, z0.px_stride, z0.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        _igemm(lib(), d, a.p, ptr(wf), None, None, z0.p, st, "igemm stem")
        stem = {"tag": "stem", "conv": conv, "bn": bn, "x": a, "z": z0, "y": y0, "relu": True, "res": False, "stats": stat(64)}
        self._bn_train(z0, bn, True, None, dev, st, out=y0, save=stem["stats"])
        Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        cur = self._act("pool", N, Hq, Wq, 64, 1, dev)
        pd = PoolDesc(N, Ho, Wo, 64, 1, 1)
        check(lib().yolo_maxpool3s2_fwd(ctypes.byref(pd), y0.p, cur.p, st), "maxpool3s2")
        blocks = []
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                ud = None
                idn = cur
                if blk.downsample is not None:
                    ud = self._unit_fwd((li, bi, "d"), cur, pk[(li, bi, "d")], N, False, None, stat(blk.downsample[1].num_features), dev, st)
                    idn = ud["y"]
                u1 = self._unit_fwd((li, bi, 1), cur, pk[(li, bi, 1)], N, True, None, stat(blk.bn1.num_features), dev, st)
                u2 = self._unit_fwd((li, bi, 2), u1["y"], pk[(li, bi, 2)], N, True, None, stat(blk.bn2.num_features), dev, st)
                u3 = self._unit_fwd((li, bi, 3), u2["y"], pk[(li, bi, 3)], N, True, idn, stat(blk.bn3.num_features), dev, st)
                blocks.append((li, bi, u1, u2, u3, ud))
                cur = u3["y"]
        out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
        check(lib().yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        self._train_gen = getattr(self, "_train_gen", 0) + 1
        return out, {"N": N, "dev": dev, "stem": stem, "blocks": blocks, "out": cur, "gen": self._train_gen}

    def backward_train(self, saved, gout: torch.Tensor) -> dict:
        """gradients of every trunk parameter for the forward recorded in `saved`: {parameter: fp32 gradient}."""
        L_ = lib()
        st = stream()
        N, dev = saved["N"], saved["dev"]
        if saved["gen"] != self._train_gen:
            raise RuntimeError("ResNetPlan: a later training forward has reused this forward's activation buffers -- call backward() "
                               "before the next forward of the same backbone (one forward in flight per plan)")
        acc, _ = self._bn_scratch
        if getattr(self, "_coef", None) is None or self._coef.device != dev:
            self._coef = torch.empty(3 * 2048, dtype=torch.float32, device=dev)
            self._zero_bias = torch.zeros(2048, dtype=torch.float32, device=dev)
        grads: dict = {}
        convs = [u["conv"] for b in saved["blocks"] for u in b[2:] if u is not None]
        offs, tot = {}, 0
        for c in convs:
            offs[id(c)] = tot
            tot += _round_up(c.weight.numel(), 64)
        scratch = torch.zeros(tot, dtype=torch.float32, device=dev)
        pending = []
        # weight gradients (and their unpack passes) on the low-priority second stream, beside the BatchNorm-backward / data-gradient
        # chain (Plan.backward does the same): the chain's HBM-bound BatchNorm passes and the MFMA-bound weight gradients mix well
        main_t = torch.cuda.current_stream(dev)
        side_t = Plan._side_stream(dev) if (WGRAD_STREAM and TIMERS is None) else None

        def flush():
            with _on_side_stream(main_t, side_t):
                _flush()

        def _flush():
            for i in range(0, len(pending), 32):
                items = [ConvUnpackItem(dwp.data_ptr(), dw.data_ptr(), c.out_channels, c.in_channels, c.kernel_size[0], c.kernel_size[1])
                         for (c, dwp, dw) in pending[i: i + 32]]
                check(L_.yolo_unpack_conv_wgrads_multi((ConvUnpackItem * len(items))(*items), len(items), stream()), "unpack_conv_wgrads_multi")
            pending.clear()

        def bn_bwd(u, dy: Act, store_masked: bool) -> Act:
            """dz of unit u from the gradient dy wrt its output, in the geometry of the conv's INPUT grid (zero-stuffed for stride 2)"""
            z, y, bn, s = u["z"], u["y"], u["bn"], u.get("s", 1)
            C = z.C
            if u["tag"] == "stem" or s == 1:
                dz = self._act((u["tag"], "dz"), N, z.H, z.W, C, 1, dev)
                strides = (dz.img_stride, dz.row_stride, dz.px_stride, dz.interior_off())
            else:
                xin = u["x"]
                dz = self._act((u["tag"], "dz"), N, xin.H, xin.W, C, 1, dev)
                strides = (dz.img_stride, s * dz.row_stride, s * dz.px_stride, dz.interior_off())
            dg, db = torch.empty_like(bn.weight, dtype=torch.float32), torch.empty_like(bn.bias, dtype=torch.float32)
            from_z = u["relu"] and not u["res"]        # conv -> BN -> ReLU: the mask is recomputed from z, y is not read
            check(L_.yolo_batchnorm_bwd(dy.p, dy.halo, y.p if (u["relu"] and not from_z) else None, y.halo, z.p, z.halo, N, z.H, z.W, C,
                                        ptr(bn.weight.detach()), ptr(u["stats"]), dz.p, strides[0], strides[1], strides[2], strides[3],
                                        1 if store_masked else 0, 1 if from_z else 0, ptr(dg), ptr(db), ptr(acc), ptr(self._coef), st),
                  f"batchnorm_bwd {u['tag']}")
            grads[bn.weight], grads[bn.bias] = dg, db
            return dz

        def wgrad(u, dz: Act):
            conv, xin, k, s, p = u["conv"], u["x"], u["k"], u["s"], u["p"]
            Hout, Wout = u["z"].H, u["z"].W
            o = offs[id(conv)]
            dwp = scratch[o: o + conv.weight.numel()]
            if Hout >= 2 and Wout >= 2 and (s > 1 or dz.Hp * dz.Wp >= 1.12 * Hout * Wout):
                wd = WgradDesc(N * Hout * Wout, dz.px_stride, xin.px_stride, conv.out_channels, conv.in_channels, k, k, p, xin.row_stride, 0, 0, 0,
                               Wout, Hout, dz.Hp * dz.Wp, dz.Wp * s, s, dz.halo * dz.Wp + dz.halo)
            else:
                wd = WgradDesc(dz.slots, dz.px_stride, xin.px_stride, conv.out_channels, conv.in_channels, k, k, p, xin.row_stride, 0, 0)
            dw = torch.empty_like(conv.weight, dtype=torch.float32)
            with _on_side_stream(main_t, side_t) as wst:
                with _timed(f"{u['tag']}.wgrad", "wgrad", 2.0 * N * Hout * Wout * conv.out_channels * conv.in_channels * k * k):
                    check(L_.yolo_wgrad(ctypes.byref(wd), xin.p, dz.p, ptr(dwp), None, wst), f"wgrad {u['tag']}")
            grads[conv.weight] = dw
            pending.append((conv, dwp, dw))

        def dgrad(u, dz: Act, add: Act | None) -> Act:
            conv, xin, k, p = u["conv"], u["x"], u["k"], u["p"]
            g = self._act((u["tag"], "gx"), N, xin.H, xin.W, conv.in_channels, 1, dev)
            d = IgemmDesc()
            d.N, d.Ho, d.Wo = N, xin.H, xin.W
            d.in_img_stride, d.in_row_stride, d.in_px_stride = dz.img_stride, dz.row_stride, dz.px_stride
            d.in_off = dz.interior_off(k - 1 - p)
            d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, k, k, conv.out_channels, conv.in_channels
            d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = g.img_stride, g.row_stride, g.px_stride, g.interior_off()
            d.slope, d.out_fp32, d.split_k = 1.0, 0, 1
            aux, bias = None, None
            d.epilogue = EPI_NONE
            if add is not None:
                d.epilogue = _hip.EPI_BIAS_ADD_LRELU          # slope 1: out = conv + 0 + aux
                d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = add.img_stride, add.row_stride, add.px_stride, add.interior_off()
                aux, bias = add.p, ptr(self._zero_bias)
            with _timed(f"{u['tag']}.dgrad", "igemm", 2.0 * N * xin.H * xin.W * conv.out_channels * conv.in_channels * k * k):
                igemm_call(d, dz.p, ptr(u["wd"]), bias, aux, g.p, st, f"dgrad {u['tag']}")
            return g

        out = saved["out"]
        gout = gout.detach()
        if gout.dtype != torch.float32 or not gout.is_contiguous():
            gout = gout.float().contiguous()
        cur_g = self._act(("g", "out"), N, out.H, out.W, out.C, 1, dev)
        check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(gout), N, out.C, out.H, out.W, cur_g.p, out.C, 1, 1, st), "gout nchw->nhwc")
        for (li, bi, u1, u2, u3, ud) in reversed(saved["blocks"]):
            if self.trace is not None:
                self.trace.append(((li, bi), "gout", cur_g.interior().float().permute(0, 3, 1, 2).contiguous()))
            dz3 = bn_bwd(u3, cur_g, True)                 # cur_g becomes g * [out > 0]: what the identity branch receives
            wgrad(u3, dz3)
            g_t2 = dgrad(u3, dz3, None)
            dz2 = bn_bwd(u2, g_t2, False)
            wgrad(u2, dz2)
            g_t1 = dgrad(u2, dz2, None)
            dz1 = bn_bwd(u1, g_t1, False)
            wgrad(u1, dz1)
            if ud is not None:
                dzd = bn_bwd(ud, cur_g, False)
                wgrad(ud, dzd)
                g_idn = dgrad(ud, dzd, None)
            else:
                g_idn = cur_g
            cur_g = dgrad(u1, dz1, g_idn)
            if self.trace is not None:
                self.trace.append(((li, bi), "gx", cur_g.interior().float().permute(0, 3, 1, 2).contiguous()))
            if len(pending) >= 24:
                flush()
        flush()
        # stem: MaxPool2d(3,2,1) backward -> BatchNorm/ReLU backward -> the direct 7x7 weight-gradient kernel
        stem = saved["stem"]
        y0, xin = stem["y"], stem["x"]
        g_y0 = self._act(("stem", "gy"), N, y0.H, y0.W, 64, 1, dev)
        pd = PoolDesc(N, y0.H, y0.W, 64, y0.halo, cur_g.halo)
        check(L_.yolo_maxpool3s2_bwd(ctypes.byref(pd), y0.p, cur_g.p, g_y0.p, g_y0.halo, st), "maxpool3s2_bwd")
        dz0 = bn_bwd(stem, g_y0, False)
        conv = stem["conv"]
        dw = torch.empty_like(conv.weight, dtype=torch.float32)
        part = getattr(self, "_stem_part", None)
        if part is None or part.device != dev:
            part = self._stem_part = torch.empty((768 * 14400,), dtype=torch.float32, device=dev)
            self._stem_db = torch.empty(64, dtype=torch.float32, device=dev)
        check(L_.yolo_wgrad_stem7(xin.p, dz0.p, N, y0.H, y0.W, xin.img_stride, xin.row_stride, dz0.img_stride, dz0.row_stride, dz0.interior_off(),
                                  ptr(dw), ptr(self._stem_db), ptr(part), part.numel(), st), "wgrad_stem7")
        grads[conv.weight] = dw
        if side_t is not None:
            main_t.wait_stream(side_t)          # every weight gradient is final before the pass returns
        return grads

    def _act(self, key, N, H, W, C, halo, dev):
        k = (key, N, H, W, C, halo, str(dev))
        a = self._bufs.get(k)
        if a is None:
            a = Act(N, H, W, C, halo, dev)
            self._bufs[k] = a
        return a

    def _conv(self, tag, a_in: Act, packed, N, relu: bool, residual: Act | None, dev, st):
        wf, b, conv = packed
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (a_in.H + 2 * p - k) // s + 1, (a_in.W + 2 * p - k) // s + 1
        a_out = self._act(tag, N, Ho, Wo, conv.out_channels, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.in_off = a_in.interior_off(p)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = s, k, k, conv.in_channels, conv.out_channels
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = a_out.img_stride, a_out.row_stride, a_out.px_stride, a_out.interior_off()
        d.slope = 0.0 if relu else 1.0
        aux = None
        if residual is not None:
            d.epilogue = _hip.EPI_BIAS_ADD_LRELU
            d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = residual.img_stride, residual.row_stride, residual.px_stride, residual.interior_off()
            aux = residual.p
        else:
            d.epilogue = EPI_BIAS_LRELU if relu else EPI_BIAS
        with _timed(str(tag), "igemm", 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * k * k):
            igemm_call(d, a_in.p, ptr(wf), ptr(b), aux, a_out.p, st, f"igemm {tag}")
        return a_out

    @_hip.device_guard
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N,3,H,W) fp32 on the device -> (N,2048,H/32,W/32) fp32."""
        _hip.require_cuda(x)
        st = stream()
        pk = self._pack_all()
        N, _, H, W = x.shape
        dev = x.device
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        a = self._act("in", N, H, W, 4, 3, dev)
        check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, 3, H, W, a.p, 4, 3, 3, st), "nchw->nhwc4")
        # stem: 7x7/s2 (+BN+ReLU) as the row-segment implicit GEMM, then MaxPool2d(3,2,1)
        wf, b, conv = pk["stem"]
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        s1 = self._act("stem", N, Ho, Wo, 64, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a.img_stride, a.row_stride, a.px_stride, 0
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 2, 7, 1, 32, 64
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = s1.img_stride, s1.row_stride, s1.px_stride, s1.interior_off()
        d.epilogue, d.slope = EPI_BIAS_LRELU, 0.0
        if STEM_KERNEL and Ho % 8 == 0 and Wo % 16 == 0:
            with _timed("stem", "stem", 2.0 * N * Ho * Wo * 64 * 147):
                check(lib().yolo_conv_stem7_fwd(a.p, ptr(wf), ptr(b), N, Ho, Wo, a.img_stride, a.row_stride, 0.0, 0, s1.p, s1.img_stride, s1.row_stride,
                                                s1.interior_off(), None, 0, 0, 0, st), "conv_stem7_fwd")
        else:
            with _timed("stem", "igemm", 2.0 * N * Ho * Wo * 64 * 147):
                _igemm(lib(), d, a.p, ptr(wf), ptr(b), None, s1.p, st, "igemm stem")
        Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        cur = self._act("pool", N, Hq, Wq, 64, 1, dev)
        pd = PoolDesc(N, Ho, Wo, 64, 1, 1)
        check(lib().yolo_maxpool3s2_fwd(ctypes.byref(pd), s1.p, cur.p, st), "maxpool3s2")
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                idn = cur if blk.downsample is None else self._conv((li, bi, "d"), cur, pk[(li, bi, "d")], N, False, None, dev, st)
                t = self._conv((li, bi, 1), cur, pk[(li, bi, 1)], N, True, None, dev, st)
                t = self._conv((li, bi, 2), t, pk[(li, bi, 2)], N, True, None, dev, st)
                cur = self._conv((li, bi, 3), t, pk[(li, bi, 3)], N, True, idn, dev, st)
        out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
        check(lib().yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        return out
