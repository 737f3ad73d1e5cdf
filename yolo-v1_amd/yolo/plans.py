"""Per-problem launch plans of yolo_igemm: the shipped table (``plans/gfx950.json``), the deterministic default for problems without an entry, the
tuner that measures the table (tools/tune_plans.py) and the code that runs a problem with its plan."""

from __future__ import annotations

import ctypes
import os

import torch

from . import _hip
from ._hip import EPI_BIAS, EPI_BIAS_ADD_LRELU, EPI_BIAS_LRELU, EPI_MUL_DLRELU, EPI_NONE, IgemmDesc, check, ptr
from .config import CONFIG as CFG
from .runtime import RT, _igemm

# ---- per-problem launch plans ---------------------------------------------------------------------------
# The best yolo_igemm configuration depends on the layer shape (tile quantisation over 256 CUs, K depth).  Plans are DATA:
# ``yolo/plans/gfx950.json`` ships the plans of every problem of the BASELINE configurations (measured once on MI355X by
# tools/tune_plans.py) and is loaded at import, so every process and every rank runs the same launches -- outputs are
# bit-identical across processes and nothing is timed, flushed or synchronised at run time.  A problem without an entry
# takes ``_default_plan`` (a deterministic function of the shape).  CFG.AUTOTUNE = True (tools/tune_plans.py only) times the
# candidates on first use and records the winner in _TUNED.
#
# plan forms (tuples; JSON lists):
#   (hint, order)                          one launch of tile configuration `hint`
#   (hint, order, px_cut, tail_hint)       pixels [0, px_cut) with `hint` (whole rounds of the chip), the rest with `tail_hint`
#   ("skew", hint, order, phases, step)    one launch, first-round workgroups start phase * step cycles apart
#   ("tile", hint, order, tile_px[, phases, step])   one launch whose tiles cover tile_px pixels (yolo_igemm_desc.tile_px)
#   ("splitk", hint, S)                    S <= 2 K-splits with fp32 atomics into a zeroed scratch + yolo_igemm_finish
#   ("slabs", hint, S, tile_px)            S K-splits stored as slabs + fixed-order reduce in yolo_igemm_finish (deterministic)
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
    if data.get("wgrad"):           # weight-gradient kernel choices measured inside the training step (Plan._wgrad_desc)
        CFG.WGRAD_CHOICE = dict(CFG.WGRAD_CHOICE or {})
        CFG.WGRAD_CHOICE.update({tuple(int(t) for t in k.split(",")): tuple(int(t) for t in v) for k, v in data["wgrad"].items()})
    return len(data.get("plans", {}))


def save_plans(path: str = PLAN_FILE, note: str = "") -> None:
    import json
    os.makedirs(os.path.dirname(path), exist_ok=True)
    body = {"arch": "gfx950", "key": "N,Ho,Wo,KH,KW,tap_len,Cout,stride,epilogue,pool2,out_px_stride,in_px_stride", "note": note,
            "plans": {_key_str(k): list(v) for k, v in sorted(_TUNED.items())}}
    if CFG.WGRAD_CHOICE:
        body["wgrad"] = {_key_str(k): list(v) for k, v in sorted(CFG.WGRAD_CHOICE.items())}
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
         20: (256, 208), 21: (256, 224),                                                     # persistent kernels (igemm_persist.hip)
         22: (64, 256)}                                                                      # 3x3 64 -> 64, weights resident in LDS, 16 x 16-pixel tiles (conv_c64.hip)
_TAIL_CANDIDATES = (5, 3, 4)
# (workgroup slots of the chip, relative time of one tile) per configuration, for _default_plan: 8-wave configurations run
# one workgroup per CU, the 4-wave ones two; times are relative to a 256x256 tile and follow the measured in-tile rates
_TILE_COST = {12: (256, 1.00), 14: (256, 0.80), 11: (256, 0.54), 5: (512, 0.36), 3: (512, 0.20), 4: (512, 0.20)}

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
    if M < 2048 or (M < 8192 and ((d.Cout + 63) // 64) * ((M + 127) // 128) < 256):
        # a handful of pixel tiles (small batches on the 14x14 / 7x7 maps) under a deep K: without a split a few workgroups walk hundreds of K steps while
        # the chip idles -- K ranges of 64 x 128 tiles as slabs, enough of them for two workgroups per CU (batch 1, measured: 90 -> 17-21 us per layer)
        ktot = d.KH * d.KW * d.tap_len
        if (CFG.SMALL_SPLIT and ktot >= 2304 and d.tap_len % 64 == 0 and d.Cout % 8 == 0 and not d.out_fp32 and not d.bn_stats and d.split_k <= 1
                and d.epilogue in (EPI_NONE, EPI_BIAS, EPI_BIAS_LRELU, EPI_MUL_DLRELU)):
            hint = 3 if (M + 63) // 64 * 64 < (M + 127) // 128 * 128 else 4       # 128 co x 64 px where that wastes fewer pixel rows, else 64 co x 128 px
            tco, tpx = _TILE[hint]
            tiles = ((d.Cout + tco - 1) // tco) * ((M + tpx - 1) // tpx)
            S = 2
            while S < 32 and tiles * S < 512 and (ktot // 64) // (2 * S) >= 8:      # two 4-wave workgroups per CU; >= 8 K steps of 64 per range
                S *= 2
            return ("slabs", hint, S, 0)
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


_BORROWED = set()       # keys whose plan in _TUNED was taken over from another batch size and has not run yet (igemm_call falls back to the default if the library refuses it)


def _borrowed_plan(key):
    """the plan measured for the same layer (every key field but N equal) at the nearest measured batch size, nearest by ratio, the larger one on a tie.
    Launch plans name a kernel configuration, a tile size, a K-range count -- nothing that depends on the number of pixels -- except the pixel-range
    forms (hint, order, cut, hint2), whose cut is a pixel count: those are not taken over.  tools/batch_sweep.py, YOLOv1 / ResNet-50 variant forward: batch 13
    1.154 -> 1.027 / 2.096 -> 1.892 ms, 24 1.482 -> 1.456 / 3.052 -> 2.673, 48 2.452 -> 2.286 / 5.119 -> 4.427."""
    n = key[0]
    if not (CFG.BORROW_PLANS and CFG.PLAN_TABLE) or n < 8:      # (below 8 images the default rule, which sizes its K ranges by the actual pixel count, measured 1-3 % faster)
        return None
    sizes = sorted({k[0] for k in _TUNED if k[1:] == key[1:] and k not in _BORROWED and k[0] != n})
    if not sizes:
        return None
    nb = min(sizes, key=lambda b: (max(b, n) / min(b, n), -b))
    plan = _TUNED[(nb,) + tuple(key[1:])]
    if isinstance(plan[0], int) and len(plan) == 4:
        return None
    return plan


def _run_plan_igemm(L_, d: IgemmDesc, plan, inp, w, bias, aux, out, st, what):
    """run one yolo_igemm problem with a launch plan (forms: see above)"""
    if d.bn_stats and (plan[0] in ("splitk", "slabs") or (isinstance(plan[0], int) and len(plan) == 4)):
        # a launch that also accumulates BatchNorm statistics must be one plain launch: keep the plan's main configuration
        # (a slab plan of the pipelined / persistent kernels, measured on the launch without statistics: their one-launch stand-in is the staggered loop)
        plan = ((14 if plan[1] in (15, 16, 17, 18, 20, 21) else plan[1]), 1) if plan[0] in ("splitk", "slabs") else (plan[0], plan[1])
    if plan[0] == "tile" and plan[1] in (20, 21) and (d.pool2 == 2 or not CFG.PERSIST or not _persist_ok(d)):
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
        acc = RT._splitk_scratch(M * d.Cout * (S if slabs else 1), zero=not slabs)
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


class _RawDevice:
    """n 16-bit (or 32-bit) words of device memory at ptr as a __cuda_array_interface__ object (torch.as_tensor takes it without a copy)"""

    def __init__(self, ptr_, n, fp32):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4" if fp32 else "<i2", "data": (int(ptr_), False), "version": 2}


def _output_snapshot(L_, d: IgemmDesc, plan, inp, w, bias, aux, out, st, what):
    """run one plan and return a copy of the problem's whole output buffer as fp32 (the tuner's numerical check)"""
    _run_plan_igemm(L_, d, plan, inp, w, bias, aux, out, st, what)
    torch.cuda.synchronize()
    addr = out.value if hasattr(out, "value") else int(out)
    t = torch.as_tensor(_RawDevice(addr, d.N * d.out_img_stride, bool(d.out_fp32)), device="cuda")
    t = t if d.out_fp32 else t.view(torch.bfloat16)
    return torch.nan_to_num(t.float(), nan=0.0, posinf=0.0, neginf=0.0).clone()


def _tune(L_, d: IgemmDesc, inp, w, bias, aux, out, st, what):
    """time the candidate plans of one problem (events on the launch stream behind a cache flush, min of 3) and return the
    fastest.  Only reached with CFG.AUTOTUNE = True (tools/tune_plans.py)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    stats_ptr, d.bn_stats = d.bn_stats, None        # tuning repeats the launch: keep it idempotent
    M = d.N * d.Ho * d.Wo

    def timed(plan):
        try:
            _run_plan_igemm(L_, d, plan, inp, w, bias, aux, out, st, what)
        except _hip.HipUnsupported:
            return None          # this configuration does not take the shape; every other error is a real failure
        ts = []
        reps = CFG.TUNE_REPS
        for _ in range(3):
            if reps <= 1:
                _flush_caches(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(max(1, reps)):
                _run_plan_igemm(L_, d, plan, inp, w, bias, aux, out, st, what)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / max(1, reps))
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
    if (d.KH == 3 and d.KW == 3 and d.tap_len == 64 and d.Cout == 64 and d.stride == 1 and d.Ho % 16 == 0 and d.Wo % 16 == 0 and not d.pool2 and not d.out_fp32
            and d.epilogue in (EPI_NONE, EPI_BIAS, EPI_BIAS_LRELU)):
        cands.append(22)      # ResNet-50's first-stage 3x3 convs: weight panel resident in LDS, input patch staged once per 16 x 16 tile (conv_c64.hip)
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
        if M < 2048 or ((d.Cout + 63) // 64) * ((M + 127) // 128) < 256:
            # a handful of pixel tiles (batch 1: 196 or 49 pixels): many K ranges of small tiles so that the weight stream is spread over the chip
            nks = d.KH * d.KW * d.tap_len // 64
            for c in (5, 3, 4):
                for S in (4, 8, 16, 32):
                    if nks // S >= 2:
                        consider(("slabs", c, S, 0))
        if d.Cout % 256 == 0 and d.tap_len % 32 == 0:
            for S in (2, 4, 8):       # the persistent kernel, K ranges as extra tiles (each range: an even number >= 6 of K steps)
                nks = d.KH * d.KW * d.tap_len // 32
                if nks % S == 0 and (nks // S) % 2 == 0 and nks // S >= 6:
                    consider(("slabs", 20, S, tile_px))
    if not times:
        d.bn_stats = stats_ptr
        return (0, 0)
    # the tuner times; before a plan enters the table it must also have COMPUTED the layer: the output of the fastest candidates against the default
    # plan's on the live operands (relative L2 over the whole output buffer, halo included; other fp32 summation orders: ~1e-3), fastest first
    best = None
    want = _output_snapshot(L_, d, _default_plan(d), inp, w, bias, aux, out, st, what)
    for cand in sorted(times, key=times.get):
        got = _output_snapshot(L_, d, cand, inp, w, bias, aux, out, st, what)
        err = float((got - want).norm() / (want.norm() + 1e-30))
        if err < 0.02:
            best = cand
            break
        print(f"tune: {what} {_tune_key(d)}: plan {cand} differs from the default plan's output by {err:.3g} -- dropped", flush=True)
        del times[cand]
    d.bn_stats = stats_ptr
    if best is None:
        return _default_plan(d)
    if CFG.TUNE_LOG is not None:
        CFG.TUNE_LOG.append((_tune_key(d), best, sorted(times.items(), key=lambda kv: kv[1])[:6]))
    return best




def igemm_call(d: IgemmDesc, inp, w, bias, aux, out, st, what: str):
    """yolo_igemm with the problem's launch plan (shipped table, else the deterministic default) -- only for plain,
    idempotent launches."""
    L_ = RT.lib()
    plain = CFG.TILE_HINT == 0 and d.split_k <= 1 and not d.w_blocked and d.tap_len % 64 == 0
    if not plain:
        d.tile_hint, d.tile_px = CFG.TILE_HINT, CFG.TILE_PX
        _igemm(L_, d, inp, w, bias, aux, out, st, what)
        return
    key = _tune_key(d)
    best = _TUNED.get(key)
    if best is None:
        # (few-pixel problems are tuned too when their weight panel is large: at batch 1 the 14x14 / 7x7 layers are a handful of workgroups walking
        # 288 K steps each unless the K range is split over the chip)
        if CFG.AUTOTUNE and CFG.TIMERS is None and (d.N * d.Ho * d.Wo >= 2048 or d.KH * d.KW * d.tap_len * d.Cout >= (1 << 20)):
            best = _tune(L_, d, inp, w, bias, aux, out, st, what)
        else:
            best = _borrowed_plan(key)
            if best is not None:
                _BORROWED.add(key)
            else:
                best = _default_plan(d)
        _TUNED[key] = best
    if key in _BORROWED:
        # first run of a plan measured at another batch size: if the library refuses it for this pixel count (it checks before it launches), the default rule
        _BORROWED.discard(key)
        try:
            _run_plan_igemm(L_, d, best, inp, w, bias, aux, out, st, what)
            return
        except _hip.HipUnsupported:
            best = _TUNED[key] = _default_plan(d)
    rec = CFG.PLAN_TIMES
    if rec is None:
        _run_plan_igemm(L_, d, best, inp, w, bias, aux, out, st, what)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _run_plan_igemm(L_, d, best, inp, w, bias, aux, out, st, what)
    e1.record()
    rec.setdefault(key, []).append((e0, e1))


def _tune_key(d: IgemmDesc):
    # pool2 = 3 (pooled map + arg-max codes, training) runs the launch plan measured for pool2 = 1 (pooled map only, inference)
    return (d.N, d.Ho, d.Wo, d.KH, d.KW, d.tap_len, d.Cout, d.stride, d.epilogue, 1 if d.pool2 == 3 else d.pool2, d.out_px_stride, d.in_px_stride)


if CFG.PLAN_TABLE:
    load_plans()
