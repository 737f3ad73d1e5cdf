#!/usr/bin/env python3
"""micro-benchmark: yolo_igemm forward on every YOLOv1 conv layer shape at N=64, per tile configuration."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo._hip import lib, check, ptr, stream, IgemmDesc, EPI_BIAS_LRELU
from yolo.engine import Act

N = int(os.environ.get("N", 64))
# HINTS entries: tile_hint [+ 100 * tile_order] [":" tile_px], e.g. "5,12,14:196"
def _parse(h):
    a, _, b = h.partition(":")
    return (int(a), int(b or 0))
hints = [_parse(h) for h in os.environ.get("HINTS", "0,1,2,3,4").split(",")]
dev = torch.device("cuda")
h = 448
tot = {k: 0.0 for k in hints}
for item in synth.YOLOV1_BACKBONE_CFG:
    if item == "M":
        h //= 2
        continue
    idx, (co, ci, k, s, p) = item
    hin = h
    h = (h + 2 * p - k) // s + 1
    only = os.environ.get("LAYERS")
    if only and str(idx) not in only.split(","):
        continue
    pool = int(os.environ.get("POOL", "0")) and idx in (0, 3, 12, 33)      # the layers in front of a MaxPool2d
    if idx == 0:
        x = Act(N, hin, hin, 4, 3, dev)
    else:
        x = Act(N, hin, hin, ci, 1, dev)
    y = Act(N, h // 2, h // 2, co, 1, dev) if pool else Act(N, h, h, co, 1, dev)
    x.t.normal_()
    w = torch.randn((co, 7, 8, 4) if idx == 0 else (co, k, k, ci), device=dev).to(torch.bfloat16)
    b = torch.randn((co,), device=dev)
    d = IgemmDesc()
    d.N, d.Ho, d.Wo = N, h, h
    d.in_img_stride, d.in_row_stride, d.in_px_stride = x.img_stride, x.row_stride, x.px_stride
    d.stride = s; d.Cout = co; d.pool2 = 1 if pool else 0
    if idx == 0:
        d.in_off = 0; d.KH, d.KW, d.tap_len = 7, 1, 32
    else:
        d.in_off = x.interior_off(p); d.KH = d.KW = k; d.tap_len = ci
    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = y.img_stride, y.row_stride, y.px_stride, y.interior_off()
    d.epilogue, d.slope, d.out_fp32, d.split_k = EPI_BIAS_LRELU, 0.1, 0, 1
    fl = 2.0 * N * h * h * co * ci * k * k
    line = f"idx {idx:2d} co {co:4d} ci {ci:4d} k {k} s {s} out {h:3d} M {N*h*h:7d} K {ci*k*k:5d} |"
    # warm the clocks up on this problem, then time every configuration in three interleaved rounds and keep the
    # minimum: the first configuration measured after an idle gap otherwise reads 10-20 % slow
    def run(hint, reps):
        d.tile_hint = hint[0] % 100
        d.tile_order = hint[0] // 100
        d.tile_px = hint[1]
        for _ in range(reps):
            check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), None, y.p, stream()))
    ok = []
    for hint in hints:
        try:
            run(hint, 3)
            ok.append(hint)
        except RuntimeError:
            pass
    torch.cuda.synchronize()
    for _ in range(3):
        for hint in ok:
            run(hint, 5)
    best = {}
    for _ in range(3):
        for hint in ok:
            run(hint, 1)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            run(hint, 5)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            best[hint] = min(ms, best.get(hint, 1e9))
    for hint in hints:
        if hint in best:
            tot[hint] += best[hint]
            line += f" h{hint[0]}:{hint[1]} {best[hint]:6.3f} ms {fl/best[hint]/1e9:6.0f} TF |"
        else:
            line += f" h{hint[0]}:{hint[1]} n/a |"
    print(line)
    del x, y
print("totals (ms):", {f"{k[0]}:{k[1]}": round(v, 3) for k, v in tot.items()})
