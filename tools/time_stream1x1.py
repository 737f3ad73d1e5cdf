#!/usr/bin/env python3
"""micro-benchmark: the thin-K pointwise layers of ResNet-50 at batch 64 / 448x448 through the tiled kernels (tile_hint 10, 5) and the
streaming 1x1 kernel (tile_hint 19, igemm_stream.hip); HBM bytes = activations in + out (+ residual)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import IgemmDesc, check, lib, ptr, stream, EPI_BIAS_LRELU, EPI_BIAS_ADD_LRELU
from yolo.engine import Act

dev = torch.device("cuda")
N = 64
for (cin, cout, hw, res) in [(64, 256, 112, True), (256, 64, 112, False), (64, 64, 112, False), (64, 256, 112, False), (256, 128, 112, False),
                             (128, 512, 56, True), (256, 1024, 28, True), (256, 512, 56, False),
                             # YOLOv1's pointwise layers
                             (192, 128, 56, False), (256, 256, 56, False), (512, 256, 28, False), (512, 512, 28, False)]:
    x = Act(N, hw, hw, cin, 1, dev); x.t.normal_()
    aux = Act(N, hw, hw, cout, 1, dev); aux.t.normal_()
    y = Act(N, hw, hw, cout, 1, dev)
    w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device=dev)
    d = IgemmDesc()
    d.N, d.Ho, d.Wo = N, hw, hw
    d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = x.img_stride, x.row_stride, x.px_stride, x.interior_off()
    d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 1, 1, cin, cout
    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = y.img_stride, y.row_stride, y.px_stride, y.interior_off()
    d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = aux.img_stride, aux.row_stride, aux.px_stride, aux.interior_off()
    d.epilogue, d.slope, d.out_fp32, d.split_k, d.tile_order = (EPI_BIAS_ADD_LRELU if res else EPI_BIAS_LRELU), 0.0, 0, 1, 1
    mb = N * hw * hw * 2 * (cin + cout * (2 if res else 1)) / 1e6
    line = f"{cin:4d} -> {cout:4d} @ {hw:3d}^2 {'+res' if res else '    '} {mb:6.0f} MB |"
    best = {}
    for rnd in range(3):
        for hint in (10, 5, 19):
            d.tile_hint = hint
            for _ in range(3):
                check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), aux.p if res else None, y.p, stream()))
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), aux.p if res else None, y.p, stream()))
            e1.record(); torch.cuda.synchronize()
            best[hint] = min(best.get(hint, 1e9), e0.elapsed_time(e1) / 10)
    for hint in (10, 5, 19):
        line += f" h{hint}: {best[hint] * 1e3:6.1f} us {mb / best[hint] / 1e3:5.2f} TB/s |"
    print(line)
    del x, y, aux
