#!/usr/bin/env python3
"""micro-benchmark: the 3x3 64 -> 64 conv of ResNet-50's first stage (batch 64, 112 x 112) through the tiled kernels and conv_c64.hip (tile_hint 22)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo import engine
from yolo._hip import lib, ptr, stream, IgemmDesc, EPI_BIAS_LRELU
from yolo.engine import Act

N, H, W, C = 64, 112, 112, 64
dev = torch.device("cuda")
a_in = Act(N, H, W, C, 1, dev); a_out = Act(N, H, W, C, 1, dev)
a_in.interior().copy_(torch.randn(N, H, W, C, device=dev))
w = (torch.randn(C, 3, 3, C, device=dev) / 24).to(torch.bfloat16)
b = torch.randn(C, device=dev)
d = IgemmDesc()
d.N, d.Ho, d.Wo = N, H, W
d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a_in.img_stride, a_in.row_stride, a_in.px_stride, a_in.interior_off(1)
d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 3, 3, C, C
d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = a_out.img_stride, a_out.row_stride, a_out.px_stride, a_out.interior_off()
d.epilogue, d.slope, d.out_fp32, d.split_k, d.pool2 = EPI_BIAS_LRELU, 0.0, 0, 1, 0
fl = 2.0 * N * H * W * C * C * 9
for rnd in range(2):
    for plan in [(4, 1), (3, 1), (10, 1), (22, 1)]:
        ts = []
        for rep in range(3):
            for _ in range(3):
                engine._run_plan_igemm(lib(), d, plan, a_in.p, ptr(w), ptr(b), None, a_out.p, stream(), "t")
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                engine._run_plan_igemm(lib(), d, plan, a_in.p, ptr(w), ptr(b), None, a_out.p, stream(), "t")
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        t = min(ts)
        print(f"plan {plan}: {t * 1e3:7.1f} us  {fl / t / 1e9:6.0f} TFLOP/s  {(2 * N * H * W * C * 2) / t / 1e9:5.2f} TB/s (in + out once)")
