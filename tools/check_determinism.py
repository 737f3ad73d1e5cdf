#!/usr/bin/env python3
"""run every yolo_igemm tile configuration several times on the same operands: outputs must be bit-identical
(no atomics involved) -- a differing run means an LDS / pipeline race."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from yolo._hip import lib, check, ptr, stream, IgemmDesc, EPI_BIAS_LRELU
from yolo.engine import Act
dev = torch.device("cuda")
N = 64
for (co, ci, k, hw) in ((1024, 512, 3, 28), (512, 256, 3, 56), (1024, 1024, 3, 14), (256, 512, 1, 28)):
    x = Act(N, hw, hw, ci, 1, dev); x.t.normal_()
    w = torch.randn((co, k, k, ci), device=dev).to(torch.bfloat16); b = torch.randn((co,), device=dev)
    d = IgemmDesc(); d.N, d.Ho, d.Wo = N, hw, hw
    d.in_img_stride, d.in_row_stride, d.in_px_stride = x.img_stride, x.row_stride, x.px_stride
    d.in_off = x.interior_off(k // 2); d.stride = 1; d.KH = d.KW = k; d.tap_len = ci; d.Cout = co
    ref = None
    for hint in (1, 5, 2, 6, 11, 12, 13, 3, 4):
        y = Act(N, hw, hw, co, 1, dev)
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = y.img_stride, y.row_stride, y.px_stride, y.interior_off()
        d.epilogue, d.slope, d.out_fp32, d.split_k, d.tile_hint = EPI_BIAS_LRELU, 0.1, 0, 1, hint
        outs = []
        for rep in range(6):
            y.t.zero_()
            check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), None, y.p, stream()))
            torch.cuda.synchronize()
            outs.append(y.t.clone())
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        nd = max(int((outs[0] != o).sum()) for o in outs[1:])
        if ref is None: ref = outs[0].float()
        rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
        print(f"co {co} ci {ci} k {k} hw {hw} hint {hint:2d}: deterministic={same} max differing elems={nd} rel vs hint1={rel:.2e}")
