#!/usr/bin/env python3
"""Where a workgroup of the persistent yolo_igemm kernels (tile_hint 20 / 21) spends its cycles (diagnostic build with s_memtime stamps).

    make -C yolo-v1_amd/csrc diag && YOLO_HIP_LIB=yolo-v1_amd/yolo/libyolo_hip_diag.so python tools/stamps_persist.py [LAYER] [HINT[:TILE_PX]] [POOL]

Stamps per wave: kernel start | stage 0 visible | end of tile 0's K loop | end of its epilogue | the same two of tile 1 | end of the last
tile's K loop | kernel end."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo._hip import lib, check, ptr, stream, IgemmDesc, EPI_BIAS_LRELU
from yolo.engine import Act

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 17
hint, _, tpx = (sys.argv[2] if len(sys.argv) > 2 else "20:196").partition(":")
pool = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N = 64
dev = torch.device("cuda")
h = 448
for item in synth.YOLOV1_BACKBONE_CFG:
    if item == "M":
        h //= 2
        continue
    idx, (co, ci, k, s, p) = item
    hin = h
    h = (h + 2 * p - k) // s + 1
    if idx != layer:
        continue
    x = Act(N, hin, hin, ci, 1, dev)
    y = Act(N, h // 2, h // 2, co, 1, dev) if pool else Act(N, h, h, co, 1, dev)
    x.t.normal_()
    w = torch.randn((co, k, k, ci), device=dev).to(torch.bfloat16)
    b = torch.randn((co,), device=dev)
    d = IgemmDesc()
    d.N, d.Ho, d.Wo = N, h, h
    d.in_img_stride, d.in_row_stride, d.in_px_stride = x.img_stride, x.row_stride, x.px_stride
    d.stride = s; d.Cout = co; d.in_off = x.interior_off(p); d.KH = d.KW = k; d.tap_len = ci; d.pool2 = pool
    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = y.img_stride, y.row_stride, y.px_stride, y.interior_off()
    d.epilogue, d.slope, d.out_fp32, d.split_k = EPI_BIAS_LRELU, 0.1, 0, 1
    d.tile_hint, d.tile_px = int(hint), int(tpx or 0)
    nk = k * k * ci // 32
    tp = int(tpx or 0) or (224 if int(hint) == 21 else 208)
    tiles = ((co + 255) // 256) * ((N * h * h + tp - 1) // tp)
    buf = torch.zeros(512 * 64, dtype=torch.int64, device=dev)
    for _ in range(20):
        check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), None, y.p, stream()))
    check(lib().yolo_debug_stamps(ptr(buf), 0))
    for _ in range(3):
        check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), None, y.p, stream()))
    torch.cuda.synchronize()
    check(lib().yolo_debug_stamps(None, 0))
    st = buf.cpu().view(512, 8, 8).double()
    ok = st[:, 0, 7] > 0
    s_ = st[ok]                                     # [wg][wave][8]
    t0 = s_[:, :, 0].min()
    med = lambda v: v.reshape(-1).median().item()
    print(f"layer {layer} hint {hint}:{tpx or 0} pool {pool}: {tiles} tiles, {int(ok.sum())} workgroups stamped, {nk} K steps per tile; medians in cycles")
    print("  prologue (table + first stage)  %7.0f" % med(s_[:, :, 1] - s_[:, :, 0]))
    print("  tile 0: K loop %7.0f = %5.0f / step | epilogue %6.0f" % (med(s_[:, :, 2] - s_[:, :, 1]), med(s_[:, :, 2] - s_[:, :, 1]) / nk, med(s_[:, :, 3] - s_[:, :, 2])))
    two = s_[:, 0, 4] > 0
    if two.any():
        t = s_[two]
        print("  tile 1: K loop %7.0f = %5.0f / step | epilogue %6.0f   (%d workgroups)" % (med(t[:, :, 4] - t[:, :, 3]), med(t[:, :, 4] - t[:, :, 3]) / nk, med(t[:, :, 5] - t[:, :, 4]), int(two.sum())))
    print("  last K loop end -> kernel end   %7.0f | whole workgroup %8.0f" % (med(s_[:, :, 7] - s_[:, :, 6]), med(s_[:, :, 7] - s_[:, :, 0])))
    print("  first workgroup starts at 0, the last one at %.0f, the last one ends at %.0f cycles" % ((s_[:, :, 0].max() - t0).item(), (s_[:, :, 7].max() - t0).item()))
    for wg in (0, 100):
        if wg < s_.shape[0]:
            m = s_[wg]
            print(f"  workgroup {wg}: stamps of wave w relative to its first")
            for wv in range(8):
                print("    w%d " % wv + " ".join("%7.0f" % (v - m[:, 0].min().item()) for v in m[wv].tolist()))
