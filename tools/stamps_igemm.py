#!/usr/bin/env python3
"""Where one K iteration of the staggered yolo_igemm kernels spends its cycles (diagnostic build with s_memtime stamps).

    make -C yolo-v1_amd/csrc diag && YOLO_HIP_LIB=yolo-v1_amd/yolo/libyolo_hip_diag.so python tools/stamps_igemm.py [LAYER] [HINT[:TILE_PX]]

Group A (waves 0-3) segments: reads issue | DMA issue | lgkm wait | barrier | MFMAs | vmcnt wait | barrier
Group B (waves 4-7) segments: reads issue | DMA issue | lgkm + vmcnt wait | barrier | - | MFMAs | barrier"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo._hip import lib, check, ptr, stream, IgemmDesc, EPI_BIAS_LRELU
from yolo.engine import Act

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 44
hint, _, tpx = (sys.argv[2] if len(sys.argv) > 2 else "14:196").partition(":")
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
    x = Act(N, hin, hin, ci, 1, dev); y = Act(N, h, h, co, 1, dev)
    x.t.normal_()
    w = torch.randn((co, k, k, ci), device=dev).to(torch.bfloat16)
    b = torch.randn((co,), device=dev)
    d = IgemmDesc()
    d.N, d.Ho, d.Wo = N, h, h
    d.in_img_stride, d.in_row_stride, d.in_px_stride = x.img_stride, x.row_stride, x.px_stride
    d.stride = s; d.Cout = co; d.in_off = x.interior_off(p); d.KH = d.KW = k; d.tap_len = ci
    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = y.img_stride, y.row_stride, y.px_stride, y.interior_off()
    d.epilogue, d.slope, d.out_fp32, d.split_k = EPI_BIAS_LRELU, 0.1, 0, 1
    d.tile_hint, d.tile_px = int(hint), int(tpx or 0)
    nk = k * k * ci // (64 if int(hint) == 11 else 32)
    buf = torch.zeros(512 * 64, dtype=torch.int64, device=dev)
    for _ in range(20):
        check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), None, y.p, stream()))
    if int(hint) == 15:
        # the pipelined kernel stamps its sections, not a K iteration
        check(lib().yolo_debug_stamps(ptr(buf), 0))
        for _ in range(3):
            check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), None, y.p, stream()))
        torch.cuda.synchronize()
        st = buf.cpu().view(512, 8, 8).double()
        ok = st[:, 0, 4] > 0
        seg = (st[:, :, 1:5] - st[:, :, 0:4])[ok].reshape(-1, 4)
        t0 = st[ok][:, :, 0].min()
        print(f"layer {layer} hint 15:{tpx or 0}: {int(ok.sum())} workgroups stamped; medians in cycles")
        print("  table %6.0f | first stage %6.0f | K loop (%d steps) %7.0f = %5.0f / step | epilogue %6.0f | total %7.0f" % (
            seg[:, 0].median(), seg[:, 1].median(), nk, seg[:, 2].median(), seg[:, 2].median() / nk, seg[:, 3].median(), seg.sum(1).median()))
        print("  first workgroup starts at 0, the last stamped one at %.0f, the last one ends at %.0f cycles" % ((st[ok][:, :, 0].max() - t0).item(), (st[ok][:, :, 4].max() - t0).item()))
        check(lib().yolo_debug_stamps(None, 0))
        sys.exit(0)
    for it in (nk // 2, nk // 2 + 1, nk // 3):
        check(lib().yolo_debug_stamps(ptr(buf), it))
        for _ in range(3):
            check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b), None, y.p, stream()))
        torch.cuda.synchronize()
        st = buf.cpu().view(512, 8, 8).double()
        seg = st[:, :, 1:] - st[:, :, :-1]            # [wg][wave][7]
        tot = st[:, :, 7] - st[:, :, 0]
        ok = tot[:, 0] > 0
        a, bb = seg[ok][:, :4].reshape(-1, 7), seg[ok][:, 4:].reshape(-1, 7)
        print(f"layer {layer} hint {hint}:{tpx or 0} k-iter {it}/{nk}: {int(ok.sum())} workgroups stamped")
        print("  group A  reads %5.0f | dma %5.0f | lgkm %5.0f | barrier %5.0f | mfma %5.0f | vmcnt %5.0f | barrier %5.0f  = %6.0f cycles" % (*a.median(0).values.tolist(), tot[ok][:, :4].median().item()))
        print("  group B  reads %5.0f | dma %5.0f | waits %5.0f | barrier %5.0f | - %5.0f | mfma %5.0f | barrier %5.0f  = %6.0f cycles" % (*bb.median(0).values.tolist(), tot[ok][:, 4:].median().item()))
        if it == nk // 3:
            for wg in (0, 100):
                m = st[wg]
                print(f"  workgroup {wg}: stamps of wave w relative to the earliest stamp")
                for wv in range(8):
                    print("    w%d " % wv + " ".join("%6.0f" % (v - m.min().item()) for v in m[wv].tolist()))
    check(lib().yolo_debug_stamps(None, 0))
