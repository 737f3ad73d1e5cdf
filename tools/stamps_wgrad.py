#!/usr/bin/env python3
"""Where one 32-pixel stage of wgrad_pipe_kernel / wgrad_wide_kernel (yolo_wgrad variant 5 / 6, VARIANT=) spends its cycles -- diagnostic build with s_memtime stamps.

    make -C yolo-v1_amd/csrc diag && YOLO_HIP_LIB=yolo-v1_amd/yolo/libyolo_hip_diag.so python tools/stamps_wgrad.py [LAYER] [STAGE]

segments: barrier wait | sub-step 0: six MFMAs + 12 transposing reads + address arithmetic | DMA pair 0 | MFMA 7 + DMA pair 1 |
          MFMA 8 | sub-step 1: eight MFMAs + 12 reads"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo._hip import lib, check, ptr, stream, WgradDesc
from yolo.engine import Act

VARIANT = int(os.environ.get("VARIANT", "5"))
NWV = 4 if VARIANT == 6 else 8            # waves per workgroup
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 12
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N, dev, h = 64, torch.device("cuda"), 448
for item in synth.YOLOV1_BACKBONE_CFG:
    if item == "M":
        h //= 2
        continue
    idx, (co, ci, k, s, p) = item
    hin = h
    h = (h + 2 * p - k) // s + 1
    if idx != layer:
        continue
    x = Act(N, h, h, ci, 1, dev); dy = Act(N, h, h, co, 1, dev)
    x.t.normal_(); dy.t.normal_()
    dwp = torch.zeros((co, k, k, ci), dtype=torch.float32, device=dev)
    wd = WgradDesc(N * h * h, dy.px_stride, x.px_stride, co, ci, k, k, p, x.row_stride, int(os.environ.get("SPLIT", "0")), 0, VARIANT, h, h, dy.Hp * dy.Wp, dy.Wp, 1, dy.Wp + 1)
    buf = torch.zeros(512 * 64, dtype=torch.int64, device=dev)
    for _ in range(10):
        check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dwp), None, stream()))
    check(lib().yolo_debug_stamps(ptr(buf), stage))
    for _ in range(3):
        check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dwp), None, stream()))
    torch.cuda.synchronize()
    check(lib().yolo_debug_stamps(None, 0))
    st = buf.cpu().view(512, 8, 8).double()[:, :NWV]
    if stage < 0:      # whole-kernel sections: start | first stage landed | K loop done | bias atomics | output stored
        ok = st[:, 0, 4] > 0
        w = st[ok]
        t0 = w[:, :, 0].min()
        sec = w[:, :, 1:5] - w[:, :, 0:4]
        print(f"layer {layer}: {int(ok.sum())} workgroups; stages per workgroup median {w[:, 0, 5].median().item():.0f}; median cycles")
        print("  prologue (tables + first stage) %7.0f | K loop %8.0f (%5.0f per stage) | bias %6.0f | output %7.0f | whole %8.0f | start spread %7.0f end spread %7.0f"
              % (sec[:, :, 0].median().item(), sec[:, :, 1].median().item(), (sec[:, :, 1] / w[:, :, 5:6].clamp(min=1)[:, 0:1, 0]).median().item(),
                 sec[:, :, 2].median().item(), sec[:, :, 3].median().item(), (w[:, :, 4] - w[:, :, 0]).median().item(),
                 (w[:, 0, 0].max() - t0).item(), (w[:, 0, 4].max() - w[:, 0, 4].min()).item()))
        continue
    if stage >= 100000:      # variant 6: s_memtime at the top of eight consecutive stages
        ok = st[:, 0, 7] > 0
        d = (st[:, :, 1:8] - st[:, :, 0:7])[ok]
        print(f"layer {layer} stages {stage - 100000}..: {int(ok.sum())} workgroups; cycles from stage top to stage top, median over workgroups, wave 0: "
              + " ".join("%5.0f" % v for v in d[:, 0].median(0).values.tolist()))
        continue
    ok = st[:, 0, 6] > 0
    seg = (st[:, :, 1:7] - st[:, :, 0:6])[ok].reshape(-1, 6)
    print(f"layer {layer} stage {stage}: {int(ok.sum())} workgroups stamped; median cycles per wave")
    print("  barrier %5.0f | 4 MFMA + 12 reads + addresses %5.0f | DMA + 2 MFMA %5.0f | DMA + 2 MFMA %5.0f | 4 MFMA + 12 reads %5.0f | 2 x (DMA + 2 MFMA) %5.0f | total %6.0f"
          % (*seg.median(0).values.tolist(), (st[:, :, 6] - st[:, :, 0])[ok].median().item()))
    if VARIANT == 6:
        print("  100 stages from this one on: median %6.0f cycles per stage" % ((st[:, :, 6] - st[:, :, 7])[ok].median().item() / 100))
        st[:, :, 6] = st[:, :, 5]
        print("  wait for the stage's operands (vmcnt) in front of the barrier: median %5.0f cycles" % (st[:, :, 0] - st[:, :, 7])[ok].median().item())
    m = st[int(ok.nonzero()[0])]
    print("  first stamped workgroup, stamps of wave w relative to the earliest:")
    for wv in range(NWV):
        print("    w%d " % wv + " ".join("%6.0f" % (v - m[:, :7].min().item()) for v in m[wv, :7].tolist()))
