#!/usr/bin/env python3
"""micro-benchmark: the data gradient of the Linear behind nn.Flatten (dxT[k][n] = sum_o W[o][k] gT[o][n], 4096 x 50176 weights, batch 64)
through yolo_wgrad's kernel variants.  VARIANT=0,5,6 python tools/time_fc_dgrad.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import lib, check, ptr, stream, WgradDesc

N, O, K = 64, 4096, 50176
dev = torch.device("cuda")
w = (torch.randn(O, K, device=dev) * 0.01).to(torch.bfloat16)
gT = torch.randn(O, N, device=dev).to(torch.bfloat16)
ref = None
for V in [int(v) for v in os.environ.get("VARIANT", "0,5,6").split(",")]:
    for split in [int(v) for v in os.environ.get("SPLITS", "0,1,2").split(",")]:
        dxT = torch.zeros((K, N), dtype=torch.float32, device=dev)
        wd = WgradDesc(O, K, N, K, N, 1, 1, 0, 0, split, 1, V)
        rc = lib().yolo_wgrad(ctypes.byref(wd), ptr(gT), ptr(w), ptr(dxT), None, stream())
        if rc != 0:
            print(f"variant {V} split {split}: refused"); continue
        torch.cuda.synchronize()
        if ref is None:
            ref = dxT.clone()
        err = ((dxT - ref).norm() / ref.norm()).item()
        ts = []
        for rep in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                check(lib().yolo_wgrad(ctypes.byref(wd), ptr(gT), ptr(w), ptr(dxT), None, stream()))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        t = min(ts)
        print(f"variant {V} split {split}: {t:.3f} ms  {O * K * 2 / t / 1e9:.2f} TB/s of weights  rel diff to first {err:.1e}")
