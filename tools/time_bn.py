#!/usr/bin/env python3
"""micro-benchmark: yolo_batchnorm_bwd (reduce + finalize + apply) and yolo_batchnorm_train_fwd on the ResNet-50 trunk's shapes at batch 64 / 448x448:
ms and effective HBM rate (bytes the passes must move: backward reduce dy + z (+ y), apply dy + z (+ y) + dz; forward z + out)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import lib, check, ptr, stream, BN_ACC_REPLICAS
from yolo.engine import Act

N = 64
dev = torch.device("cuda")
shapes = [(112, 64), (112, 256), (56, 128), (56, 512), (28, 256), (28, 1024), (14, 512), (14, 2048)]
for hw, C in shapes:
    for from_z in (1, 0):
        dy, y, z, dz = (Act(N, hw, hw, C, 1, dev) for _ in range(4))
        for a in (dy, y, z):
            a.interior().normal_()
        gamma = torch.rand(C, device=dev) + 0.5
        mis = torch.zeros(4 * C, device=dev)
        mis[C:2 * C] = 1.0
        mis[2 * C:3 * C] = 1.0
        dg, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        acc = torch.zeros(BN_ACC_REPLICAS * 2 * C, dtype=torch.float64, device=dev)
        coef = torch.zeros(3 * C, device=dev)

        def run():
            check(lib().yolo_batchnorm_bwd(dy.p, 1, None if from_z else y.p, 1, z.p, 1, N, hw, hw, C, ptr(gamma), ptr(mis), dz.p, dz.img_stride, dz.row_stride,
                                           dz.px_stride, dz.interior_off(), 0, from_z, ptr(dg), ptr(dbt), ptr(acc), ptr(coef), stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        el = N * hw * hw * C * 2
        byts = el * ((2 if from_z else 3) + (3 if from_z else 4))
        print(f"{hw:4d} x {hw:4d} x {C:5d} relu_from_z {from_z}: {ms:7.3f} ms  {byts / ms / 1e9:6.2f} TB/s effective ({byts / 1e6:7.0f} MB)")
        del dy, y, z, dz
