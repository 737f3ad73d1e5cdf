#!/usr/bin/env python3
"""sweep of the split count for the three FC1 (4096 x 50176, batch 64) kernels: forward (yolo_igemm split-K),
weight gradient and data gradient (both yolo_wgrad).  HBM-bound: 411 MB bf16 weight stream / 822 MB fp32 gradient."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import lib, check, ptr, stream, WgradDesc, IgemmDesc, EPI_NONE
N, O, K = 64, 4096, 50176
dev = torch.device("cuda")
w = (torch.randn(O, K, device=dev) * 0.01).to(torch.bfloat16)
x = torch.randn(N, K, device=dev).to(torch.bfloat16)
g = torch.randn(N, O, device=dev).to(torch.bfloat16)
gT = g.t().contiguous()
dw = torch.zeros(O, K, device=dev)
dxT = torch.zeros(K, N, device=dev)
acc = torch.zeros(N, O, device=dev)

def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for split in (0, 1, 2, 3, 4, 6, 8):
    wd = WgradDesc(O, K, N, K, N, 1, 1, 0, 0, split, 1)
    ms = timeit(lambda: check(lib().yolo_wgrad(ctypes.byref(wd), ptr(gT), ptr(w), ptr(dxT), None, stream())))
    print(f"fc1 dgrad (wgrad kernel) split {split}: {ms:.4f} ms  {O * K * 2 / ms / 1e6:.0f} GB/s of weights")
for split in (1, 0):
    wd = WgradDesc(N, O, K, O, K, 1, 1, 0, 0, split, 0)
    ms = timeit(lambda: check(lib().yolo_wgrad(ctypes.byref(wd), ptr(x), ptr(g), ptr(dw), None, stream())))
    print(f"fc1 wgrad split {split}: {ms:.4f} ms  {O * K * 4 / ms / 1e6:.0f} GB/s of gradient")
for splits in (8, 16, 24, 32, 48):
    d = IgemmDesc()
    d.N, d.Ho, d.Wo = N, 1, 1
    d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = K, 0, K, 0
    d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 1, 1, K, O
    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = O, 0, O, 0
    d.slope, d.out_fp32, d.epilogue, d.split_k = 0.1, 1, EPI_NONE, splits
    ms = timeit(lambda: check(lib().yolo_igemm(ctypes.byref(d), ptr(x), ptr(w), None, None, ptr(acc), stream())))
    print(f"fc1 forward (plain [O][K] weights) split_k {splits}: {ms:.4f} ms  {O * K * 2 / ms / 1e6:.0f} GB/s")

# blocked weight panels (the layout the engine uses for inference): [ceil(O/128)][K/64][128][64]
wf32 = w.float()
panels = torch.empty(((O + 127) // 128) * (K // 64) * 128 * 64, dtype=torch.bfloat16, device=dev)
check(lib().yolo_pack_fc_weight_blocked(ptr(wf32), O, K, ptr(panels), stream()))
del wf32
for hint in (0, 3, 4):
    for splits in (24, 48, 96):
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, 1, 1
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = K, 0, K, 0
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 1, 1, K, O
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = O, 0, O, 0
        d.slope, d.out_fp32, d.epilogue, d.split_k, d.w_blocked, d.tile_hint = 0.1, 1, EPI_NONE, splits, 1, hint
        ms = timeit(lambda: check(lib().yolo_igemm(ctypes.byref(d), ptr(x), ptr(panels), None, None, ptr(acc), stream())))
        print(f"fc1 forward (blocked panels) hint {hint} split_k {splits}: {ms:.4f} ms  {O * K * 2 / ms / 1e6:.0f} GB/s")
