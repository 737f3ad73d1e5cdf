import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
import torch
from yolo._hip import lib, check, ptr, stream, IgemmDesc, EPI_NONE, EPI_BIAS_LRELU, BN_ACC_REPLICAS
from yolo.engine import Act
dev = torch.device("cuda")
torch.manual_seed(0)
def run(N, H, ci, co, k, s, hint, tpx, stats, epi):
    p = (k - 1) // 2
    Ho = (H + 2 * p - k) // s + 1
    x = Act(N, H, H, ci, 1, dev); y = Act(N, Ho, Ho, co, 1, dev)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x.interior().copy_(torch.randn((N, H, H, ci), device=dev, generator=g).to(torch.bfloat16))
    w = (torch.randn((co, k, k, ci), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn((co,), device=dev, generator=g)
    d = IgemmDesc()
    d.N, d.Ho, d.Wo = N, Ho, Ho
    d.in_img_stride, d.in_row_stride, d.in_px_stride = x.img_stride, x.row_stride, x.px_stride
    d.in_off = x.interior_off(p); d.stride = s; d.KH = d.KW = k; d.tap_len = ci; d.Cout = co
    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = y.img_stride, y.row_stride, y.px_stride, y.interior_off()
    d.epilogue, d.slope, d.out_fp32, d.split_k = epi, 0.1, 0, 1
    d.tile_hint, d.tile_px = hint, tpx
    st = None
    if stats:
        st = torch.zeros(BN_ACC_REPLICAS * 2 * co, dtype=torch.float64, device=dev)
        d.bn_stats = st.data_ptr()
    check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b) if epi != EPI_NONE else None, None, y.p, stream()), f"hint {hint}")
    torch.cuda.synchronize()
    s_out = st.view(BN_ACC_REPLICAS, 2, co).sum(0) if stats else None
    return y.interior().float().clone(), s_out
cases = [(4, 112, 64, 256, 1, 1), (4, 112, 64, 64, 3, 1), (4, 112, 256, 128, 1, 1), (4, 112, 128, 128, 3, 2), (4, 56, 256, 512, 1, 2), (4, 56, 512, 128, 1, 1),
         (4, 28, 1024, 256, 1, 1), (4, 28, 256, 256, 3, 1), (4, 28, 256, 256, 3, 2), (8, 14, 512, 2048, 1, 1), (16, 14, 512, 512, 3, 1)]
for (N, H, ci, co, k, s) in cases:
    for stats, epi in ((False, EPI_BIAS_LRELU), (True, EPI_NONE)):
        ref, sref = run(N, H, ci, co, k, s, 5, 0, stats, epi)
        line = f"N {N} H {H} {ci}->{co} k{k} s{s} stats {int(stats)}:"
        for hint, tpx in ((14, 0), (14, 196), (15, 196), (12, 0)):
            if hint == 15 and stats: continue
            try:
                got, sg = run(N, H, ci, co, k, s, hint, tpx, stats, epi)
            except RuntimeError as e:
                line += f" h{hint}:{tpx} unsupported |"; continue
            rel = ((got - ref).norm() / ref.norm()).item()
            srel = ((sg - sref).norm() / sref.norm()).item() if stats else 0.0
            line += f" h{hint}:{tpx} rel {rel:.2e} stats {srel:.1e} |"
        print(line)
