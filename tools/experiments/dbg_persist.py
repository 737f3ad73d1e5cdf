import sys, os
sys.path.insert(0, "yolo-v1_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import torch
from test_gpu_layers import _persist_problem
from yolo import engine
from yolo._hip import lib, ptr, stream
d, a_in, w, b, aux, a_out = _persist_problem(8, 64, 512, 3, 80, 96, "lrelu")
def run(pl):
    a_out.t.fill_(7.0)
    engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(w), ptr(b), None, a_out.p, stream(), "t")
    return a_out.interior().clone()
ref = run(("tile", 15, 1, 196)); got = run(("tile", 20, 1, 196))
bad = (got != ref)
print("bad frac", bad.float().mean().item())
idx = bad.nonzero()
print("channels bad histogram (per 16):", torch.bincount(idx[:, 3] // 16, minlength=32).tolist())
px = (idx[:, 0] * 80 * 96 + idx[:, 1] * 96 + idx[:, 2])
print("pixel slot within 196-tile histogram:", torch.bincount(px % 196, minlength=196).tolist())
print("tile index histogram (first 40):", torch.bincount(px // 196)[:40].tolist())
