"""Which measured plan changes a training step's gradients by more than rounding?  One forward + loss + backward at BATCH (MODEL=yolov1|resnet50, the
ResNet trunk frozen in eval mode as in tests/test_gpu_resnet64.py) with NO table entries (deterministic defaults, SMALL_SPLIT off: plain launches) as
the reference, then with ONE shipped entry at a time: prints the relative L2 difference of every head gradient for the entries that move any of them
by more than 1 %.

    BATCH=8 MODEL=resnet50 python tools/experiments/plan_bisect.py"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo import ResNetBackbone, YOLOLoss, YOLOv1, engine
from yolo import plans as P

B = int(os.environ.get("BATCH", "8"))
resnet = os.environ.get("MODEL", "resnet50") == "resnet50"
torch.manual_seed(0)
m = (YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=True)) if resnet else YOLOv1()).cuda().eval()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
m.head.train() if resnet else m.train()
x = torch.from_numpy(synth.synth_images(B, 31)).cuda()
t = torch.from_numpy(synth.synth_targets(B, 33, max_obj=3)).cuda()
crit = YOLOLoss()
params = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
shipped = {k: v for k, v in P._TUNED.items() if k[0] == B}


def grads(table, split):
    P._TUNED.clear()
    P._TUNED.update(table)
    engine.SMALL_SPLIT = split
    for _, p in params:
        p.grad = None
    loss, _ = crit(m(x), t)
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().float().clone() for n, p in params}, set(k for k in P._TUNED if k[0] == B)


def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


ref, used = grads({}, False)
again, _ = grads({}, False)
print("rerun of the reference:", max(rel(again[n], ref[n]) for n in ref))
allg, _ = grads(shipped, True)
print("whole shipped table:", {n: round(rel(allg[n], ref[n]), 4) for n in ref if rel(allg[n], ref[n]) > 0.01})
for k in sorted(used):
    if k not in shipped or tuple(shipped[k]) == (0, 0):
        continue
    g, _ = grads({k: shipped[k]}, False)
    d = {n: round(rel(g[n], ref[n]), 4) for n in ref}
    worst = max(d.values())
    print(k, shipped[k], "max", worst, ({n: v for n, v in d.items() if v > 0.01} if worst > 0.01 else ""))
