"""tests/test_gpu_resnet64.py::test_resnet_batch64_head_gradients taken apart: gradients of the DetectionHead behind the frozen ResNet trunk at batch 64
against the mean over eight batches of 8, with the batch-8 problems on (a) the shipped plans, (b) no batch-8 entries (plans borrowed from the neighbouring batch sizes, else the default rule; when this was
written: the default rule), (c) plain launches -- and how far the
PREDICTIONS of the sub-batches are from the batch-64 ones, since YOLOLoss's responsible-box choice turns a rounding of a prediction into a
different gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo import ResNetBackbone, YOLOLoss, YOLOv1, engine
from yolo import plans as P

torch.manual_seed(0)
g = YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=True)).cuda().eval()
for mod in g.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
g.head.train()
x = torch.from_numpy(synth.synth_images(64, 31)).cuda()
t = torch.from_numpy(synth.synth_targets(64, 33, max_obj=3)).cuda()
crit = YOLOLoss()
params = [(n, p) for n, p in g.head.named_parameters()]


def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def grads_of(xs, ts, dpred=None):
    for _, p in params:
        p.grad = None
    pred = g(xs)
    if dpred is None:
        loss, _ = crit(pred, ts)
        loss.backward()
    else:
        pred.backward(dpred)
    torch.cuda.synchronize()
    return pred.detach().clone(), {n: p.grad.detach().float().clone() for n, p in params}


shipped = dict(P._TUNED)
# dL/dpred of the batch-64 pass, to push through the sub-batches as well (teacher forcing: takes the loss's discontinuity out)
pred64 = g(x).detach().requires_grad_(True)
l64, _ = crit(pred64, t)
l64.backward()
dpred64 = pred64.grad.detach().clone()
p64, g64 = grads_of(x, t)
_, g64_tf = grads_of(x, t, dpred64)
print("batch 64: loss path vs teacher-forced path:", {n: round(rel(g64_tf[n], g64[n]), 5) for n in ("fc_layers.4.weight", "conv_layers.0.weight")})
for name, table, split in (("shipped", shipped, True), ("no batch-8 entries (borrowed from batch 16 / 4, else the rule)", {k: v for k, v in shipped.items() if k[0] != 8}, True),
                           ("plain", {k: v for k, v in shipped.items() if k[0] != 8}, False)):
    P._TUNED.clear(); P._TUNED.update(table); engine.SMALL_SPLIT = split
    acc = acc_tf = None
    dp = []
    for i in range(0, 64, 8):
        pi, gi = grads_of(x[i:i + 8], t[i:i + 8])
        _, gt = grads_of(x[i:i + 8], t[i:i + 8], dpred64[i:i + 8] * 8.0)       # (YOLOLoss divides by the local N)
        dp.append(rel(pi, p64[i:i + 8]))
        acc = gi if acc is None else {n: acc[n] + gi[n] for n in acc}
        acc_tf = gt if acc_tf is None else {n: acc_tf[n] + gt[n] for n in acc_tf}
    print(name, "| predictions of the sub-batches vs batch 64:", [round(v, 4) for v in dp])
    print("   free-running:", {n: round(rel(g64[n], acc[n] / 8), 4) for n in g64})
    print("   teacher-forced dL/dpred:", {n: round(rel(g64_tf[n], acc_tf[n] / 8), 4) for n in g64})
P._TUNED.clear(); P._TUNED.update(shipped); engine.SMALL_SPLIT = True
