#!/usr/bin/env python3
"""experiment: where the HOST time of one batch-64 training step goes (cProfile over 10 steps, GPU work not awaited)"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from yolo import YOLOv1, YOLOLoss
from yolo.optim import Adam
dev = torch.device("cuda")
model = YOLOv1().to(dev).train()
x = torch.randn(64, 3, 448, 448, device=dev)
tgt = torch.from_numpy(synth.synth_targets(64, seed=1)).to(dev)
crit = YOLOLoss()
opt = Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, max_grad_norm=10.0)
opt.attach_plan(model.hip_plan())
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = crit(model(x), tgt)
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
