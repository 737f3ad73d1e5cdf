import os, sys, time
sys.path.insert(0, "yolo-v1_amd"); sys.path.insert(0, "tests/golden")
import torch, synth
from yolo import YOLOv1, YOLOLoss, engine
from yolo.optim import Adam
dev = torch.device("cuda")
torch.manual_seed(0)
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
for side in (True, False, True, False):
    engine.WGRAD_STREAM = side
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"side stream {side}: host enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step", flush=True)
# host time of ONE step issued into an empty queue (no back-pressure from the HIP queue): is the step CPU-bound?
for side in (True, False):
    engine.WGRAD_STREAM = side
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        ts.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
    print(f"side stream {side}: host time of one step into an empty queue: " + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
