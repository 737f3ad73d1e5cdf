#!/usr/bin/env python3
"""Merge two launch-plan tables by their IN-SITU times: every problem on which the tables differ keeps the plan that ran faster inside
the real passes (YOLOv1 inference + training step, ResNet-50 variant inference + training step at the given batch), measured with events
around each yolo_igemm call (engine.PLAN_TIMES) over interleaved repeats -- a single launch timed in isolation (what the tuner does) ranks
near-equal plans differently than the network does.
usage: merge_plans.py TABLE_A.json TABLE_B.json OUT.json [--batch 64] [--no-resnet]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from yolo import ResNetBackbone, YOLOLoss, YOLOv1, engine
from yolo import plans as P

ap = argparse.ArgumentParser()
ap.add_argument("a"); ap.add_argument("b"); ap.add_argument("out")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--no-resnet", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda")
B = a.batch
x = torch.randn(B, 3, 448, 448, device=dev)
tgt = torch.from_numpy(synth.synth_targets(B, 1)).to(dev)
crit = YOLOLoss()
tables = {}
for name in (a.a, a.b):
    tables[name] = {tuple(int(t) for t in k.split(",")): tuple(v) for k, v in json.load(open(name))["plans"].items()}


def passes(model, train):
    def run():
        if train:
            model.train()
            for p_ in model.parameters():
                p_.grad = None
            loss, _ = crit(model(x), tgt)
            loss.backward()
        else:
            model.eval()
            with torch.no_grad():
                model(x)
    return run


models = [("yolov1", YOLOv1().to(dev))]
if not a.no_resnet:
    models.append(("resnet50", YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=False)).to(dev)))
acc = {name: {} for name in tables}          # table -> key -> [ms per pass]
for mname, model in models:
    for train in (False, True):
        run = passes(model, train)
        for rep in range(3):
            for name, tab in tables.items():
                P._TUNED.clear()
                P._TUNED.update(tab)
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                engine.PLAN_TIMES = {}
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                rec, engine.PLAN_TIMES = engine.PLAN_TIMES, None
                for k, evs in rec.items():
                    acc[name].setdefault(k, []).append(sum(e0.elapsed_time(e1) for e0, e1 in evs) / 3)
        print(f"{mname} {'train' if train else 'inference'}: timed", flush=True)
ta, tb = tables[a.a], tables[a.b]
merged, moved, gain = dict(ta), 0, 0.0
for k in tb:
    if k not in ta:
        merged[k] = tb[k]
        continue
    if ta[k] == tb[k] or k not in acc[a.a] or k not in acc[a.b]:
        continue
    ma, mb = min(acc[a.a][k]), min(acc[a.b][k])
    if mb < 0.985 * ma:              # the second table's plan must win by more than the noise of the measurement
        merged[k] = tb[k]
        moved += 1
        gain += ma - mb
        print(f"{','.join(map(str, k))}: {ta[k]} {ma * 1e3:.1f} us -> {tb[k]} {mb * 1e3:.1f} us")
print(f"{moved} problems take the second table's plan, {gain * 1e3:.0f} us of in-situ time over the passes they appear in")
body = json.load(open(a.a))
body["plans"] = {",".join(map(str, k)): list(v) for k, v in sorted(merged.items())}
body["note"] = (body.get("note", "") + " | merged in situ with " + os.path.basename(a.b)).strip(" |")
with open(a.out, "w") as f:
    json.dump(body, f, indent=0, separators=(",", ":"))
    f.write("\n")
print("wrote", len(merged), "plans to", a.out)
