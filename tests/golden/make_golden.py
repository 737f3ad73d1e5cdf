#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING the reference (dev container only).

    python tests/golden/make_golden.py [--ref /root/reference]

The reference (mattiaskvist/yolo-v1) cannot travel to the GPU box, so its outputs are frozen
here as data: inputs + expected outputs, nothing of its source.  Import recipe = SURVEY.md 8(c):
  * src/yolo/loss.py, metrics.py, schemas.py import as-is (torch / numpy / tqdm / pydantic);
  * src/yolo/inference.py and models.py do `import torchvision` at module scope for names that
    the functions frozen here never execute (transforms.Compose for the PIL pre-process,
    resnet50 for ResNetBackbone).  torchvision is not installed, so an EMPTY name holder is
    registered for the duration of this script only.  No arithmetic goes through it:
    decode / IoU / NMS are pure Python+torch, YOLOv1Backbone is stock torch.nn.
    ResNetBackbone is NOT constructed (its arithmetic lives in torchvision 0.23.0, absent) ->
    ResNet50 numerics stay "parity unpinned" (DESIGN.md).
Outputs (all small, committed):
  loss_cases.npz      YOLOLoss.forward + autograd dL/dpred            (src/yolo/loss.py:55-212)
  post_cases.npz      decode / IoU / NMS, both variants                (inference.py:141-317, metrics.py:173-341)
  map_case.npz/.json  mAPMetric.update+compute on 16 synthetic images  (metrics.py:78-171,343-651)
  layers_small.npz    stock torch conv/pool/linear I/O at tiny sizes   (models.py layer hyper-params)
  backbone_full.npz   reference YOLOv1() forward on one 448x448 image with synth weights:
                      per-layer checksums + final (7,7,30)             (models.py:47-84,239-245,256-276)
  dataset_cases.json/.npz  VOCDetectionYOLO._extract_bboxes_from_annotation / _encode_target / _parse_voc_annotation on
                      synthetic annotation dicts (dataset.py:411-532).  src/yolo/dataset.py imports torchvision.transforms.v2,
                      tv_tensors and datasets.VOCDetection at module scope; the three methods frozen here touch none of them
                      (plain Python floats + torch.zeros), so the same kind of empty name holder is registered and the
                      methods are called on an instance made WITHOUT __init__ (S, B, C, class_to_idx set as __init__ does,
                      dataset.py:186-194).
"""

from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402


def _load(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference(ref_root: str):
    src = os.path.join(ref_root, "src", "yolo")
    # name holder for the absent torchvision (see module docstring)
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.models.resnet50 = None
    tv.models.ResNet50_Weights = None
    tv.transforms = types.ModuleType("torchvision.transforms")
    for n in ("Compose", "Resize", "ToTensor", "Normalize"):
        setattr(tv.transforms, n, lambda *a, **k: None)
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tv.models)
    sys.modules.setdefault("torchvision.transforms", tv.transforms)
    pkg = types.ModuleType("yolo")
    pkg.__path__ = [src]
    sys.modules["yolo"] = pkg
    ref = types.SimpleNamespace()
    ref.schemas = _load("yolo.schemas", os.path.join(src, "schemas.py"))
    ref.loss = _load("yolo.loss", os.path.join(src, "loss.py"))
    ref.metrics = _load("yolo.metrics", os.path.join(src, "metrics.py"))
    ref.models = _load("yolo.models", os.path.join(src, "models.py"))
    ref.inference = _load("yolo.inference", os.path.join(src, "inference.py"))
    # dataset.py: more names of the absent torchvision, none of them executed by the methods frozen here
    tv.transforms.v2 = types.ModuleType("torchvision.transforms.v2")
    tv.tv_tensors = types.ModuleType("torchvision.tv_tensors")
    tv.datasets = types.ModuleType("torchvision.datasets")
    tv.datasets.VOCDetection = type("VOCDetection", (), {})
    sys.modules.setdefault("torchvision.transforms.v2", tv.transforms.v2)
    sys.modules.setdefault("torchvision.tv_tensors", tv.tv_tensors)
    sys.modules.setdefault("torchvision.datasets", tv.datasets)
    ref.dataset = _load("yolo.dataset", os.path.join(src, "dataset.py"))
    return ref


# --------------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------------
def loss_inputs():
    """name -> (pred, tgt, lambda_coord, lambda_noobj) as fp32 numpy."""
    cases = {}
    for seed in range(8):
        for N in (1, 4):
            p = synth.synth_normal((N, 7, 7, 30), 100 + seed, 0.5, seed) + 0.25
            t = synth.synth_targets(N, seed)
            cases[f"rand_s{seed}_n{N}"] = (p, t, 5.0, 0.5)
    cases["rand_s0_n64"] = (synth.synth_normal((64, 7, 7, 30), 200, 0.5, 0) + 0.25, synth.synth_targets(64, 100), 5.0, 0.5)
    # every image empty -> the three object terms are constant zeros (loss.py:139,146,159)
    cases["empty_n3"] = (synth.synth_normal((3, 7, 7, 30), 201, 0.5, 0), np.zeros((3, 7, 7, 30), np.float32), 5.0, 0.5)
    # predictions in [0,1] like a trained net, dense targets, non-default lambdas
    p = synth.synth_uniform((4, 7, 7, 30), 202, 0.5, 0) + 0.5
    cases["unit_dense_n4"] = (p, synth.synth_targets(4, 5, max_obj=30), 3.0, 0.25)
    # exact match of box 0 with the target (coord loss exactly 0; IoU path live in backward)
    t = synth.synth_targets(2, 9, max_obj=6)
    p = synth.synth_uniform((2, 7, 7, 30), 203, 0.4, 0) + 0.5
    obj = t[..., 4] > 0
    p[obj, 0:4] = t[obj, 0:4]
    p[obj, 4] = 0.2
    cases["exact_box0_n2"] = (p, t, 5.0, 0.5)
    # both predicted boxes identical -> IoU tie -> argmax picks box 0 (loss.py:110)
    p = synth.synth_uniform((2, 7, 7, 30), 204, 0.4, 0) + 0.5
    p[..., 5:9] = p[..., 0:4]
    cases["tie_boxes_n2"] = (p, t.copy(), 5.0, 0.5)
    # negative / tiny w,h -> clamp(min=1e-6) under sqrt has zero gradient (loss.py:132-133)
    p = synth.synth_normal((2, 7, 7, 30), 205, 0.5, 0)
    p[..., 2] = -np.abs(p[..., 2])
    p[..., 8] = 1e-7
    cases["neg_wh_n2"] = (p, t.copy(), 5.0, 0.5)
    # disjoint boxes: both IoUs exactly 0 -> argmax 0; clamp(min=0) boundary (loss.py:206-208)
    p = synth.synth_uniform((2, 7, 7, 30), 206, 0.01, 0)
    p[..., 0:2] += 5.0
    p[..., 5:7] -= 5.0
    p[..., 2:4] = 0.1
    p[..., 7:9] = 0.1
    cases["disjoint_n2"] = (p, t.copy(), 5.0, 0.5)
    # hand-made target that uses slot 1 only (conf1>0, conf0==0) -> target_box_idx 1
    t1 = np.zeros((1, 7, 7, 30), np.float32)
    t1[0, 2, 5, 5:10] = [0.3, 0.6, 0.4, 0.2, 1.0]
    t1[0, 2, 5, 10 + 7] = 1.0
    t1[0, 4, 1, 0:5] = [0.7, 0.2, 0.15, 0.5, 1.0]
    t1[0, 4, 1, 10 + 14] = 1.0  # class 14 = channel 24, part of the 4::5 slice
    cases["slot1_n1"] = (synth.synth_uniform((1, 7, 7, 30), 207, 0.5, 0) + 0.5, t1, 5.0, 0.5)
    # touching edges: pred x2 == target x1 exactly -> maximum/minimum tie (grad/2) and clamp at 0
    t2 = np.zeros((1, 7, 7, 30), np.float32)
    t2[0, 3, 3, 0:5] = [0.5, 0.5, 0.25, 0.25, 1.0]
    t2[0, 3, 3, 10] = 1.0
    p2 = np.zeros((1, 7, 7, 30), np.float32)
    p2[0, 3, 3, 0:5] = [0.25, 0.5, 0.25, 0.25, 0.5]      # x2 = 0.375 = target x1
    p2[0, 3, 3, 5:10] = [0.5, 0.5, 0.25, 0.5, 0.7]       # same x-extent as target -> max/min ties
    cases["edge_ties_n1"] = (p2, t2, 5.0, 0.5)
    return cases


def gen_loss(ref, out):
    names = []
    store = {}
    for name, (p, t, lc, ln) in loss_inputs().items():
        crit = ref.loss.YOLOLoss(S=7, B=2, C=20, lambda_coord=lc, lambda_noobj=ln)
        pt = torch.from_numpy(p.copy()).requires_grad_(True)
        tt = torch.from_numpy(t.copy())
        total, d = crit(pt, tt)
        total.backward()
        store[f"{name}__pred"] = p
        store[f"{name}__tgt"] = t
        store[f"{name}__lambdas"] = np.array([lc, ln], np.float64)
        store[f"{name}__out5"] = np.array([d["total"], d["coord"], d["conf_obj"], d["conf_noobj"], d["class"]], np.float64)
        store[f"{name}__dpred"] = pt.grad.numpy().copy()
        names.append(name)
        print(f"  loss {name:18s} total={d['total']:.6f}")
    # compute_iou static method on a few box pairs (loss.py:174-212)
    b1 = torch.from_numpy(synth.synth_uniform((2, 7, 7, 2, 4), 210, 0.5, 0) + 0.5)
    b2 = torch.from_numpy(synth.synth_uniform((2, 7, 7, 1, 4), 211, 0.5, 0) + 0.5)
    store["iou__b1"] = b1.numpy()
    store["iou__b2"] = b2.numpy()
    store["iou__out"] = ref.loss.YOLOLoss.compute_iou(b1, b2).numpy()
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(out, "loss_cases.npz"), **store)


# --------------------------------------------------------------------------------------------
# decode / IoU / NMS
# --------------------------------------------------------------------------------------------
class _Holder(torch.nn.Module):
    def __init__(self, S=7, B=2):
        super().__init__()
        self.S, self.B = S, B


def post_inputs():
    """name -> (pred (n,7,7,30) fp32, conf_thr, nms_thr, in_unit_range)"""
    rng = np.random.Generator(np.random.PCG64([0, 31]))
    cases = {}
    u = rng.uniform(0, 1, size=(8, 7, 7, 30)).astype(np.float32)
    for ct in (0.01, 0.3, 0.5):
        for nt in (0.3, 0.4, 0.5):
            cases[f"unit_c{ct}_n{nt}"] = (u, ct, nt, True)
    # "trained-like": sparse confidences, clustered boxes so that NMS really suppresses
    tl = rng.uniform(0, 1, size=(8, 7, 7, 30)).astype(np.float32)
    tl[..., 4] = rng.beta(0.5, 2.0, size=(8, 7, 7)).astype(np.float32)
    tl[..., 9] = rng.beta(0.5, 2.0, size=(8, 7, 7)).astype(np.float32)
    tl[..., 2:4] = rng.uniform(0.3, 0.9, size=(8, 7, 7, 2)).astype(np.float32)
    tl[..., 7:9] = rng.uniform(0.3, 0.9, size=(8, 7, 7, 2)).astype(np.float32)
    tl[..., 10:] = 0.05
    hot = rng.integers(0, 3, size=(8, 7, 7))
    for n in range(8):
        for i in range(7):
            for j in range(7):
                tl[n, i, j, 10 + hot[n, i, j]] = 0.9
    cases["trained_c0.1_n0.4"] = (tl, 0.1, 0.4, True)
    cases["trained_c0.01_n0.5"] = (tl, 0.01, 0.5, True)
    # equal confidences everywhere -> stable sort keeps scan order; identical boxes in neighbours
    eq = np.zeros((2, 7, 7, 30), np.float32)
    eq[..., 0:5] = [0.5, 0.5, 0.4, 0.4, 0.5]
    eq[..., 5:10] = [0.5, 0.5, 0.4, 0.4, 0.5]
    eq[..., 10] = 0.5
    eq[1, :, :, 12] = 0.5  # class tie -> argmax takes the first (class 0)
    eq[1, 3, :, 13] = 0.75
    cases["equal_conf"] = (eq, 0.2, 0.4, True)
    # raw network-like outputs (negative w/h, values outside [0,1]) -> metrics variant only
    raw = (rng.standard_normal(size=(6, 7, 7, 30)) * 0.4 + 0.3).astype(np.float32)
    cases["raw_c0.01_n0.4"] = (raw, 0.01, 0.4, False)
    cases["raw_c0.05_n0.3"] = (raw, 0.05, 0.3, False)
    # nothing survives the threshold
    cases["none_survive"] = (u[:2] * 0.1, 0.5, 0.4, True)
    return cases


def gen_post(ref, out):
    store = {}
    names = []
    inf = ref.inference.YOLOInference(_Holder(), device="cpu")
    for name, (pred, ct, nt, unit) in post_inputs().items():
        names.append(name)
        store[f"{name}__pred"] = pred
        store[f"{name}__thr"] = np.array([ct, nt], np.float64)
        met = ref.metrics.mAPMetric(num_classes=20, conf_threshold=ct, nms_threshold=nt)
        for n in range(pred.shape[0]):
            pt = torch.from_numpy(pred[n].copy())
            dets = met._parse_predictions(pt)
            rec = np.array([[c, f, *b] for c, f, b in dets], np.float64).reshape(-1, 6)
            store[f"{name}__m{n}_dec"] = rec
            kept = met._apply_nms(list(dets))
            # kept entries are the same tuple objects -> recover indices into the decoded list by identity
            ids = {id(d): k for k, d in enumerate(dets)}
            store[f"{name}__m{n}_keep"] = np.array([ids[id(d)] for d in kept], np.int32)
            if unit:
                idets = inf.parse_predictions(pt, ct)
                irec = np.array([[d.class_id, d.confidence, d.bbox.x, d.bbox.y, d.bbox.width, d.bbox.height] for d in idets], np.float64).reshape(-1, 6)
                assert irec.shape == rec.shape and np.array_equal(irec, rec), "the two decoders disagree"
                ikept = inf.non_max_suppression(list(idets), nms_threshold=nt)
                iids = {id(d): k for k, d in enumerate(idets)}
                store[f"{name}__i{n}_keep"] = np.array([iids[id(d)] for d in ikept], np.int32)
        print(f"  post {name:22s} imgs={pred.shape[0]}")
    # ground-truth parse (metrics.py:220-256)
    tg = synth.synth_targets(6, 3, max_obj=8)
    tg[0, 1, 1, 9] = 1.0  # slot-1-only cell: ignored by _parse_ground_truth (reads slot 0 only)
    store["gt__tgt"] = tg
    met = ref.metrics.mAPMetric(num_classes=20)
    for n in range(6):
        g = met._parse_ground_truth(torch.from_numpy(tg[n].copy()))
        store[f"gt__{n}"] = np.array([[c, *b] for c, b in g], np.float64).reshape(-1, 5)
    # scalar IoU known answers, both formulas (inference.py:212-249 with EPSILON; metrics.py:298-341 without)
    pairs = np.array([
        [0.5, 0.5, 0.3, 0.3, 0.5, 0.5, 0.3, 0.3],
        [0.2, 0.2, 0.1, 0.1, 0.8, 0.8, 0.1, 0.1],
        [0.5, 0.5, 0.4, 0.4, 0.6, 0.6, 0.4, 0.4],
        [0.3, 0.3, 0.2, 0.2, 0.4, 0.4, 0.2, 0.2],
        [0.5, 0.5, 0.0, 0.0, 0.5, 0.5, 0.2, 0.2],
        [0.5, 0.5, 0.0, 0.0, 0.5, 0.5, 0.0, 0.0],
        [0.5, 0.5, 0.2, 0.2, 0.52, 0.52, 0.2, 0.2],
        [0.25, 0.5, 0.25, 0.25, 0.5, 0.5, 0.25, 0.25],
    ], np.float64)
    rng = np.random.Generator(np.random.PCG64([0, 32]))
    pairs = np.concatenate([pairs, rng.uniform(0, 1, size=(56, 8)).astype(np.float32).astype(np.float64)])
    BB = ref.schemas.BoundingBox
    store["ioupairs__in"] = pairs
    store["ioupairs__metrics"] = np.array([met._calculate_iou(tuple(r[:4]), tuple(r[4:])) for r in pairs], np.float64)
    store["ioupairs__inference"] = np.array(
        [inf.iou(BB(x=r[0], y=r[1], width=r[2], height=r[3]), BB(x=r[4], y=r[5], width=r[6], height=r[7])) for r in pairs], np.float64)
    # crafted NMS lists (tuples; metrics variant accepts anything, inference variant needs [0,1])
    crafted = {
        "kat_metrics": ([(0, 0.9, (0.5, 0.5, 0.2, 0.2)), (0, 0.8, (0.52, 0.52, 0.2, 0.2)), (1, 0.85, (0.7, 0.7, 0.15, 0.15))], 0.5),
        "kat_inference": ([(0, 0.9, (0.5, 0.5, 0.3, 0.3)), (0, 0.7, (0.52, 0.52, 0.3, 0.3))], 0.3),
        "iou_eq_thr": ([(0, 0.9, (0.25, 0.5, 0.5, 0.5)), (0, 0.8, (0.5, 0.5, 0.5, 0.5)), (0, 0.7, (0.75, 0.5, 0.5, 0.5))], 1.0 / 3.0),
        "zero_area": ([(2, 0.9, (0.5, 0.5, 0.0, 0.0)), (2, 0.8, (0.5, 0.5, 0.0, 0.0)), (2, 0.7, (0.5, 0.5, 0.2, 0.2))], 0.4),
        "equal_conf_chain": ([(1, 0.5, (0.1 + 0.02 * k, 0.5, 0.2, 0.2)) for k in range(12)], 0.5),
        "three_classes": ([(k % 3, 0.3 + 0.05 * ((7 * k) % 11), (0.3 + 0.01 * k, 0.4, 0.3, 0.3)) for k in range(20)], 0.4),
    }
    cn = []
    for name, (dets, thr) in crafted.items():
        cn.append(name)
        store[f"craft_{name}__in"] = np.array([[c, f, *b] for c, f, b in dets], np.float64)
        store[f"craft_{name}__thr"] = np.array([thr], np.float64)
        met = ref.metrics.mAPMetric(num_classes=20, nms_threshold=thr)
        kept = met._apply_nms(list(dets))
        ids = {id(d): k for k, d in enumerate(dets)}
        store[f"craft_{name}__mkeep"] = np.array([ids[id(d)] for d in kept], np.int32)
        D = ref.schemas.Detection
        idets = [D(class_id=c, class_name=None, confidence=f, bbox=BB(x=b[0], y=b[1], width=b[2], height=b[3])) for c, f, b in dets]
        ikept = inf.non_max_suppression(list(idets), nms_threshold=thr)
        iids = {id(d): k for k, d in enumerate(idets)}
        store[f"craft_{name}__ikeep"] = np.array([iids[id(d)] for d in ikept], np.int32)
    # metrics variant only: negative width passes through un-validated
    neg = [(0, 0.9, (0.5, 0.5, -0.2, 0.2)), (0, 0.8, (0.5, 0.5, 0.2, 0.2)), (0, 0.7, (0.5, 0.5, -0.2, -0.2))]
    met = ref.metrics.mAPMetric(num_classes=20, nms_threshold=0.4)
    kept = met._apply_nms(list(neg))
    ids = {id(d): k for k, d in enumerate(neg)}
    store["craft_negw__in"] = np.array([[c, f, *b] for c, f, b in neg], np.float64)
    store["craft_negw__thr"] = np.array([0.4], np.float64)
    store["craft_negw__mkeep"] = np.array([ids[id(d)] for d in kept], np.int32)
    store["names"] = np.array(names)
    store["crafted"] = np.array(cn)
    np.savez_compressed(os.path.join(out, "post_cases.npz"), **store)


# --------------------------------------------------------------------------------------------
# mAP
# --------------------------------------------------------------------------------------------
def gen_map(ref, out):
    rng = np.random.Generator(np.random.PCG64([0, 41]))
    N = 16
    tgt = synth.synth_targets(N, 11, max_obj=5)
    pred = rng.uniform(0, 0.2, size=(N, 7, 7, 30)).astype(np.float32)
    # noisy copies of the ground truth in slot 0, distractors in slot 1
    obj = tgt[..., 4] > 0
    noise = rng.normal(0, 0.04, size=(N, 7, 7, 4)).astype(np.float32)
    pred[..., 0:4] = np.where(obj[..., None], np.clip(tgt[..., 0:4] + noise, 0.01, 0.99), pred[..., 0:4])
    pred[..., 4] = np.where(obj, rng.uniform(0.4, 1.0, size=(N, 7, 7)), pred[..., 4]).astype(np.float32)
    pred[..., 10:] = np.where(obj[..., None], np.clip(tgt[..., 10:] * 0.8 + rng.uniform(0, 0.3, size=(N, 7, 7, 20)), 0, 1), pred[..., 10:]).astype(np.float32)
    met = ref.metrics.mAPMetric(num_classes=20, conf_threshold=0.05, nms_threshold=0.4)
    met.update(torch.from_numpy(pred.copy()), torch.from_numpy(tgt.copy()))
    res = met.compute()
    np.savez_compressed(os.path.join(out, "map_case.npz"), pred=pred, tgt=tgt)
    with open(os.path.join(out, "map_case.json"), "w") as f:
        json.dump({k: float(v) for k, v in res.items()}, f, indent=1, sort_keys=True)
    print(f"  mAP  mAP50={res['mAP50']:.4f} mAP50:95={res['mAP50:95']:.4f} P={res['precision']:.4f} R={res['recall']:.4f}")


# --------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------
def gen_layers(ref, out):
    import torch.nn.functional as F
    store = {}
    k = 300
    specs = {  # name: (N, Cin, H, W, Cout, ksize, stride, pad)
        "c3x3": (2, 32, 10, 10, 64, 3, 1, 1),
        "c3x3s2": (2, 32, 14, 14, 32, 3, 2, 1),
        "c1x1": (2, 64, 9, 9, 32, 1, 1, 0),
        "c7x7s2": (1, 3, 32, 32, 64, 7, 2, 3),
    }
    for name, (N, ci, H, W, co, ks, st, pd) in specs.items():
        x = synth.synth_normal((N, ci, H, W), k, 1.0); k += 1
        w = synth.synth_uniform((co, ci, ks, ks), k, (3.0 / (ci * ks * ks)) ** 0.5); k += 1
        b = synth.synth_uniform((co,), k, 0.1); k += 1
        xt = torch.from_numpy(x.copy()).requires_grad_(True)
        wt = torch.from_numpy(w.copy()).requires_grad_(True)
        bt = torch.from_numpy(b.copy()).requires_grad_(True)
        y = F.leaky_relu(F.conv2d(xt, wt, bt, stride=st, padding=pd), 0.1)
        gy = torch.from_numpy(synth.synth_normal(tuple(y.shape), k, 1.0)); k += 1
        y.backward(gy)
        store.update({f"{name}__x": x, f"{name}__w": w, f"{name}__b": b, f"{name}__y": y.detach().numpy(), f"{name}__gy": gy.numpy(),
                      f"{name}__gx": xt.grad.numpy(), f"{name}__gw": wt.grad.numpy(), f"{name}__gb": bt.grad.numpy(),
                      f"{name}__cfg": np.array([ks, st, pd], np.int32)})
    x = synth.synth_normal((2, 32, 8, 8), k, 1.0); k += 1
    xt = torch.from_numpy(x.copy()).requires_grad_(True)
    y = F.max_pool2d(xt, 2, 2)
    gy = torch.from_numpy(synth.synth_normal(tuple(y.shape), k, 1.0)); k += 1
    y.backward(gy)
    store.update({"pool__x": x, "pool__y": y.detach().numpy(), "pool__gy": gy.numpy(), "pool__gx": xt.grad.numpy()})
    x = synth.synth_normal((4, 256), k, 1.0); k += 1
    w = synth.synth_uniform((96, 256), k, (3.0 / 256) ** 0.5); k += 1
    b = synth.synth_uniform((96,), k, 0.1); k += 1
    xt, wt, bt = (torch.from_numpy(a.copy()).requires_grad_(True) for a in (x, w, b))
    y = F.leaky_relu(F.linear(xt, wt, bt), 0.1)
    gy = torch.from_numpy(synth.synth_normal(tuple(y.shape), k, 1.0)); k += 1
    y.backward(gy)
    store.update({"fc__x": x, "fc__w": w, "fc__b": b, "fc__y": y.detach().numpy(), "fc__gy": gy.numpy(),
                  "fc__gx": xt.grad.numpy(), "fc__gw": wt.grad.numpy(), "fc__gb": bt.grad.numpy()})
    np.savez_compressed(os.path.join(out, "layers_small.npz"), **store)
    print("  layers_small done")


def gen_backbone(ref, out):
    """Reference YOLOv1() (YOLOv1Backbone + FC head, eval mode) on one synthetic image with the
    synth weights.  Stored: final (1,7,7,30), the (1,1024,7,7) backbone output, and per-module
    (mean, mean|.|, max|.|) so a wrong layer can be located."""
    torch.set_num_threads(8)
    model = ref.models.YOLOv1()
    sd = {k: torch.from_numpy(v) for k, v in synth.yolov1_state_dict().items()}
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = torch.from_numpy(synth.synth_images(1, 0))
    stats = {}
    hooks = []
    for idx, m in enumerate(model.backbone.features):
        hooks.append(m.register_forward_hook(lambda _m, _i, o, idx=idx: stats.__setitem__(idx, [o.mean().item(), o.abs().mean().item(), o.abs().max().item()])))
    with torch.no_grad():
        feat = model.backbone(x)
        y = model(x)
    for h in hooks:
        h.remove()
    # second check-point: same weights, batch of 2 different images, final output only
    x2 = torch.from_numpy(synth.synth_images(2, 7))
    with torch.no_grad():
        y2 = model(x2)
    np.savez_compressed(os.path.join(out, "backbone_full.npz"),
                        y=y.numpy(), feat=feat.numpy(), y2=y2.numpy(),
                        layer_idx=np.array(sorted(stats), np.int32),
                        layer_stats=np.array([stats[i] for i in sorted(stats)], np.float64))
    print(f"  backbone_full: y mean|.|={y.abs().mean().item():.4f} feat mean|.|={feat.abs().mean().item():.4f}")
    return model


# --------------------------------------------------------------------------------------------
# dataset: annotation dict -> boxes / class ids -> (S, S, 5B + C) target
# --------------------------------------------------------------------------------------------
def dataset_annotations():
    """name -> (annotation dict in the shape of torchvision's VOCDetection.parse_voc_xml, S, B)"""
    def obj(name, xmin, ymin, xmax, ymax):
        return {"name": name, "pose": "Unspecified", "truncated": "0", "difficult": "0",
                "bndbox": {"xmin": str(xmin), "ymin": str(ymin), "xmax": str(xmax), "ymax": str(ymax)}}

    def ann(w, h, objects):
        return {"annotation": {"folder": "VOC2007", "filename": "000001.jpg", "size": {"width": str(w), "height": str(h), "depth": "3"},
                               "segmented": "0", "object": objects}}

    cases = {
        "two_objects": (ann(500, 375, [obj("dog", 48, 240, 195, 371), obj("person", 8, 12, 352, 498)]), 7, 2),
        "single_object_not_a_list": (ann(353, 500, obj("cat", 1, 1, 353, 500)), 7, 2),                 # dataset.py:436-438
        "unknown_class_skipped": (ann(500, 375, [obj("unicorn", 10, 10, 100, 100), obj("sofa", 200, 100, 400, 300)]), 7, 2),
        "same_cell_first_wins": (ann(448, 448, [obj("car", 100, 100, 120, 120), obj("bus", 96, 96, 126, 126), obj("bird", 300, 300, 310, 330)]), 7, 2),
        "edges_and_clamps": (ann(500, 375, [obj("boat", 0, 0, 500, 375), obj("chair", 490, 365, 500, 375), obj("cow", 450, 0, 560, 40),
                                            obj("train", 250, 187.5, 250, 187.5)]), 7, 2),      # full frame, corner, beyond the frame, zero size
        "fractional_pixels": (ann(333, 250, [obj("aeroplane", 33.3, 20.25, 166.6, 199.75), obj("tvmonitor", 200.1, 10.9, 320.7, 120.2)]), 7, 2),
        "grid14_b3": (ann(640, 480, [obj("horse", 64, 48, 320, 240), obj("sheep", 321, 241, 639, 479), obj("bottle", 600, 10, 640, 60)]), 14, 3),
    }
    import random
    rnd = random.Random(7)
    names = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse", "motorbike",
             "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
    for k in range(6):
        w, h = rnd.choice([(500, 375), (375, 500), (500, 333), (480, 360), (320, 240)])
        objs = []
        for _ in range(rnd.randint(1, 9)):
            x0, y0 = rnd.randint(1, w - 20), rnd.randint(1, h - 20)
            objs.append(obj(rnd.choice(names), x0, y0, rnd.randint(x0 + 1, w), rnd.randint(y0 + 1, h)))
        cases[f"random_{k}"] = (ann(w, h, objs), 7, 2)
    return cases


def gen_dataset(ref, out):
    cls = ref.dataset.VOCDetectionYOLO
    store, meta = {}, {}
    for name, (annotation, S, B) in dataset_annotations().items():
        ds = object.__new__(cls)                     # no __init__: no torchvision, no files (see module docstring)
        ds.S, ds.B, ds.C = S, B, len(cls.VOC_CLASSES)
        ds.class_to_idx = {n: i for i, n in enumerate(cls.VOC_CLASSES)}
        bboxes, class_ids = ds._extract_bboxes_from_annotation(annotation)
        target = ds._parse_voc_annotation(annotation)
        assert torch.equal(target, ds._encode_target(bboxes, class_ids))
        meta[name] = {"annotation": annotation, "S": S, "B": B}
        store[f"{name}__bboxes"] = np.array(bboxes, np.float64).reshape(-1, 4)
        store[f"{name}__class_ids"] = np.array(class_ids, np.int64)
        store[f"{name}__target"] = target.numpy()
        print(f"  {name}: {len(class_ids)} objects, {int((target[..., 4] > 0).sum())} cells")
    np.savez_compressed(os.path.join(out, "dataset_cases.npz"), **store)
    with open(os.path.join(out, "dataset_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    ref = load_reference(a.ref)
    only = set(a.only.split(",")) if a.only else None
    for name, fn in (("loss", gen_loss), ("post", gen_post), ("map", gen_map), ("layers", gen_layers), ("backbone", gen_backbone), ("dataset", gen_dataset)):
        if only is None or name in only:
            print(f"[{name}]")
            fn(ref, HERE)


if __name__ == "__main__":
    main()
