"""GPU parity: HIP decode / IoU / NMS (through the C ABI) vs the oracle and the reference fixtures.
Bar: bit-exact records (fp64), class ids and kept indices."""

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from yolo import ops as _ops
    return _ops


def _run(ops, pred_np, ct, nt, variant):
    pred = torch.from_numpy(pred_np).cuda()
    return ops.postprocess_host(pred, ct, nt, variant, 7, 2, 20)


def test_golden_decode_nms_bit_exact(ops, golden):
    g = golden("post_cases.npz")
    for name in [str(n) for n in g["names"]]:
        pred = g[f"{name}__pred"]
        ct, nt = (float(v) for v in g[f"{name}__thr"])
        res_m = _run(ops, pred, ct, nt, 1)
        res_i = _run(ops, pred, ct, nt, 0)
        for n in range(pred.shape[0]):
            rec, keep = res_m[n]
            assert rec.shape == g[f"{name}__m{n}_dec"].shape, (name, n)
            assert np.array_equal(rec, g[f"{name}__m{n}_dec"]), (name, n)
            assert np.array_equal(keep, g[f"{name}__m{n}_keep"]), (name, n, "metrics order")
            if f"{name}__i{n}_keep" in g:
                assert np.array_equal(res_i[n][1], g[f"{name}__i{n}_keep"]), (name, n, "inference order")


def test_crafted_lists_bit_exact(ops, golden):
    g = golden("post_cases.npz")
    for name in [str(n) for n in g["crafted"]] + ["negw"]:
        rec = g[f"craft_{name}__in"]
        thr = float(g[f"craft_{name}__thr"][0])
        n = len(rec)
        rec_d = torch.zeros((1, 128, 6), dtype=torch.float64, device="cuda")
        rec_d[0, :n] = torch.from_numpy(rec).cuda()
        cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
        keep, kc = ops.nms(rec_d, cnt, thr, 1)
        assert np.array_equal(keep[0, : int(kc[0])].cpu().numpy(), g[f"craft_{name}__mkeep"]), name
        if f"craft_{name}__ikeep" in g:
            keep, kc = ops.nms(rec_d, cnt, thr, 0)
            assert np.array_equal(keep[0, : int(kc[0])].cpu().numpy(), g[f"craft_{name}__ikeep"]), name


def test_config5_batch64_vs_oracle(ops):
    """BASELINE config 5's post-processing input: (64,7,7,30) ~U(0,1), conf 0.3, nms 0.4."""
    rng = np.random.Generator(np.random.PCG64([0, 99]))
    pred = rng.uniform(0, 1, size=(64, 7, 7, 30)).astype(np.float32)
    for variant in (0, 1):
        res = _run(ops, pred, 0.3, 0.4, variant)
        tot = kept = 0
        for n in range(64):
            rec = O.decode(pred[n], 0.3)
            assert np.array_equal(res[n][0], rec), n
            assert np.array_equal(res[n][1], O.nms(rec, 0.4, variant)), (n, variant)
            tot += len(rec)
            kept += len(res[n][1])
        assert tot > 3000 and 0 < kept < tot


def test_raw_outputs_and_ties_vs_oracle(ops):
    """Un-normalised network-like outputs (negative w/h, conf > 1), heavy ties, empty images."""
    rng = np.random.Generator(np.random.PCG64([1, 99]))
    pred = (rng.standard_normal(size=(32, 7, 7, 30)) * 0.5 + 0.3).astype(np.float32)
    pred[5] = 0.0                      # nothing survives
    pred[6, ..., 4] = 0.5              # all-equal confidences
    pred[6, ..., 9] = 0.5
    pred[6, ..., 10:] = 0.5            # class tie -> argmax 0
    pred[7] = np.round(pred[7] * 4) / 4  # quantised -> many equal confidences and IoUs
    for variant in (0, 1):
        for ct, nt in ((0.01, 0.4), (0.1, 0.3), (-1.0, 0.5)):
            res = _run(ops, pred, ct, nt, variant)
            for n in range(pred.shape[0]):
                rec = O.decode(pred[n], ct)
                assert np.array_equal(res[n][0], rec), (n, ct)
                assert np.array_equal(res[n][1], O.nms(rec, nt, variant)), (n, variant, ct, nt)


def test_ground_truth_decode(ops, golden):
    g = golden("post_cases.npz")
    tg = g["gt__tgt"]
    rec, cnt = ops.decode_gt(torch.from_numpy(tg).cuda(), 7, 2, 20)
    rec, cnt = rec.cpu().numpy(), cnt.cpu().numpy()
    for n in range(tg.shape[0]):
        assert np.array_equal(rec[n, : cnt[n]], g[f"gt__{n}"])


def test_pairwise_iou_bit_exact(ops, golden):
    g = golden("post_cases.npz")
    pairs = g["ioupairs__in"]
    a = torch.from_numpy(pairs[:, :4]).cuda()
    b = torch.from_numpy(pairs[:, 4:]).cuda()
    for variant, key in ((0, "ioupairs__inference"), (1, "ioupairs__metrics")):
        m = ops.pairwise_iou(a, b, variant).cpu().numpy()
        assert np.array_equal(np.diag(m), g[key])
        # off-diagonal entries against the oracle
        for i in (0, 3, 17):
            for j in (1, 9, 40):
                assert m[i, j] == O.iou(pairs[i, :4], pairs[j, 4:], variant)


def test_other_grid_sizes(ops):
    """S=14 / B=3 / C=5 -> 588 candidates per image: decode loops, NMS limit respected."""
    rng = np.random.Generator(np.random.PCG64([2, 99]))
    S, B, C = 11, 1, 5
    pred = rng.uniform(0, 1, size=(3, S, S, B * 5 + C)).astype(np.float32)
    rec, cnt = ops.decode(torch.from_numpy(pred).cuda(), 0.2, S, B, C)
    rec, cnt = rec.cpu().numpy(), cnt.cpu().numpy()
    for n in range(3):
        assert np.array_equal(rec[n, : cnt[n]], O.decode(pred[n], 0.2, S, B, C))
    keep, kc = ops.nms(torch.from_numpy(rec).cuda(), torch.from_numpy(cnt).cuda(), 0.4, 1)
    for n in range(3):
        assert np.array_equal(keep[n, : int(kc[n])].cpu().numpy(), O.nms(rec[n, : cnt[n]], 0.4, 1))


def test_device_side_map_matching_equals_host_protocol():
    """mAPMetric with every update on the device -> TP/FP matching by yolo_map_match, compute() vectorised;
    must give the SAME floats as the host protocol (the reference's greedy matching restated in yolo.metrics) on the same
    accumulated lists -- all ~80 keys, incl. size buckets and overall precision / recall -- and handle images without
    predictions or without ground truth."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    import synth
    from yolo.metrics import mAPMetric
    rng = np.random.Generator(np.random.PCG64([3, 11]))
    m = mAPMetric(num_classes=20, conf_threshold=0.2)
    for b in range(3):
        tgt = synth.synth_targets(16, seed=40 + b)
        # predictions correlated with the targets (so that there are true positives at several thresholds) + noise
        pred = rng.uniform(0, 0.35, size=(16, 7, 7, 30)).astype(np.float32)
        obj = tgt[..., 4] > 0
        pred[obj, :5] = tgt[obj, :5] + rng.normal(0, 0.03, size=(int(obj.sum()), 5)).astype(np.float32)
        pred[obj, 5:10] = tgt[obj, :5] + rng.normal(0, 0.08, size=(int(obj.sum()), 5)).astype(np.float32)
        pred[obj, 10:] = tgt[obj, 10:] * 0.9 + 0.05
        if b == 1:
            pred[3] = 0.0          # an image without predictions
            tgt[5] = 0.0           # an image without ground truth
        m.update(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda())
    assert m._dev_images == 48 == len(m.all_predictions)
    fast = m.compute()
    m._dev_images = -1             # force the host protocol on the very same lists
    slow = m.compute()
    assert fast.keys() == slow.keys() and len(fast) > 70
    for k in fast:
        assert fast[k] == slow[k], (k, fast[k], slow[k])
    assert fast["mAP50"] > 0.3 and fast["num_small_objects"] + fast["num_medium_objects"] + fast["num_large_objects"] > 0
    # lists filled by hand (the reference's own tests do that) fall back to the host protocol
    m2 = mAPMetric(num_classes=20)
    m2.all_predictions.append([(0, 0.9, (0.5, 0.5, 0.2, 0.2))])
    m2.all_ground_truths.append([(0, (0.5, 0.5, 0.2, 0.2))])
    assert abs(m2.compute()["mAP50"] - 1.0 / 20) < 1e-6


def test_nms_beyond_128_boxes_bit_exact(ops):
    """grids with S*S*B > 128 (the reference's tests build S = 14, B = 3: tests/test_backbone.py:141-173 -> 588 boxes per
    image): yolo_nms sweeps row by row with a 1024-bit mask -- records, kept indices and both output orders equal the
    oracle's; confidence ties and equal boxes included; more than 1024 boxes are refused loudly."""
    S, B, C = 14, 3, 20
    rng = np.random.Generator(np.random.PCG64([14, 3]))
    pred = rng.uniform(0, 1, size=(5, S, S, 5 * B + C)).astype(np.float32)
    pred[1, :, :, 4::5][:, :, :B] = np.float32(0.5)          # whole image at one confidence: ties keep the scan order
    pred[2, ...] = pred[2, 0:1, 0:1, :]                      # every cell predicts the same box offsets and class
    for ct, nt in ((0.05, 0.4), (0.3, 0.5)):
        rec, cnt = ops.decode(torch.from_numpy(pred).cuda(), ct, S, B, C)
        assert rec.shape[1] == S * S * B == 588
        rec_h, cnt_h = rec.cpu().numpy(), cnt.cpu().numpy()
        assert cnt_h.max() > 128
        for variant in (0, 1):
            keep, kc = ops.nms(rec, cnt, nt, variant)
            keep_h, kc_h = keep.cpu().numpy(), kc.cpu().numpy()
            for n in range(pred.shape[0]):
                ref_rec = O.decode(pred[n], ct, S, B, C)
                assert np.array_equal(rec_h[n, : cnt_h[n]], ref_rec), (n, ct)
                assert np.array_equal(keep_h[n, : kc_h[n]], O.nms(ref_rec, nt, variant)), (n, ct, variant)
    too_many = torch.zeros((1, 1025, 6), dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError):
        ops.nms(too_many, torch.tensor([1025], dtype=torch.int32, device="cuda"), 0.4, 0)


def test_inference_nms_of_a_gpu_model_stays_on_the_device():
    """YOLOInference.non_max_suppression with a model on the GPU and more than 128 detections (S = 14, B = 3) runs yolo_nms,
    not a host loop, and returns the reference order (src/yolo/inference.py:283-317)."""
    import torch.nn as nn
    from yolo.inference import YOLOInference
    from yolo.schemas import BoundingBox, Detection

    class Tiny(nn.Module):
        S, B = 14, 3

        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(1))

    inf = YOLOInference(Tiny().cuda(), device="cuda")
    rng = np.random.Generator(np.random.PCG64([5, 88]))
    dets = [Detection(bbox=BoundingBox(x=float(rng.uniform(0.2, 0.8)), y=float(rng.uniform(0.2, 0.8)), width=float(rng.uniform(0.1, 0.4)),
                                       height=float(rng.uniform(0.1, 0.4))), confidence=float(rng.uniform(0.05, 1.0)), class_id=int(rng.integers(0, 4)))
            for _ in range(300)]
    rec = np.array([[d.class_id, d.confidence, d.bbox.x, d.bbox.y, d.bbox.width, d.bbox.height] for d in dets], np.float64)
    import yolo._post_cpu as pc
    called = []
    orig = pc.nms
    pc.nms = lambda *a, **k: called.append(1) or orig(*a, **k)
    try:
        kept = inf.non_max_suppression(dets, nms_threshold=0.4)
    finally:
        pc.nms = orig
    assert not called, "a GPU model must not fall back to the host NMS loop"
    assert [dets.index(k) for k in kept] == O.nms(rec, 0.4, 0).tolist()


def test_map_case_fixture_through_the_device_path():
    """SURVEY 8f-3 / a11 pinned to the reference: the reference-run fixture tests/golden/map_case.{npz,json} (make_golden.py
    ran the reference's mAPMetric on these tensors) fed through mAPMetric.update on DEVICE tensors -- yolo_decode, yolo_nms,
    yolo_decode_gt and yolo_map_match -- must reproduce every key of the reference's compute() dictionary."""
    import json, os
    from yolo.metrics import mAPMetric
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = np.load(os.path.join(here, "map_case.npz"))
    ref = json.load(open(os.path.join(here, "map_case.json")))
    m = mAPMetric(20, conf_threshold=0.05, nms_threshold=0.4)
    pred, tgt = torch.from_numpy(d["pred"]).cuda(), torch.from_numpy(d["tgt"]).cuda()
    half = pred.shape[0] // 2
    m.update(pred[:half], tgt[:half])          # two device batches
    m.update(pred[half:], tgt[half:])
    assert m._dev_images == pred.shape[0] == len(m.all_predictions), "the device matching path must have been taken"
    res = m.compute()
    assert set(res) == set(ref)
    for k, v in ref.items():
        assert float(res[k]) == pytest.approx(v, abs=1e-12), k
