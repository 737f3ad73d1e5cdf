"""CPU: pin oracle/ (the C restatement) to the reference's behaviour.

Fixtures come from running the reference itself (tests/golden/make_golden.py); the known-answer
tests restate the ones in the reference's own suite (tests/test_yolo.py:196-313,
tests/test_metrics.py:35-117,208-222 of mattiaskvist/yolo-v1).
"""

import numpy as np
import pytest

from oracle import oracle as O


# ------------------------------------------------------------------ loss
def _loss_names(golden):
    return [str(n) for n in golden("loss_cases.npz")["names"]]


def test_loss_forward_and_grad_match_reference(golden):
    g = golden("loss_cases.npz")
    for name in _loss_names(golden):
        lc, ln = g[f"{name}__lambdas"]
        out5, dpred = O.loss_fwd_bwd(g[f"{name}__pred"], g[f"{name}__tgt"], lambda_coord=float(lc), lambda_noobj=float(ln))
        ref5 = g[f"{name}__out5"]
        # north_star tolerance: <= 1e-4 on the fp32 loss
        np.testing.assert_allclose(out5, ref5, rtol=1e-5, atol=1e-6, err_msg=name)
        assert np.max(np.abs(out5 - ref5)) <= 1e-4 * max(1.0, np.max(np.abs(ref5))), name
        np.testing.assert_allclose(dpred, g[f"{name}__dpred"], rtol=2e-5, atol=2e-6, err_msg=name)


def test_loss_iou_matches_reference(golden):
    g = golden("loss_cases.npz")
    out = O.loss_iou(g["iou__b1"], g["iou__b2"])
    np.testing.assert_allclose(out, g["iou__out"], rtol=1e-6, atol=1e-7)


def test_loss_bad_target_slot_raises():
    # a class channel in the 4::5 slice set while both conf slots are 0 -> reference gather raises
    t = np.zeros((1, 7, 7, 30), np.float32)
    t[0, 1, 1, 14] = 1.0
    with pytest.raises(RuntimeError):
        O.loss_fwd_bwd(np.zeros((1, 7, 7, 30), np.float32), t)


# ------------------------------------------------------------------ decode / NMS
def test_decode_and_nms_match_reference(golden):
    g = golden("post_cases.npz")
    n_boxes = 0
    for name in [str(n) for n in g["names"]]:
        pred = g[f"{name}__pred"]
        ct, nt = g[f"{name}__thr"]
        for n in range(pred.shape[0]):
            rec = O.decode(pred[n], ct)
            ref = g[f"{name}__m{n}_dec"]
            assert rec.shape == ref.shape, (name, n)
            assert np.array_equal(rec, ref), (name, n)  # bit-exact doubles, exact class ids
            keep = O.nms(rec, nt, O.METRICS)
            assert np.array_equal(keep, g[f"{name}__m{n}_keep"]), (name, n, "metrics")
            if f"{name}__i{n}_keep" in g:
                keep = O.nms(rec, nt, O.INFERENCE)
                assert np.array_equal(keep, g[f"{name}__i{n}_keep"]), (name, n, "inference")
            n_boxes += len(rec)
    assert n_boxes > 3000


def test_crafted_nms_lists(golden):
    g = golden("post_cases.npz")
    for name in [str(n) for n in g["crafted"]] + ["negw"]:
        rec = g[f"craft_{name}__in"]
        thr = float(g[f"craft_{name}__thr"][0])
        assert np.array_equal(O.nms(rec, thr, O.METRICS), g[f"craft_{name}__mkeep"]), name
        if f"craft_{name}__ikeep" in g:
            assert np.array_equal(O.nms(rec, thr, O.INFERENCE), g[f"craft_{name}__ikeep"]), name


def test_ground_truth_parse(golden):
    g = golden("post_cases.npz")
    tg = g["gt__tgt"]
    for n in range(tg.shape[0]):
        assert np.array_equal(O.decode_gt(tg[n]), g[f"gt__{n}"])


def test_scalar_iou_pairs(golden):
    g = golden("post_cases.npz")
    pairs = g["ioupairs__in"]
    for r, m, i in zip(pairs, g["ioupairs__metrics"], g["ioupairs__inference"]):
        assert O.iou(r[:4], r[4:], O.METRICS) == m
        assert O.iou(r[:4], r[4:], O.INFERENCE) == i


# ---- known-answer tests restated from the reference's own suite ----
def test_kat_iou_reference_suite():
    # reference tests/test_yolo.py:196-227
    box = (0.5, 0.5, 0.3, 0.3)
    assert O.iou(box, box, O.INFERENCE) == pytest.approx(1.0, abs=1e-4)
    assert O.iou((0.2, 0.2, 0.1, 0.1), (0.8, 0.8, 0.1, 0.1), O.INFERENCE) == pytest.approx(0.0, abs=1e-5)
    assert 0 < O.iou((0.5, 0.5, 0.4, 0.4), (0.6, 0.6, 0.4, 0.4), O.INFERENCE) < 1
    a, b = (0.3, 0.3, 0.2, 0.2), (0.4, 0.4, 0.2, 0.2)
    assert O.iou(a, b, O.INFERENCE) == pytest.approx(O.iou(b, a, O.INFERENCE), abs=1e-5)
    # reference tests/test_metrics.py:35-55,208-222
    assert O.iou((0.5, 0.5, 0.2, 0.2), (0.5, 0.5, 0.2, 0.2), O.METRICS) == pytest.approx(1.0, abs=1e-5)
    assert O.iou((0.2, 0.2, 0.1, 0.1), (0.8, 0.8, 0.1, 0.1), O.METRICS) == 0.0
    assert O.iou((0.5, 0.5, 0.0, 0.0), (0.5, 0.5, 0.2, 0.2), O.METRICS) == 0.0
    assert O.iou((0.5, 0.5, 0.0, 0.0), (0.5, 0.5, 0.0, 0.0), O.METRICS) == 0.0


def test_kat_nms_reference_suite():
    # reference tests/test_metrics.py:98-117
    rec = np.array([[0, 0.9, 0.5, 0.5, 0.2, 0.2], [0, 0.8, 0.52, 0.52, 0.2, 0.2], [1, 0.85, 0.7, 0.7, 0.15, 0.15]])
    keep = O.nms(rec, 0.5, O.METRICS)
    assert len(keep) == 2 and [rec[k, 1] for k in keep if rec[k, 0] == 0] == [0.9]
    # reference tests/test_yolo.py:229-313
    assert len(O.nms(np.zeros((0, 6)), 0.5, O.INFERENCE)) == 0
    one = np.array([[0, 0.9, 0.5, 0.5, 0.3, 0.3]])
    assert list(O.nms(one, 0.5, O.INFERENCE)) == [0]
    two = np.array([[0, 0.9, 0.5, 0.5, 0.3, 0.3], [0, 0.7, 0.52, 0.52, 0.3, 0.3]])
    assert list(O.nms(two, 0.3, O.INFERENCE)) == [0]
    two[1, 0] = 1
    assert len(O.nms(two, 0.3, O.INFERENCE)) == 2
    far = np.array([[0, 0.9, 0.2, 0.2, 0.1, 0.1], [0, 0.8, 0.8, 0.8, 0.1, 0.1]])
    assert len(O.nms(far, 0.5, O.INFERENCE)) == 2


def test_kat_decode_reference_suite():
    # reference tests/test_yolo.py:85-124 and tests/test_metrics.py:57-96
    pred = np.zeros((7, 7, 30), np.float32)
    pred[2, 3, 0:5] = [0.5, 0.5, 0.3, 0.3, 0.9]
    pred[2, 3, 10] = 0.8
    rec = O.decode(pred, 0.5)
    assert len(rec) == 1 and rec[0, 0] == 0
    assert rec[0, 1] == 0.7199999916553494  # 0.9f * 0.8f in double
    assert len(O.decode(pred, 0.9)) == 0
    tgt = np.zeros((7, 7, 30), np.float32)
    tgt[3, 3, 0:5] = [0.5, 0.5, 0.3, 0.3, 1.0]
    tgt[3, 3, 10] = 1.0
    gt = O.decode_gt(tgt)
    assert gt.shape == (1, 5) and gt[0, 0] == 0 and gt[0, 1] == pytest.approx(3.5 / 7)


# ------------------------------------------------------------------ layers
def test_naive_layers_match_torch_fixtures(golden):
    g = golden("layers_small.npz")
    for name in ("c3x3", "c3x3s2", "c1x1", "c7x7s2"):
        ks, st, pd = (int(v) for v in g[f"{name}__cfg"])
        y = O.conv2d(g[f"{name}__x"], g[f"{name}__w"], g[f"{name}__b"], st, pd, slope=0.1)
        np.testing.assert_allclose(y, g[f"{name}__y"], rtol=1e-4, atol=1e-5, err_msg=name)
    np.testing.assert_array_equal(O.maxpool2(g["pool__x"]), g["pool__y"])
    np.testing.assert_allclose(O.linear(g["fc__x"], g["fc__w"], g["fc__b"], slope=0.1), g["fc__y"], rtol=1e-4, atol=1e-5)


def test_pure_python_postprocessing_matches_reference_fixtures(golden):
    """oracle/post_py.py (the reference's per-scalar Python path, bench.py's second CPU leg) against the reference-run fixtures and
    the crafted tie lists: bit-exact records and kept indices, like the C oracle"""
    import torch
    from oracle import post_py as PY
    g = golden("post_cases.npz")
    n_boxes = 0
    for name in [str(n) for n in g["names"]][:6]:
        pred = g[f"{name}__pred"]
        ct, nt = g[f"{name}__thr"]
        for n in range(min(pred.shape[0], 4)):
            rec = PY.decode_py(torch.from_numpy(pred[n]), float(ct))
            assert np.array_equal(rec, g[f"{name}__m{n}_dec"]), (name, n)
            assert np.array_equal(PY.nms_py(rec, float(nt)), g[f"{name}__m{n}_keep"]), (name, n)
            n_boxes += len(rec)
    assert n_boxes > 300
    for name in [str(n) for n in g["crafted"]] + ["negw"]:
        rec = g[f"craft_{name}__in"]
        assert np.array_equal(PY.nms_py(rec, float(g[f"craft_{name}__thr"][0])), g[f"craft_{name}__mkeep"]), name
