"""CPU: the module surface behaves like the reference's (restates the assertions of the reference's
own suite: tests/test_backbone.py, tests/test_yolo.py, tests/test_metrics.py of mattiaskvist/yolo-v1)
and the host logic (numpy post-processing, mAP) agrees with the reference fixtures."""

import json
import os
import tempfile

import numpy as np
import pytest
import torch
from PIL import Image

import synth
from yolo import (Backbone, BoundingBox, Detection, DetectionHead, YOLOLoss, YOLOv1, YOLOv1Backbone, mAPMetric)
from yolo.inference import YOLOInference

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    return YOLOv1(num_classes=20, S=7, B=2).eval()


@pytest.fixture(scope="module")
def engine(model):
    return YOLOInference(model, device="cpu")


@pytest.fixture()
def sample_image():
    with tempfile.NamedTemporaryFile(suffix=".jpg", delete=False) as tmp:
        Image.new("RGB", (448, 448), color="red").save(tmp.name)
    yield tmp.name
    os.unlink(tmp.name)


# ---------------------------------------------------------------- models (reference tests/test_backbone.py)
def test_backbone_base_raises():
    with pytest.raises(NotImplementedError):
        Backbone()(torch.randn(1, 3, 448, 448))


def test_yolov1_shapes_and_state_dict(model):
    with torch.no_grad():
        y = model(torch.randn(2, 3, 448, 448))
        f = model.backbone(torch.randn(1, 3, 448, 448))
    assert y.shape == (2, 7, 7, 30) and not torch.isnan(y).any()
    assert f.shape == (1, 1024, 7, 7)
    assert isinstance(model.backbone, YOLOv1Backbone)
    assert (model.num_classes, model.S, model.B) == (20, 7, 2)
    keys = list(model.state_dict().keys())
    conv_idx = [0, 3, 6, 8, 10, 12, 15, 17, 19, 21, 23, 25, 27, 29, 31, 33, 36, 38, 40, 42, 44, 46, 48, 50]
    want = [f"backbone.features.{i}.{p}" for i in conv_idx for p in ("weight", "bias")] + ["head.1.weight", "head.1.bias", "head.4.weight", "head.4.bias"]
    assert keys == want                                   # SURVEY.md 8b checkpoint contract
    assert sum(p.numel() for p in model.parameters()) == 271_703_550
    assert model.state_dict()["head.1.weight"].shape == (4096, 50176)


def test_constructor_variants():
    for S in (7, 14):
        assert YOLOv1(S=S).S == S
    class Custom(Backbone):
        def forward(self, x):
            return x
    with pytest.raises(ValueError):
        YOLOv1(backbone=Custom())
    m = YOLOv1(backbone=Custom(), detection_head=torch.nn.Identity())
    assert isinstance(m.head, torch.nn.Identity)


def test_detection_head_shapes():
    head = DetectionHead(64, num_classes=20, S=7, B=2)      # narrow input: same structure, affordable on CPU
    assert list(head.state_dict().keys())[:2] == ["conv_layers.0.weight", "conv_layers.0.bias"]
    assert "fc_layers.4.bias" in head.state_dict()
    with torch.no_grad():
        assert head(torch.randn(2, 64, 14, 14)).shape == (2, 7, 7, 30)


def test_gradient_reaches_input(model):
    x = torch.randn(1, 3, 448, 448, requires_grad=True)
    model(x).sum().backward()
    assert x.grad is not None and not torch.isnan(x.grad).any()


def test_loads_the_reference_fixture_weights_and_reproduces_its_output(model):
    """same state_dict + same input as the run of the reference that produced backbone_full.npz"""
    import copy
    m = copy.deepcopy(model)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.yolov1_state_dict().items()}, strict=True)
    g = np.load(os.path.join(GOLDEN, "backbone_full.npz"))
    with torch.no_grad():
        y = m.eval()(torch.from_numpy(synth.synth_images(1, 0)))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- inference (reference tests/test_yolo.py)
def test_inference_engine(engine, model, sample_image):
    assert engine.model is model and engine.device == "cpu" and hasattr(engine, "transform") and not engine.model.training
    lo = engine.predict(sample_image, conf_threshold=0.1, nms_threshold=0.4)
    hi = engine.predict(sample_image, conf_threshold=0.9, nms_threshold=0.4)
    assert isinstance(lo, list) and len(hi) <= len(lo)
    with pytest.raises(FileNotFoundError):
        engine.predict("nonexistent_image.jpg")
    t = engine.preprocess_image(engine.load_image(sample_image))
    assert t.shape == (1, 3, 448, 448) and -5 <= t.min() and t.max() <= 5


def test_parse_predictions_kat(engine):
    pred = torch.zeros(7, 7, 30)
    pred[2, 3, 0:5] = torch.tensor([0.5, 0.5, 0.3, 0.3, 0.9])
    pred[2, 3, 10] = 0.8
    dets = engine.parse_predictions(pred, conf_threshold=0.5)
    assert len(dets) == 1 and isinstance(dets[0], Detection) and isinstance(dets[0].class_id, int)
    assert dets[0].confidence == 0.7199999916553494 and dets[0].class_name == "class_0"
    assert dets[0].bbox.x == pytest.approx(3.5 / 7) and dets[0].bbox.y == pytest.approx(2.5 / 7)
    assert len(engine.parse_predictions(pred, conf_threshold=0.9)) == 0
    bad = pred.clone()
    bad[2, 3, 2] = -0.3                                  # negative width -> pydantic ValidationError like the reference
    with pytest.raises(ValueError):
        engine.parse_predictions(bad, conf_threshold=0.5)


def test_bounding_box():
    b = BoundingBox(x=0.5, y=0.5, width=0.4, height=0.6)
    assert b.area == pytest.approx(0.24) and b.to_corners() == pytest.approx((0.3, 0.2, 0.7, 0.8))
    px = b.to_pixel_coords(640, 480)
    assert all(abs(a - e) <= 1 for a, e in zip(px, (192, 96, 448, 384)))
    with pytest.raises(ValueError):
        BoundingBox(x=1.5, y=0.5, width=0.3, height=0.3)
    r = BoundingBox.from_corners(0.1, 0.2, 0.5, 0.8)
    assert (r.x, r.y, r.width, r.height) == pytest.approx((0.3, 0.5, 0.4, 0.6))
    assert str(b) == "(0.30, 0.20, 0.70, 0.80)"


def test_iou_and_nms_kats(engine):
    box = BoundingBox(x=0.5, y=0.5, width=0.3, height=0.3)
    assert engine.iou(box, box) == pytest.approx(1.0, abs=1e-4)
    assert engine.iou(BoundingBox(x=0.2, y=0.2, width=0.1, height=0.1), BoundingBox(x=0.8, y=0.8, width=0.1, height=0.1)) == pytest.approx(0.0, abs=1e-5)
    a, b = BoundingBox(x=0.3, y=0.3, width=0.2, height=0.2), BoundingBox(x=0.4, y=0.4, width=0.2, height=0.2)
    assert 0 < engine.iou(a, b) < 1 and engine.iou(a, b) == pytest.approx(engine.iou(b, a), abs=1e-5)

    def det(c, conf, x, y, w, h):
        return Detection(class_id=c, class_name="t", confidence=conf, bbox=BoundingBox(x=x, y=y, width=w, height=h))
    with pytest.warns(DeprecationWarning):
        assert engine.non_max_suppression([], iou_threshold=0.5) == []
    one = [det(0, 0.9, 0.5, 0.5, 0.3, 0.3)]
    assert engine.non_max_suppression(one, nms_threshold=0.5) == one
    two = [det(0, 0.9, 0.5, 0.5, 0.3, 0.3), det(0, 0.7, 0.52, 0.52, 0.3, 0.3)]
    kept = engine.non_max_suppression(two, nms_threshold=0.3)
    assert len(kept) == 1 and kept[0].confidence == 0.9
    assert len(engine.non_max_suppression([two[0], det(1, 0.8, 0.52, 0.52, 0.3, 0.3)], nms_threshold=0.3)) == 2
    assert len(engine.non_max_suppression([det(0, 0.9, 0.2, 0.2, 0.1, 0.1), det(0, 0.8, 0.8, 0.8, 0.1, 0.1)])) == 2


def test_cpu_postprocessing_matches_reference_fixtures():
    """the numpy decode / NMS (CPU-device path of the package) vs post_cases.npz, bit-exact"""
    from yolo import _post_cpu as P
    g = np.load(os.path.join(GOLDEN, "post_cases.npz"))
    for name in [str(n) for n in g["names"]]:
        pred = g[f"{name}__pred"]
        ct, nt = g[f"{name}__thr"]
        for n in range(pred.shape[0]):
            rec = P.decode(pred[n], ct, 7, 2)
            assert np.array_equal(rec, g[f"{name}__m{n}_dec"]), (name, n)
            assert np.array_equal(P.nms(rec, nt, P.METRICS), g[f"{name}__m{n}_keep"]), (name, n)
            if f"{name}__i{n}_keep" in g:
                assert np.array_equal(P.nms(rec, nt, P.INFERENCE), g[f"{name}__i{n}_keep"]), (name, n)
    for r, m, i in zip(g["ioupairs__in"], g["ioupairs__metrics"], g["ioupairs__inference"]):
        assert P.iou_scalar(r[:4], r[4:], P.METRICS) == m and P.iou_scalar(r[:4], r[4:], P.INFERENCE) == i


# ---------------------------------------------------------------- metrics (reference tests/test_metrics.py)
def test_metric_helpers_and_kats():
    m = mAPMetric(num_classes=20)
    assert (m.S, m.B, m.conf_threshold, m.nms_threshold, len(m.iou_thresholds)) == (7, 2, 0.01, 0.4, 10)
    assert m._calculate_iou((0.5, 0.5, 0.2, 0.2), (0.5, 0.5, 0.2, 0.2)) == pytest.approx(1.0, abs=1e-5)
    assert m._calculate_iou((0.2, 0.2, 0.1, 0.1), (0.8, 0.8, 0.1, 0.1)) == 0.0
    assert m._calculate_iou((0.5, 0.5, 0.0, 0.0), (0.5, 0.5, 0.0, 0.0)) == 0.0
    m = mAPMetric(num_classes=20, nms_threshold=0.5)
    kept = m._apply_nms([(0, 0.9, (0.5, 0.5, 0.2, 0.2)), (0, 0.8, (0.52, 0.52, 0.2, 0.2)), (1, 0.85, (0.7, 0.7, 0.15, 0.15))])
    assert len(kept) == 2 and [d[1] for d in kept if d[0] == 0] == [0.9]
    m = mAPMetric(num_classes=20, conf_threshold=0.1)
    pred = torch.zeros(7, 7, 30)
    pred[3, 3, 0:5] = torch.tensor([0.5, 0.5, 0.3, 0.3, 0.9])
    pred[3, 3, 10] = 1.0
    dets = m._parse_predictions(pred)
    assert len(dets) == 1 and dets[0][0] == 0 and dets[0][1] > 0.1
    gts = m._parse_ground_truth(pred)
    assert len(gts) == 1 and gts[0][0] == 0


def test_perfect_and_empty_predictions():
    m = mAPMetric(num_classes=20, iou_thresholds=[0.5])
    p = torch.zeros(5, 7, 7, 30)
    p[:, 3, 3, 0:5] = torch.tensor([0.5, 0.5, 0.3, 0.3, 1.0])
    p[:, 3, 3, 10] = 1.0
    m.update(p, p.clone())
    r = m.compute()
    for k in ("AP50_class_0", "AP50:95_class_0", "precision", "recall"):
        assert r[k] == pytest.approx(1.0, abs=1e-5)
    m = mAPMetric(num_classes=20, conf_threshold=0.9)
    q = torch.zeros(2, 7, 7, 30)
    q[:, 3, 3, 4] = 0.1
    m.update(q, p[:2])
    assert m.compute()["recall"] == 0.0
    assert mAPMetric(20).compute() == {"mAP50:95": 0.0, "mAP50": 0.0, "mAP75": 0.0, "precision": 0.0, "recall": 0.0}


def test_map_matches_reference_fixture():
    d = np.load(os.path.join(GOLDEN, "map_case.npz"))
    ref = json.load(open(os.path.join(GOLDEN, "map_case.json")))
    m = mAPMetric(20, conf_threshold=0.05, nms_threshold=0.4)
    m.update(torch.from_numpy(d["pred"]), torch.from_numpy(d["tgt"]))
    res = m.compute()
    assert set(res) == set(ref)
    for k, v in ref.items():
        assert float(res[k]) == pytest.approx(v, abs=1e-12), k


# ---------------------------------------------------------------- loss (CPU-device path vs reference fixtures)
def test_cpu_loss_matches_reference_fixtures():
    g = np.load(os.path.join(GOLDEN, "loss_cases.npz"))
    for name in [str(n) for n in g["names"]]:
        lc, ln = (float(v) for v in g[f"{name}__lambdas"])
        p = torch.from_numpy(g[f"{name}__pred"]).requires_grad_(True)
        total, d = YOLOLoss(lambda_coord=lc, lambda_noobj=ln)(p, torch.from_numpy(g[f"{name}__tgt"]))
        total.backward()
        got = np.array([d[k] for k in ("total", "coord", "conf_obj", "conf_noobj", "class")])
        np.testing.assert_allclose(got, g[f"{name}__out5"], rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(p.grad.numpy(), g[f"{name}__dpred"], rtol=1e-4, atol=1e-6, err_msg=name)
    t = torch.zeros(1, 7, 7, 30)
    t[0, 1, 1, 14] = 1.0
    with pytest.raises(RuntimeError):
        YOLOLoss()(torch.zeros(1, 7, 7, 30), t)


# ---------------------------------------------------------------- ResNet50 variant (reference tests/test_backbone.py:57-86,116-126)
def test_resnet_backbone_surface():
    from yolo import ResNetBackbone
    bb = ResNetBackbone(pretrained=False, freeze=True)
    assert all(not p.requires_grad for p in bb.parameters())
    assert all(p.requires_grad for p in ResNetBackbone(pretrained=False, freeze=False).parameters())
    assert sum(p.numel() for p in bb.parameters()) == 23_508_032          # resnet50 minus its fc layer
    keys = list(bb.state_dict().keys())
    assert keys[0] == "extractor.0.weight" and "extractor.4.0.downsample.0.weight" in keys and "extractor.7.2.bn3.running_var" in keys
    with torch.no_grad():
        f = bb.eval()(torch.randn(1, 3, 448, 448))
    assert f.shape == (1, 2048, 14, 14)
    m = YOLOv1(backbone=bb)
    assert isinstance(m.head, DetectionHead) and sum(p.numel() for p in m.head.parameters()) == 258_737_598
    with pytest.raises(ImportError):
        ResNetBackbone(pretrained=True)      # ImageNet weights need torchvision + a download


# ---------------------------------------------------------------- BASELINE config 0: predict.py --device cpu (plumbing)
def test_predict_script_on_cpu(sample_image, capsys):
    """single 448x448 image, YOLOv1Backbone, CPU forward through the predict.py entry point"""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("predict_cli", os.path.join(root, "yolo-v1_amd", "predict.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ["predict.py", sample_image, "--device", "cpu", "--backbone", "yolov1", "--conf-threshold", "0.01"]
    try:
        mod.main()
    finally:
        sys.argv = argv
    assert "detections" in capsys.readouterr().out


def test_checkpoint_with_map_metrics_loads_with_weights_only(tmp_path):
    """val_losses carries NumPy floats (np.mean of APs): the file must still load with torch.load(weights_only=True),
    which evaluate.py / predict.py / --resume use."""
    import numpy as np
    import torch
    import torch.nn as nn
    from yolo.training import checkpoints
    m = nn.Linear(4, 3)
    opt = torch.optim.Adam(m.parameters())
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [1])
    val = {"total": np.float64(1.5), "mAP50:95": np.float64(0.25), "mAP50": np.float64(0.5), "mAP75": np.float64(0.125)}
    checkpoints.save_checkpoint(tmp_path / "a.pth", 3, m, opt, sch, {"total": np.float32(2.0)}, val)
    checkpoints.save_best_map_model(tmp_path / "b.pth", 3, m, opt, val, 0.25)
    for f in ("a.pth", "b.pth"):
        ck = torch.load(tmp_path / f, map_location="cpu", weights_only=True)
        assert ck["epoch"] == 3 and ck["mAP50"] == 0.5 and type(ck["val_loss"]) is float


def test_north_star_functional_spellings():
    """cellboxes_to_boxes / non_max_suppression / mean_average_precision / YoloLoss: thin wrappers over the reference-named API."""
    import numpy as np
    import torch
    import yolo
    from yolo.inference import YOLOInference
    rng = np.random.default_rng(3)
    pred = torch.from_numpy(rng.uniform(0, 1, size=(2, 7, 7, 30)).astype(np.float32))
    boxes = yolo.cellboxes_to_boxes(pred, 0.3)
    inf = YOLOInference(yolo.YOLOv1(), device="cpu")
    dets = inf.parse_predictions(pred[0], 0.3)
    assert len(boxes) == 2 and len(boxes[0]) == len(dets) and boxes[0][0][1] == dets[0].confidence
    kept = yolo.non_max_suppression(boxes[0], 0.4)
    ref = inf.non_max_suppression(dets, nms_threshold=0.4)
    assert len(kept) == len(ref) and all(k[1] == d.confidence and int(k[0]) == d.class_id for k, d in zip(kept, ref))
    tgt = torch.zeros(2, 7, 7, 30)
    out = yolo.mean_average_precision(pred, tgt)
    assert "mAP50:95" in out and yolo.YoloLoss is yolo.YOLOLoss


def test_draw_detections_takes_the_reference_call():
    """src/predict.py:113 calls draw_detections(image, detections, class_names, conf_threshold) against
    src/yolo/utils/visualization.py:34-41: same positional meaning, a copy is returned, low-confidence boxes are skipped,
    legacy (class_id, conf, x, y, w, h) tuples are accepted."""
    import inspect
    from PIL import Image
    from yolo.schemas import BoundingBox, Detection
    from yolo.utils import VOC_CLASSES, draw_detections
    assert list(inspect.signature(draw_detections).parameters) == ["image", "detections", "class_names", "conf_threshold", "box_width", "font_size"]
    img = Image.new("RGB", (200, 100), "black")
    dets = [Detection(bbox=BoundingBox(x=0.5, y=0.5, width=0.4, height=0.4), confidence=0.9, class_id=11, class_name="dog"),
            Detection(bbox=BoundingBox(x=0.2, y=0.2, width=0.2, height=0.2), confidence=0.2, class_id=3, class_name="boat")]
    out = draw_detections(img, dets, VOC_CLASSES, 0.5)
    assert out is not img and out.size == img.size and img.getbbox() is None        # the input is untouched
    px = out.load()
    assert px[60, 50] != (0, 0, 0) and px[139, 50] != (0, 0, 0)                     # left / right edge of the 0.9 box
    assert px[30, 20] == (0, 0, 0)                                                  # the 0.2 box is below the threshold
    assert draw_detections(img, dets, VOC_CLASSES, 0.1).load()[20, 20] != (0, 0, 0)
    legacy = draw_detections(img, [(11, 0.9, 0.5, 0.5, 0.4, 0.4)], VOC_CLASSES)
    assert legacy.load()[60, 50] == px[60, 50]
    assert draw_detections(img, dets, box_width=1, font_size=10).size == img.size   # keyword form of the remaining parameters


def test_loss_parts_behaves_like_the_reference_dict():
    """yolo.loss.LossParts (the lazily fetched form of the reference's {"total": ..., ...} float dict, loss.py:165-169): every way of
    reading or copying it sees the values, the bad-slot flag raises at the first read (fake event + host buffer: no GPU needed)."""
    import copy
    import json
    import torch
    from yolo.loss import LossParts

    class Ev:
        def synchronize(self):
            pass

        def query(self):
            return True

    def mk(flag=0.0):
        return LossParts(Ev(), torch.tensor([5.0, 1.0, 2.0, 3.0, 4.0, flag, 0.0, 0.0]))

    want = {"total": 5.0, "coord": 1.0, "conf_obj": 2.0, "conf_noobj": 3.0, "class": 4.0}
    assert dict(mk()) == want and {**mk()} == want and list(mk().values()) == list(want.values()) and copy.deepcopy(mk()) == want
    assert json.loads(json.dumps(mk())) == want and mk()["coord"] == 1.0 and mk().get("class") == 4.0 and list(mk()) == list(want)
    assert len(mk()) == 5 and "total" in mk() and mk() == want and f"{mk()['total']:.1f}" == "5.0"
    d = {}
    d.update(mk())
    assert d == want and sum(mk().values()) == 15.0
    with pytest.raises(RuntimeError, match="index out of bounds"):
        mk(1.0)["total"]
    with pytest.raises(RuntimeError, match="index out of bounds"):
        dict(mk(1.0))


def test_bench_launches_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` (no RANK in the environment) must itself start N ranks under torch.distributed.run
    before touching a GPU, relay rank 0's JSON line, and fail when fewer than N devices are visible."""
    import subprocess
    import bench
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench._launch_ranks(8)
    assert e.value.code == 2
    seen = {}

    def fake_run(cmd, stdout=None, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 0, stdout=b"RCCL version 2.x banner\n{\"n_gpus\": 8}\n")
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench._launch_ranks(8)
    assert e.value.code == 0
    out = capsys.readouterr()
    assert out.out.strip() == '{"n_gpus": 8}' and "banner" in out.err
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_launch_with_batchnorm_statistics_gets_a_kernel_that_has_them(monkeypatch):
    """plans are keyed without yolo_igemm_desc.bn_stats, so a plan measured on the plain launch (inference, frozen statistics) also reaches the launch that
    accumulates BatchNorm sums; only the tiled kernels (tile_hint 1 .. 14) have that epilogue, and it must be ONE launch: every plan form of the
    pipelined / persistent kernels and every split falls back to its one-launch stand-in (found by tuning batch 32: ("slabs", 20, ..) -> tile_hint 20 raised)."""
    from yolo import plans
    from yolo._hip import IgemmDesc
    seen = []
    monkeypatch.setattr(plans, "_igemm", lambda L_, d, *a: seen.append((d.tile_hint, d.split_k, d.px_begin, d.px_end)))
    for plan in (("slabs", 20, 4, 196), ("slabs", 15, 4, 196), ("slabs", 4, 16, 0), ("splitk", 5, 3), ("tile", 20, 1, 196), ("tile", 21, 1, 224), ("tile", 15, 1, 196),
                 ("tile", 20, 1, 208, 3, 7000), ("skew", 15, 1, 3, 4000), (15, 1), (5, 1), (11, 1), (5, 1, 1024, 3)):
        d = IgemmDesc()
        d.N, d.Ho, d.Wo, d.KH, d.KW, d.tap_len, d.Cout, d.stride, d.split_k = 8, 14, 14, 3, 3, 512, 512, 1, 1
        d.bn_stats = 1234
        seen.clear()
        plans._run_plan_igemm(None, d, plan, None, None, None, None, None, None, "test")
        assert len(seen) == 1, (plan, seen)
        hint, split, b, e = seen[0]
        assert 1 <= hint <= 14 and split <= 1 and b == 0 and e == 0, (plan, seen)


def test_unmeasured_batch_sizes_borrow_the_nearest_measured_plan(monkeypatch):
    """plans._borrowed_plan: nearest measured batch size by ratio (the larger one on a tie), same layer key otherwise; pixel-range forms, whose cut is a
    pixel count, are not taken over; config.BORROW_PLANS / PLAN_TABLE switch it off; a plan the library refuses at the new size falls back to the rule."""
    from yolo import plans
    from yolo import _hip
    from yolo.config import CONFIG
    from yolo._hip import IgemmDesc
    rest = (14, 14, 3, 3, 512, 1024, 1, 2, 0, 1024, 512)
    table = {(8,) + rest: ("slabs", 4, 8, 0), (16,) + rest: ("tile", 20, 1, 196), (32,) + rest: (5, 1, 4096, 3), (64,) + rest: ("tile", 21, 1, 224),
             (16, 7, 7) + rest[2:]: (11, 1)}
    monkeypatch.setattr(plans, "_TUNED", dict(table))
    monkeypatch.setattr(plans, "_BORROWED", set())
    assert plans._borrowed_plan((13,) + rest) == ("tile", 20, 1, 196)          # 16 (x1.23) before 8 (x1.63)
    assert plans._borrowed_plan((11,) + rest) == ("slabs", 4, 8, 0)            # 8 (x1.375) before 16 (x1.45)
    assert plans._borrowed_plan((12,) + rest) == ("tile", 20, 1, 196)          # 16 (x1.33) before 8 (x1.5)
    assert plans._borrowed_plan((128,) + rest) == ("tile", 21, 1, 224)
    assert plans._borrowed_plan((6,) + rest) is None                            # below 8 images: the default rule sizes its K ranges by the pixel count
    assert plans._borrowed_plan((30,) + rest) is None                           # nearest is 32, a pixel-range form
    assert plans._borrowed_plan((13, 28, 28) + rest[2:]) is None                # no such layer at any batch size
    monkeypatch.setattr(CONFIG, "BORROW_PLANS", False)
    assert plans._borrowed_plan((13,) + rest) is None
    monkeypatch.setattr(CONFIG, "BORROW_PLANS", True)
    # igemm_call: the borrowed plan runs once under a guard; refused -> the default rule takes the key for good
    d = IgemmDesc()
    d.N, d.Ho, d.Wo, d.KH, d.KW, d.tap_len, d.Cout, d.stride, d.epilogue, d.pool2, d.out_px_stride, d.in_px_stride = (13,) + rest
    d.split_k = 1
    calls = []

    def fake(L_, dd, plan, *a):
        calls.append(plan)
        if plan == ("tile", 20, 1, 196):
            raise _hip.HipUnsupported("refused")

    monkeypatch.setattr(plans, "_run_plan_igemm", fake)
    monkeypatch.setattr(plans.RT, "lib", lambda: None)
    plans.igemm_call(d, None, None, None, None, None, None, "test")
    assert calls[0] == ("tile", 20, 1, 196) and len(calls) == 2 and calls[1] == plans._default_plan(d)
    assert plans._TUNED[(13,) + rest] == calls[1] and not plans._BORROWED
    plans.igemm_call(d, None, None, None, None, None, None, "test")
    assert len(calls) == 3 and calls[2] == calls[1]


def test_shipped_plan_table_and_weight_gradient_choices(tmp_path):
    """yolo/plans/gfx950.json: every entry parses into a plan the engine knows, the "wgrad" section (kernel choices measured inside the
    training step, tools/search_wgrad.py) reaches Plan._wgrad_desc, and save_plans / load_plans round-trip both sections."""
    import json
    from yolo import engine
    from yolo import plans as P
    from yolo.config import CONFIG
    from yolo.executor import Layer, Plan
    body = json.load(open(P.PLAN_FILE))
    assert body["arch"] == "gfx950" and len(body["plans"]) >= 200
    for k, v in body["plans"].items():
        key = tuple(int(t) for t in k.split(","))
        assert len(key) == 12 and key[0] in (1, 2, 4, 8, 16, 32, 64), k
        assert isinstance(v[0], int) or v[0] in ("tile", "slabs", "splitk", "skew"), (k, v)
    choice = CONFIG.WGRAD_CHOICE
    assert choice and all(len(k) == 7 and v[0] in (0, 5, 6) and v[1] in (0, 1) for k, v in choice.items())
    assert {",".join(map(str, k)): list(v) for k, v in choice.items()} == body["wgrad"]

    class _A:      # the geometry fields of an activation buffer that _wgrad_desc reads
        def __init__(self, H, W, C):
            self.H, self.W, self.C, self.halo = H, W, C, 1
            self.Hp, self.Wp = H + 2, W + 2
            self.px_stride, self.row_stride = C, (W + 2) * C
    (N, H, W, Cout, Cin, K, s), (variant, flat) = next(iter(choice.items()))
    L = Layer.__new__(Layer)
    L.Hout, L.Wout, L.Cout, L.Cin, L.K, L.stride, L.pad = H, W, Cout, Cin, K, s, 1
    wd = Plan._wgrad_desc(L, _A(H, W, Cout), _A(H, W, Cin), N)
    assert wd.variant == variant and (wd.geo_W == 0) == bool(flat)
    old = CONFIG.WGRAD_CHOICE
    try:
        CONFIG.WGRAD_CHOICE = None
        rule = Plan._wgrad_desc(L, _A(H, W, Cout), _A(H, W, Cin), N)
        assert (rule.variant, rule.geo_W == 0) != (variant, bool(flat)), "an override that equals the shape rule says nothing"
        CONFIG.WGRAD_CHOICE = old
        out = tmp_path / "t.json"
        P.save_plans(str(out), "round trip")
        again = json.load(open(out))
        assert again["plans"] == body["plans"] and again["wgrad"] == body["wgrad"]
    finally:
        CONFIG.WGRAD_CHOICE = old
