"""End-to-end check of the command-line surface on the GPU: train.py (two epochs, mAP validation, checkpoints), --resume,
evaluate.py on the best checkpoint, predict.py on an image -- the same scripts and checkpoint keys a user of the
reference's src/train.py / src/evaluate.py / src/predict.py works with.  Synthetic images only (no dataset on the box);
each script runs in its own process, as a user would start it."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "yolo-v1_amd")

pytestmark = pytest.mark.gpu


def _run(args, cwd):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{args}\n--- stdout\n{r.stdout[-2000:]}\n--- stderr\n{r.stderr[-4000:]}"
    return r.stdout


def test_train_resume_evaluate_predict(tmp_path):
    from PIL import Image
    import numpy as np

    ck = tmp_path / "ck"
    common = ["--backbone", "yolov1", "--batch-size", "8", "--num-workers", "0", "--synthetic", "16", "--checkpoint-dir", str(ck)]
    out = _run([os.path.join(PKG, "train.py"), "--epochs", "2", "--save-frequency", "1", "--compute-map"] + common, str(tmp_path))
    assert "done:" in out
    for name in ("yolo_latest.pth", "yolo_epoch_1.pth", "yolo_epoch_2.pth", "yolo_best.pth"):
        assert (ck / name).exists(), name
    st = torch.load(ck / "yolo_latest.pth", map_location="cpu", weights_only=True)
    assert st["epoch"] == 2 and {"model_state_dict", "optimizer_state_dict", "train_loss", "val_loss"} <= set(st)
    assert all(torch.isfinite(v).all() for v in st["model_state_dict"].values())

    out = _run([os.path.join(PKG, "train.py"), "--epochs", "3", "--resume", str(ck / "yolo_latest.pth")] + common, str(tmp_path))
    assert "done:" in out
    st3 = torch.load(ck / "yolo_latest.pth", map_location="cpu", weights_only=True)
    assert st3["epoch"] == 3
    w2, w3 = st["model_state_dict"], st3["model_state_dict"]
    assert any(not torch.equal(w2[k], w3[k]) for k in w2)          # the resumed epoch trained

    res = tmp_path / "eval.txt"
    out = _run([os.path.join(PKG, "evaluate.py"), "--checkpoint", str(ck / "yolo_best.pth"), "--backbone", "yolov1", "--synthetic", "8",
                "--batch-size", "8", "--output", str(res)], str(tmp_path))
    keys = dict(line.split(": ") for line in res.read_text().strip().splitlines())
    assert "mAP50" in keys and "mAP50:95" in keys
    assert 0.0 <= float(keys["mAP50"]) <= 1.0

    img = tmp_path / "img.jpg"
    Image.fromarray(np.random.default_rng(0).integers(0, 255, (375, 500, 3), dtype=np.uint8)).save(img)
    outdir = tmp_path / "pred"
    _run([os.path.join(PKG, "predict.py"), str(img), "--checkpoint", str(ck / "yolo_best.pth"), "--backbone", "yolov1", "--output-dir", str(outdir)], str(tmp_path))
    assert (outdir / "img.jpg").exists()     # the annotated copy (schemas validate every kept box, as the reference's do)


def test_default_training_model_resnet50_unfrozen(tmp_path):
    """the reference's default run: ResNet-50 backbone, NOT frozen (src/train.py:144, freeze_backbone=False), DetectionHead;
    random initialisation because no ImageNet weights can be downloaded here"""
    ck = tmp_path / "ck"
    out = _run([os.path.join(PKG, "train.py"), "--epochs", "1", "--batch-size", "4", "--num-workers", "0", "--synthetic", "8", "--no-pretrained",
                "--checkpoint-dir", str(ck)], str(tmp_path))
    assert "done:" in out
    st = torch.load(ck / "yolo_latest.pth", map_location="cpu", weights_only=True)
    sd = st["model_state_dict"]
    assert "backbone.extractor.0.weight" in sd and "backbone.extractor.7.2.bn3.running_var" in sd
    assert all(torch.isfinite(v).all() for v in sd.values())
    assert int(sd["backbone.extractor.1.num_batches_tracked"]) == 2          # two optimizer steps updated the BatchNorm statistics
