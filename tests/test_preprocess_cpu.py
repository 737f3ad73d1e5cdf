"""Preprocessing without a GPU: the oracle's NumPy restatement of Pillow's 8-bit BILINEAR resampling and the
product's coefficient tables, both pinned bit for bit against PIL.Image.resize itself; ToTensor + Normalize against torch."""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))

from oracle import preprocess_ref as R   # noqa: E402

SIZES = [(375, 500), (500, 333), (448, 448), (224, 224), (1000, 1500), (448, 300), (37, 53), (600, 448), (449, 447)]


def test_resize_restatement_equals_pillow():
    rng = np.random.default_rng(0)
    for h, w in SIZES:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((448, 448), Image.BILINEAR))
        assert np.array_equal(R.resize_bilinear_u8(img, 448, 448), ref), (h, w)
    img = rng.integers(0, 256, size=(300, 200, 3), dtype=np.uint8)       # non-square target, up- and down-scaling mixed
    assert np.array_equal(R.resize_bilinear_u8(img, 120, 640), np.asarray(Image.fromarray(img).resize((640, 120), Image.BILINEAR)))


def test_product_tables_equal_the_oracle_tables():
    from yolo.preprocess import bilinear_tables
    for a, b in [(500, 448), (333, 448), (1500, 448), (448, 448), (53, 448), (224, 448), (449, 448), (447, 448)]:
        bo, co = R.bilinear_coeffs(a, b)
        bp, cp, k = bilinear_tables(a, b)
        assert np.array_equal(bo, bp) and np.array_equal(co, cp) and cp.shape[1] == k, (a, b)
        assert (cp.sum(axis=1) - (1 << 22)).__abs__().max() <= k                 # weights sum to 1.0 up to rounding


def test_to_tensor_normalize_equals_torch():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(64, 48, 3), dtype=np.uint8)
    t = torch.from_numpy(img.copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
    ref = (t - torch.tensor(R.MEAN).view(3, 1, 1)) / torch.tensor(R.STD).view(3, 1, 1)
    assert np.array_equal(R.to_tensor_normalize(img), ref.numpy())


def test_host_transform_equals_the_restatement():
    """yolo.inference's host transform (what the reference's torchvision transform computes) == oracle restatement."""
    from yolo.inference import _Preprocess
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(375, 500, 3), dtype=np.uint8)
    got = _Preprocess()(Image.fromarray(img)).numpy()
    assert np.array_equal(got, R.preprocess(img))
