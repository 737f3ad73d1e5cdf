"""yolo_preprocess_u8 (resize + ToTensor + Normalize on the device) vs the oracle restatement / Pillow: bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))

from oracle import preprocess_ref as R   # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(375, 500), (500, 333), (448, 448), (224, 224), (1000, 1500), (448, 300), (37, 53), (600, 448)])
def test_device_preprocess_is_bit_exact(hw):
    from yolo.preprocess import preprocess_u8
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    imgs = rng.integers(0, 256, size=(3,) + hw + (3,), dtype=np.uint8)
    out, act = preprocess_u8(torch.from_numpy(imgs).cuda(), (448, 448), nhwc4_halo=3)
    for i in range(3):
        pil = np.asarray(Image.fromarray(imgs[i]).resize((448, 448), Image.BILINEAR))
        ref = R.to_tensor_normalize(pil)
        assert np.array_equal(out[i].cpu().numpy(), ref), f"image {i}: fp32 NCHW differs from Pillow + torch arithmetic"
    # stem-ready buffer == what the engine derives from the fp32 tensor (bf16 rounding, channel 3 and halo zero)
    v = act.view().float().cpu()
    inner = v[:, 3:-3, 3:-3, :]
    assert torch.equal(inner[..., :3].permute(0, 3, 1, 2), out.to(torch.bfloat16).float().cpu())
    assert inner[..., 3].abs().sum() == 0 and v[:, :3].abs().sum() == 0 and v[:, :, -3:].abs().sum() == 0


def test_inference_preprocess_image_uses_the_device_path_and_matches_host():
    from yolo import YOLOv1
    from yolo.inference import YOLOInference, _Preprocess
    rng = np.random.default_rng(5)
    img = Image.fromarray(rng.integers(0, 256, size=(333, 500, 3), dtype=np.uint8))
    inf = YOLOInference(YOLOv1(), device="cuda")
    got = inf.preprocess_image(img)
    assert got.is_cuda and got.shape == (1, 3, 448, 448)
    assert torch.equal(got.cpu()[0], _Preprocess()(img))
    inf.model.cpu()


def test_forward_uint8_equals_transform_then_forward():
    from yolo import YOLOv1
    from yolo.preprocess import preprocess_u8
    torch.manual_seed(0)
    m = YOLOv1().cuda().eval()
    rng = np.random.default_rng(9)
    u8 = torch.from_numpy(rng.integers(0, 256, size=(2, 375, 500, 3), dtype=np.uint8)).cuda()
    x, _ = preprocess_u8(u8, (448, 448))
    with torch.no_grad():
        ref = m(x)
    got = m.forward_uint8(u8)
    # same bf16 stem input either way; the split-K Linear adds fp32-atomic ordering noise between any two forward passes
    assert got.shape == (2, 7, 7, 30)
    torch.testing.assert_close(got, ref, rtol=0, atol=1e-3 * ref.abs().mean().item())
    m.cpu()
