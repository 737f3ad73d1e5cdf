"""GPU, BASELINE.json's full sizes (batch 64, 448x448): size-independent properties where an element-wise
oracle would take minutes on the host."""

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from yolo import YOLOv1
    torch.manual_seed(3)
    return YOLOv1().cuda().eval()


def test_batch64_forward_is_batch_independent(model):
    """image i's prediction must not depend on which batch it rides in: batch-64 forward (tuned big-tile kernels,
    fused pools, split-K plans) vs the same images in batches of 8 and 1, to rounding noise; and bit-identical when the
    same batch is run twice."""
    x = torch.from_numpy(synth.synth_images(64, 17)).cuda()
    with torch.no_grad():
        y64 = model(x)
        y8 = torch.cat([model(x[i:i + 8]) for i in range(0, 64, 8)])
        y1 = model(x[37:38])
        f64 = model.backbone(x)
        f8 = model.backbone(x[8:16])
        f64b = model.backbone(x)
    assert y64.shape == (64, 7, 7, 30) and torch.isfinite(y64).all()
    # conv stack: every launch plan is deterministic, but plans differ with the batch size (large tiles, split-K in two
    # halves for the 7x7 layers at batch 64) and so does the fp32 summation order -> equal to bf16 rounding noise
    df = (f64[8:16] - f8).float()
    assert df.norm().item() <= 4e-3 * f8.float().norm().item() and df.abs().max().item() <= 0.05 * f8.abs().max().item()
    assert torch.equal(f64, f64b)                                  # same batch, same plans: bit-reproducible
    scale = y64.abs().mean().item()
    assert (y64 - y8).abs().max().item() < 5e-3 * scale + 1e-5
    assert (y64[37:38] - y1).abs().max().item() < 5e-3 * scale + 1e-5


def test_forward_is_deterministic_and_matches_cpu_on_a_sample(model):
    """two runs agree (up to FC atomics), and 2 of the 64 images checked against stock torch on the host"""
    import copy
    from test_gpu_layers import bf16_faithful
    x = torch.from_numpy(synth.synth_images(64, 18)).cuda()
    with torch.no_grad():
        a = model(x)
        b = model(x)
    assert (a - b).abs().max().item() < 1e-3 * a.abs().mean().item() + 1e-5
    ref = copy.deepcopy(model).cpu()
    net = torch.nn.Sequential(bf16_faithful(ref.backbone.features), bf16_faithful(ref.head))
    with torch.no_grad():
        yc = net(x[[5, 60]].cpu().to(torch.bfloat16).float()).view(-1, 7, 7, 30)
    rel = ((a[[5, 60]].cpu() - yc).norm() / yc.norm()).item()
    assert rel < 0.02, rel


def test_nms_properties_batch512():
    """idempotence, confidence order and class purity of NMS on 512 images (8-GPU global batch of config 4/5)"""
    from yolo import ops
    rng = np.random.Generator(np.random.PCG64([9, 99]))
    pred = torch.from_numpy(rng.uniform(0, 1, size=(512, 7, 7, 30)).astype(np.float32)).cuda()
    rec, cnt = ops.decode(pred, 0.3, 7, 2, 20)
    keep, kc = ops.nms(rec, cnt, 0.4, 0)
    rec_h, cnt_h, keep_h, kc_h = rec.cpu().numpy(), cnt.cpu().numpy(), keep.cpu().numpy(), kc.cpu().numpy()
    assert (kc_h <= cnt_h).all() and (kc_h > 0).all()
    # survivors, re-submitted, all survive again and keep their order (idempotence)
    rec2 = torch.zeros_like(rec)
    for n in range(512):
        k = keep_h[n, : kc_h[n]]
        conf = rec_h[n, k, 1]
        assert (np.diff(conf) <= 0).all()                         # inference variant: confidence order
        rec2[n, : kc_h[n]] = torch.from_numpy(rec_h[n, k]).cuda()
    keep2, kc2 = ops.nms(rec2, kc.clone(), 0.4, 0)
    assert torch.equal(kc2, kc)
    k2 = keep2.cpu().numpy()
    for n in range(0, 512, 37):
        assert (k2[n, : kc_h[n]] == np.arange(kc_h[n])).all()
    # metrics variant keeps the same SET, grouped by class
    keepm, kcm = ops.nms(rec, cnt, 0.4, 1)
    km, kcm_h = keepm.cpu().numpy(), kcm.cpu().numpy()
    for n in range(0, 512, 29):
        cls = rec_h[n, km[n, : kcm_h[n]], 0]
        changes = (np.diff(cls) != 0).sum()
        assert changes == len(np.unique(cls)) - 1                 # each class forms one contiguous group


def test_loss_is_mean_over_images():
    """YOLOLoss divides by N: the loss of a batch equals the mean of the per-image losses (N = 64)"""
    from yolo import YOLOLoss
    pred = torch.from_numpy(synth.synth_normal((64, 7, 7, 30), 77, 0.5) + 0.25).cuda()
    tgt = torch.from_numpy(synth.synth_targets(64, 31)).cuda()
    crit = YOLOLoss()
    total, _ = crit(pred, tgt)
    singles = torch.stack([crit(pred[i:i + 1], tgt[i:i + 1])[0] for i in range(64)])
    assert abs(total.item() - singles.mean().item()) < 1e-4 * max(1.0, abs(total.item()))
