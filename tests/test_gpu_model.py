"""GPU parity of the whole YOLOv1 (24-conv backbone + FC head) forward and training step.

Oracle: (a) the reference model's own output on a fixed image with fixed weights
(tests/golden/backbone_full.npz, produced by running mattiaskvist/yolo-v1's YOLOv1 on the CPU);
(b) stock torch.nn on the host CPU with the same parameters (the reference's arithmetic for these
layers).  bf16 storage / fp32 accumulate through 26 layers: tolerance 3% of the output scale
(measured error is ~0.5%).
"""

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from yolo import YOLOv1
    m = YOLOv1()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.yolov1_state_dict().items()}, strict=True)
    return m


def _rel(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def test_forward_matches_reference_fixture(model, golden):
    g = golden("backbone_full.npz")
    m = model.cuda().eval()
    with torch.no_grad():
        y = m(torch.from_numpy(synth.synth_images(1, 0)).cuda())
        y2 = m(torch.from_numpy(synth.synth_images(2, 7)).cuda())
        feat = m.backbone(torch.from_numpy(synth.synth_images(1, 0)).cuda())
    assert y.shape == (1, 7, 7, 30) and feat.shape == (1, 1024, 7, 7)
    for got, ref, what in ((y, g["y"], "y"), (y2, g["y2"], "y2"), (feat, g["feat"], "feat")):
        r = _rel(got, torch.from_numpy(ref))
        assert r < 0.03, (what, r)
        assert (got.cpu() - torch.from_numpy(ref)).abs().max() < 0.15 * np.abs(ref).max(), what
    model.cpu()


def test_train_step_gradients_vs_cpu_autograd(model):
    """loss + backward on 2 images: every parameter gradient vs torch CPU autograd (fp32)."""
    import copy
    from yolo import YOLOLoss
    N = 2
    x = torch.from_numpy(synth.synth_images(N, 3))
    t = torch.from_numpy(synth.synth_targets(N, 21, max_obj=4))
    ref = copy.deepcopy(model).cpu().eval()     # eval: dropout off on both sides
    crit = YOLOLoss()
    lr, _ = crit(ref(x), t)
    lr.backward()
    m = model.cuda().eval()
    m.zero_grad()
    lg, dg = crit(m(x.cuda()), t.cuda())
    lg.backward()
    assert abs(lg.item() - lr.item()) < 0.05 * abs(lr.item())
    worst = 0.0
    for (n1, p1), (_, p2) in zip(m.named_parameters(), ref.named_parameters()):
        assert p1.grad is not None, n1
        g1, g2 = p1.grad.float().cpu().flatten(), p2.grad.flatten()
        cos = torch.dot(g1, g2) / (g1.norm() * g2.norm() + 1e-30)
        rel = (g1 - g2).norm() / (g2.norm() + 1e-30)
        worst = max(worst, rel.item())
        assert cos > 0.995 and rel < 0.1, (n1, cos.item(), rel.item())
    model.cpu()
    print("worst relative gradient error", worst)


def test_loss_decreases_with_adam(model):
    """the drop-in train step (zero_grad / forward / loss / backward / clip / Adam) runs and learns."""
    import copy
    from yolo import YOLOLoss
    m = copy.deepcopy(model).cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=5e-4)
    crit = YOLOLoss()
    x = torch.from_numpy(synth.synth_images(4, 5)).cuda()
    t = torch.from_numpy(synth.synth_targets(4, 22)).cuda()
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss, d = crit(m(x), t)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=10.0)
        opt.step()
        losses.append(d["total"])
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
