"""GPU parity: fused HIP loss forward+backward (C ABI) vs reference fixtures and the oracle.
Bar (north_star): <= 1e-4 on the fp32 loss; gradients to fp32 round-off."""

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from yolo import ops as _ops
    return _ops


def test_golden_loss_cases(ops, golden):
    g = golden("loss_cases.npz")
    for name in [str(n) for n in g["names"]]:
        lc, ln = (float(v) for v in g[f"{name}__lambdas"])
        out, dpred = ops.loss_fwd_bwd(torch.from_numpy(g[f"{name}__pred"]).cuda(), torch.from_numpy(g[f"{name}__tgt"]).cuda(), 7, 2, 20, lc, ln)
        out = out.cpu().numpy()
        ref5 = g[f"{name}__out5"]
        assert out[5] == 0.0
        assert np.max(np.abs(out[:5] - ref5)) <= 1e-4 * max(1.0, np.max(np.abs(ref5))), (name, out[:5], ref5)
        np.testing.assert_allclose(out[:5], ref5, rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(dpred.cpu().numpy(), g[f"{name}__dpred"], rtol=2e-5, atol=2e-6, err_msg=name)


def test_batch64_and_512_vs_oracle(ops):
    import synth
    for N, seed in ((64, 3), (512, 4)):
        pred = synth.synth_normal((N, 7, 7, 30), 900 + seed, 0.5, seed) + 0.25
        tgt = synth.synth_targets(N, seed)
        out, dpred = ops.loss_fwd_bwd(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), 7, 2, 20, 5.0, 0.5)
        ref5, refg = O.loss_fwd_bwd(pred, tgt)
        np.testing.assert_allclose(out.cpu().numpy()[:5], ref5, rtol=1e-5, atol=1e-6)
        # same fp32 per-cell arithmetic on both sides -> gradients agree to the last bits
        np.testing.assert_allclose(dpred.cpu().numpy(), refg, rtol=1e-6, atol=1e-8)


def test_error_flag_for_bad_target_slot(ops):
    t = np.zeros((2, 7, 7, 30), np.float32)
    t[1, 1, 1, 14] = 1.0
    out, _ = ops.loss_fwd_bwd(torch.zeros((2, 7, 7, 30)).cuda(), torch.from_numpy(t).cuda(), 7, 2, 20, 5.0, 0.5)
    assert out.cpu().numpy()[5] == 1.0


def test_loss_iou(ops, golden):
    g = golden("loss_cases.npz")
    out = ops.loss_iou(torch.from_numpy(g["iou__b1"]).cuda(), torch.from_numpy(g["iou__b2"]).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), g["iou__out"], rtol=1e-6, atol=1e-7)


def test_other_shapes(ops):
    """B=3, C=7, S=5: the 4::5 slice then covers channels 4,9,14,19."""
    rng = np.random.Generator(np.random.PCG64([5, 98]))
    S, B, C = 5, 3, 7
    D = B * 5 + C
    pred = rng.standard_normal(size=(6, S, S, D)).astype(np.float32) * 0.5 + 0.3
    tgt = np.zeros((6, S, S, D), np.float32)
    for n in range(6):
        for _ in range(4):
            i, j, slot = rng.integers(0, S), rng.integers(0, S), rng.integers(0, B)
            tgt[n, i, j, slot * 5: slot * 5 + 5] = [*rng.uniform(0, 1, 2), *rng.uniform(0.1, 0.9, 2), 1.0]
            tgt[n, i, j, B * 5 + rng.integers(0, C)] = 1.0
    out, dpred = ops.loss_fwd_bwd(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), S, B, C, 5.0, 0.5)
    ref5, refg = O.loss_fwd_bwd(pred, tgt, S, B, C)
    np.testing.assert_allclose(out.cpu().numpy()[:5], ref5, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dpred.cpu().numpy(), refg, rtol=1e-6, atol=1e-8)


def test_loss_components_are_fetched_when_read():
    """YOLOLoss on device tensors returns LossParts: the reference's dict of five floats (loss.py:165-169), copied to the host by a side
    stream and awaited at the first read -- equal to the eager copy; the reference's IndexError case (a target selecting a box slot
    >= B) surfaces at that read, or at the next loss call when the dict is never read."""
    from yolo import YOLOLoss
    from yolo.loss import LossParts
    rng = np.random.Generator(np.random.PCG64([7, 7]))
    pred = torch.from_numpy(rng.normal(size=(4, 7, 7, 30)).astype(np.float32)).cuda().requires_grad_(True)
    tgt = np.zeros((4, 7, 7, 30), np.float32)
    tgt[0, 2, 3, 0:5] = [0.5, 0.5, 0.3, 0.4, 1.0]
    tgt[0, 2, 3, 10 + 4] = 1.0
    tgt = torch.from_numpy(tgt).cuda()
    crit = YOLOLoss()
    total, parts = crit(pred, tgt)
    assert isinstance(parts, LossParts) and isinstance(parts, dict) and list(parts) == ["total", "coord", "conf_obj", "conf_noobj", "class"]
    crit.eager_parts = True
    total_e, eager = crit(pred, tgt)
    _, again = crit.__class__()(pred, tgt)                # a fresh, unread LossParts: every way of copying it must see the values
    assert dict(again) == eager and {**crit.__class__()(pred, tgt)[1]} == eager and list(crit.__class__()(pred, tgt)[1].values()) == list(eager.values())
    import copy, json
    assert copy.deepcopy(crit.__class__()(pred, tgt)[1]) == eager and json.loads(json.dumps(crit.__class__()(pred, tgt)[1])) == eager
    assert type(eager) is dict and parts == eager and dict(parts.items()) == eager and parts["total"] == eager["total"]
    assert abs(parts["total"] - float(total.detach())) <= 1e-6 * abs(float(total.detach())) and "coord" in repr(parts)
    total.backward()                                     # the autograd side is unchanged
    assert pred.grad is not None
    # a target that selects slot 2 (channel 14 of the 4::5 slice, B = 2)
    bad = torch.zeros((2, 7, 7, 30)).cuda()
    bad[1, 1, 1, 14] = 1.0
    crit.eager_parts = False
    _, p1 = crit(torch.zeros((2, 7, 7, 30)).cuda(), bad)
    with pytest.raises(RuntimeError, match="index out of bounds"):
        p1["total"]
    _, p2 = crit(torch.zeros((2, 7, 7, 30)).cuda(), bad)           # never read ...
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="index out of bounds"):
        crit(pred, tgt)                                             # ... so the next call reports it
    crit.eager_parts = True
    with pytest.raises(RuntimeError, match="index out of bounds"):
        crit(torch.zeros((2, 7, 7, 30)).cuda(), bad)
