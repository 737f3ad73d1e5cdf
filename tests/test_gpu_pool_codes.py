"""Training keeps, of a conv -> LeakyReLU -> MaxPool2d(2,2) group (src/yolo/models.py:49-55), only the pooled map and a 2-bit
arg-max code per pooled element (yolo_igemm pool2 = 3, yolo_conv_stem7_fwd pool2 = 3); the backward pass rebuilds the pool's and
the LeakyReLU's gradient from those (yolo_maxpool2_bwd_codes, yolo_wgrad_stem7_codes).  That path must equal the one that stores
the un-pooled activation (which the teacher-forced tests of test_gpu_model.py pin to stock torch): same forward bit for bit,
codes = torch's max_pool2d indices, same data gradients bit for bit, parameter gradients equal up to the order of the weight-
gradient kernels' fp32 atomics."""

import ctypes

import pytest
import torch
import torch.nn.functional as F

import synth

pytestmark = pytest.mark.gpu


def _unpack_codes(codes: torch.Tensor, act) -> torch.Tensor:
    """uint16 codes indexed like the pooled map / 8 -> [N][H][W][C] window positions (0..3) of the interior"""
    N, Hp, Wp, C = act.N, act.Hp, act.Wp, act.C
    c = codes.view(torch.int16).to(torch.int32).cpu() & 0xFFFF
    c = c.view(N, Hp, Wp, C // 8)
    pos = torch.stack([(c >> (2 * k)) & 3 for k in range(8)], dim=-1).reshape(N, Hp, Wp, C)
    h = act.halo
    return pos[:, h: h + act.H, h: h + act.W, :]


def _run(m, x, t, keep):
    from yolo import YOLOLoss
    plan = m.hip_plan()
    plan.debug_keep = keep
    for p in m.parameters():
        p.grad = None
    pred = m(x)
    loss, _ = YOLOLoss()(pred, t)
    loss.backward()
    torch.cuda.synchronize()
    ws, _ = plan.last
    plan.debug_keep = False
    plan.last = None
    return pred.detach().clone(), [p.grad.detach().clone() for p in m.parameters()], ws


def test_training_step_with_argmax_codes_equals_the_stored_activation_path():
    from yolo import YOLOv1
    torch.manual_seed(0)
    m = YOLOv1()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.yolov1_state_dict().items()}, strict=True)
    m = m.cuda().eval()                       # eval: dropout off; the plan still runs its training forward (gradients are required)
    N = 4
    x = torch.from_numpy(synth.synth_images(N, 21)).cuda()
    t = torch.from_numpy(synth.synth_targets(N, 22, max_obj=3)).cuda()
    plan = m.hip_plan()

    pred_a, grads_a, ws = _run(m, x, t, True)                 # un-pooled activations stored
    pools = [li for li, L in enumerate(plan.layers) if L.kind == "pool"]
    full = {li: ws["acts"][li - 1].interior().float().cpu() for li in pools}                    # [N][H][W][C]
    dz_a = {li: ws["grads"][li - 1].t.clone() for li in pools if (li - 1) in ws["grads"]}      # gradient at the conv output (pool + LeakyReLU backward)
    assert not ws["codes"]

    pred_b, grads_b, ws = _run(m, x, t, "codes")              # the product path
    assert sorted(ws["codes"]) == [li - 1 for li in pools], "every conv + pool group of YOLOv1 takes the codes path"
    assert torch.equal(pred_a, pred_b), "forward differs between the two paths"
    for li in pools:
        pooled = ws["acts"][li]
        pos = _unpack_codes(ws["misc"][("codes", li - 1)], pooled)
        y = full[li].permute(0, 3, 1, 2)
        _, idx = F.max_pool2d(y, 2, 2, return_indices=True)                                     # flat index into the H x W plane
        W = y.shape[-1]
        want = ((idx // W) % 2) * 2 + (idx % W) % 2
        assert torch.equal(pos.permute(0, 3, 1, 2).long(), want), f"pool {li}: arg-max codes differ from torch.max_pool2d's indices"
        if (li - 1) in dz_a:
            assert torch.equal(dz_a[li], ws["grads"][li - 1].t), f"pool {li}: gradient at the conv output differs"
    for ga, gb, (name, _) in zip(grads_a, grads_b, m.named_parameters()):
        scale = ga.abs().max().item()
        err = (ga - gb).abs().max().item()
        assert err <= 2e-5 * scale + 1e-9, f"{name}: gradients differ between the paths: {err:.3e} (scale {scale:.3e})"


def test_pool_backward_from_codes_with_ties():
    """yolo_maxpool2_bwd_codes == yolo_maxpool2_bwd_lrelu on activations full of ties (values from a set of five)"""
    from yolo._hip import PoolDesc, check, lib, ptr, stream
    from yolo.engine import Act
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 3, 12, 20, 24
    y = Act(N, H, W, C, 1, dev)
    vals = torch.tensor([-1.0, -0.25, 0.0, 0.5, 2.0])
    yi = vals[torch.randint(0, 5, (N, H, W, C), generator=g)].to(torch.bfloat16)
    y.interior().copy_(yi.cuda())
    dp = Act(N, H // 2, W // 2, C, 1, dev)
    dp.interior().copy_(torch.randn((N, H // 2, W // 2, C), generator=g).to(torch.bfloat16).cuda())
    ref, got = Act(N, H, W, C, 1, dev), Act(N, H, W, C, 1, dev)
    pd = PoolDesc(N, H, W, C, 1, 1)
    check(lib().yolo_maxpool2_bwd_lrelu(ctypes.byref(pd), y.p, dp.p, 0.1, ref.p, stream()), "bwd")
    # pooled activation + codes as a fused epilogue would leave them (first maximum in window order)
    yp = Act(N, H // 2, W // 2, C, 1, dev)
    win = yi.float().view(N, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(N, H // 2, W // 2, C, 4)
    mx, am = win.max(dim=-1)
    first = (win == mx.unsqueeze(-1)).float().argmax(dim=-1)             # first position holding the maximum
    yp.interior().copy_(mx.to(torch.bfloat16).cuda())
    code = torch.zeros((N, yp.Hp, yp.Wp, C // 8), dtype=torch.int32)
    f8 = first.view(N, H // 2, W // 2, C // 8, 8)
    packed = sum(f8[..., k].int() << (2 * k) for k in range(8))
    code[:, 1: 1 + H // 2, 1: 1 + W // 2, :] = packed
    codes = code.to(torch.int16).reshape(-1).cuda()          # values < 2^16: the int16 cast keeps the bit pattern
    check(lib().yolo_maxpool2_bwd_codes(ctypes.byref(pd), yp.p, ptr(codes), dp.p, 0.1, got.p, stream()), "bwd codes")
    torch.cuda.synchronize()
    assert torch.equal(ref.t, got.t)
