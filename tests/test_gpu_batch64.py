"""The training backward at the BENCHMARKED size (BASELINE configs[2]: batch 64), which exercises code paths the small-batch
tests never reach: yolo_wgrad's two-segment schedule with split > 1 atomics at 64x the pixels, the shipped launch plans of
the data-gradient (256 x 208 tiles of 196 pixels, pipelined loop, slabs, pixel ranges) under EPI_MUL_DLRELU.

  * weight gradients: LINEARITY -- the batch-64 gradient of a layer equals the fixed-order sum of eight batch-8 launches of
    yolo_wgrad on the very activations / output gradients the batch-64 pass stored (fp32 round-off; the batch-8 launches are
    the configuration the teacher-forced tests of test_gpu_model.py pin to stock torch);
  * data gradients: teacher-forced on 2 of the 64 images against stock torch (conv2d_input on the stored output gradient);
  * FC1's weight gradient against a float64 product on the host.
Reference lines: src/yolo/models.py:47-84,239-245 (layers), src/yolo/loss.py:87-172, src/yolo/training/trainer.py:61-95."""

import ctypes

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def step64():
    """one batch-64 forward + loss + backward of YOLOv1 (eval mode: dropout off) with the workspace kept"""
    from yolo import YOLOLoss, YOLOv1
    torch.manual_seed(0)
    m = YOLOv1()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.yolov1_state_dict().items()}, strict=True)
    m = m.cuda().eval()
    plan = m.hip_plan()
    plan.debug_keep = True
    N = 64
    x = torch.from_numpy(synth.synth_images(N, 11)).cuda()
    t = torch.from_numpy(synth.synth_targets(N, 31, max_obj=4)).cuda()
    pred = m(x)
    seen = []
    pred.register_hook(lambda g: seen.append(g.detach().clone()))      # dL/dpred as it enters the network's backward
    loss, _ = YOLOLoss()(pred, t)
    loss.backward()
    torch.cuda.synchronize()
    ws, fc_saved = plan.last
    plan.debug_keep = False
    yield {"model": m, "plan": plan, "ws": ws, "fc": fc_saved, "N": N, "gpred": seen[0]}
    plan.last = None


def _conv_layers(plan):
    return [(li, L) for li, L in enumerate(plan.layers) if L.kind == "conv"]


# representative layers: conv2 (112x112, 64 -> 192, paired taps), conv7 (56x56, 256 -> 512), a 1x1 (28x28, 512 -> 512), conv18
# (28x28, 512 -> 1024), the stride-2 layer (14x14 -> 7x7, zero-stuffed gradient), a 7x7x1024 layer -- by position among the 24 convs
_PICK = (1, 5, 14, 15, 21, 23)


def test_weight_gradients_are_linear_in_the_batch(step64):
    from yolo._hip import check, lib, ptr, stream
    plan, ws, N = step64["plan"], step64["ws"], step64["N"]
    convs = _conv_layers(plan)
    plan._apply_geom(ws)
    for k in _PICK:
        li, L = convs[k]
        g, xin = ws["grads"][li], ws["acts"][li - 1]
        got_w = L.weight.grad.permute(0, 2, 3, 1).contiguous()          # OIHW -> the packed [Cout][KH][KW][Cin]
        got_b = L.bias.grad
        acc_w = torch.zeros_like(got_w, dtype=torch.float64)
        acc_b = torch.zeros_like(got_b, dtype=torch.float64)
        sub = 8
        wd = plan._wgrad_desc(L, g, xin, sub)
        for n0 in range(0, N, sub):
            dwp = torch.zeros(L.Cout * L.K * L.K * L.Cin, dtype=torch.float32, device="cuda")
            db = torch.zeros(L.Cout, dtype=torch.float32, device="cuda")
            xp = ctypes.c_void_p(xin.t.data_ptr() + n0 * xin.img_stride * 2)
            gp = ctypes.c_void_p(g.t.data_ptr() + n0 * g.img_stride * 2)
            check(lib().yolo_wgrad(ctypes.byref(wd), xp, gp, ptr(dwp), ptr(db), stream()), f"wgrad conv{li} images {n0}..")
            acc_w += dwp.view_as(got_w).double()
            acc_b += db.double()
        for got, ref, what in ((got_w, acc_w, "weight"), (got_b, acc_b, "bias")):
            err = (got.double() - ref).abs().max().item()
            scale = ref.abs().max().item()
            # fp32 accumulation of ~1e5 .. 8e5 products per element in different orders: a few 1e-6 relative to the largest element
            assert err <= 2e-5 * scale + 1e-7, f"conv{li} {what} gradient: max |batch64 - sum of 8 x batch8| = {err:.3e} (scale {scale:.3e})"


def test_data_gradients_of_two_of_the_64_images_vs_stock_torch(step64):
    from torch.nn.grad import conv2d_input
    from test_gpu_layers import _bf, _close
    plan, ws = step64["plan"], step64["ws"]
    convs = _conv_layers(plan)
    plan._apply_geom(ws)
    imgs = [5, 63]

    def nchw(act, stuffed=False):
        v = act.interior()[imgs].float().cpu()
        if stuffed:
            v = v[:, 0::2, 0::2, :]
        return v.permute(0, 3, 1, 2).contiguous()

    for k in _PICK:
        li, L = convs[k]
        prev = plan.layers[li - 1]
        w = _bf(L.weight.detach().float().cpu())
        dz = nchw(ws["grads"][li], stuffed=(L.stride == 2))[:, :, : L.Hout, : L.Wout]
        dx_ref = conv2d_input((len(imgs), L.Cin, L.Hin, L.Win), w, dz, stride=L.stride, padding=L.pad)
        if prev.kind == "conv":
            yprev = nchw(ws["acts"][li - 1])
            dx_ref = dx_ref * torch.where(yprev > 0, 1.0, 0.1)
            got = nchw(ws["grads"][li - 1], stuffed=(prev.stride == 2 and not prev.first))[:, :, : prev.Hout, : prev.Wout]
        else:
            got = nchw(ws["misc"][("gpool", li)])
        _close(got, _bf(dx_ref), 3.0, f"conv{li} data gradient (images {imgs} of 64)")


def test_fc1_weight_gradient_at_batch_64(step64):
    """FC1 (50176 -> 4096, 822 MB of gradient) at batch 64: the split schedule of yolo_wgrad on the Linear layer, against a
    float64 product on the host built from what the GPU pass stored (its input and hidden activation) and the loss gradient."""
    from test_gpu_layers import _bf, _close
    plan, fc, N = step64["plan"], step64["fc"], step64["N"]
    fcs = [(li, L) for li, L in enumerate(plan.layers) if L.kind == "fc"]
    (l1, L1), (l2, L2) = fcs
    xin, y1, mask = fc[l1]
    assert mask is None
    gb2 = _bf(step64["gpred"].reshape(N, -1).float().cpu())                 # the gradient as the kernels store it
    gprev = gb2 @ _bf(L2.weight.detach().float().cpu())                       # data gradient of FC2, fp32
    gb1 = _bf(gprev * torch.where(y1.float().cpu() > 0, 1.0, 0.1))
    ref = (gb1.double().t() @ xin.float().cpu().double()).float()
    _close(L1.weight.grad, ref, 3.0, "FC1 weight gradient at batch 64", frac=0.002)
    _close(L1.bias.grad, gb1.double().sum(0).float(), 3.0, "FC1 bias gradient at batch 64", frac=0.002)
    ref2 = (gb2.double().t() @ y1.float().cpu().double()).float()
    _close(L2.weight.grad, ref2, 3.0, "FC2 weight gradient at batch 64")
