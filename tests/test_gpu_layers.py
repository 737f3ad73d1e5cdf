"""GPU parity of the MFMA layer kernels (yolo_igemm / yolo_wgrad / pool / layout), driven through the
same engine.Plan the models use, against (a) fixtures produced with stock torch ops (the reference's
arithmetic for these layers, tests/golden/layers_small.npz) and (b) torch CPU autograd on the same
modules.

Tolerance: operands and stored activations are bf16 (8 significant bits), accumulation is fp32.
Against an fp32 reference evaluated on the SAME bf16-rounded operands only the output rounding and
summation order remain: |err| <= 2^-7 * |ref| + 2^-8 * rms(ref) (one bf16 ulp).  Gradients pass
through one or two further bf16 roundings -> 3 * that bound.
"""

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -8


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _close(got, ref, k=1.0, what="", frac=0.0):
    """elementwise |err| <= k*(2^-7*|ref| + 2^-8*rms(ref)) -- one bf16 ulp is up to 2^-7 relative --
    `frac` = share of elements allowed outside.
    (Through a LeakyReLU a bf16 rounding can flip the gate of a pre-activation that is ~0, which
    changes that unit's gradient contribution x10: gradient checks of multi-layer chains allow a
    small share of such outliers and bound the global relative L2 error instead.)"""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    rms = ref.pow(2).mean().sqrt().item()
    bound = k * EPS * (2 * ref.abs() + rms) + 1e-6
    bad = (got - ref).abs() > bound
    rel = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    assert bad.float().mean().item() <= frac and rel < max(0.02, 4 * k * EPS), \
        f"{what}: {int(bad.sum())}/{bad.numel()} outside tolerance, max err {(got - ref).abs().max().item():.4g}, rms {rms:.4g}, rel L2 {rel:.4g}"


class Q(nn.Module):
    """straight-through bf16 rounding: what the GPU path does when it STORES an activation"""

    def forward(self, x):
        return x + (_bf(x) - x).detach()


def bf16_faithful(mods):
    """stock-torch CPU reference that mirrors the GPU's storage precision: bf16-rounded weights and a
    rounding of every stored activation (after conv+LeakyReLU / hidden Linear+LeakyReLU), so that
    LeakyReLU gates and max-pool arg-maxes are decided on the same values on both sides."""
    mods = list(mods)
    out = []
    for i, m in enumerate(mods):
        out.append(m)
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(m, nn.LeakyReLU) or (isinstance(m, nn.Conv2d) and not isinstance(nxt, nn.LeakyReLU)):
            out.append(Q())
    with torch.no_grad():
        for m in mods:
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                m.weight.copy_(_bf(m.weight))
    return nn.Sequential(*out)


def _run_both(mods_cpu, x, gy=None):
    """run the same module list on CPU (stock torch, bf16-faithful) and on the GPU plan"""
    from yolo import engine
    import copy
    mods_gpu = copy.deepcopy(mods_cpu).cuda()
    ref = bf16_faithful(mods_cpu)
    xc = _bf(x).clone().requires_grad_(True)
    yc = ref(xc)
    plan = engine.Plan.from_modules(list(mods_gpu), x.shape[1], False)
    xg = x.clone().cuda().requires_grad_(True)
    mods_gpu.eval()
    yg = engine.run_plan(plan, xg, False)
    if gy is None:
        return yc, yg, None, None
    yc.backward(gy)
    yg.backward(gy.cuda())
    gc = [xc.grad] + [p.grad for p in mods_cpu.parameters()]
    gg = [xg.grad] + [p.grad for p in mods_gpu.parameters()]
    return yc, yg, gc, gg


def test_golden_conv_layers(golden):
    g = golden("layers_small.npz")
    for name in ("c3x3", "c3x3s2", "c1x1"):
        ks, st, pd = (int(v) for v in g[f"{name}__cfg"])
        w = torch.from_numpy(g[f"{name}__w"])
        conv = nn.Conv2d(w.shape[1], w.shape[0], ks, st, pd)
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.bias.copy_(torch.from_numpy(g[f"{name}__b"]))
        mods = nn.Sequential(conv, nn.LeakyReLU(0.1))
        x = torch.from_numpy(g[f"{name}__x"])
        gy = torch.from_numpy(g[f"{name}__gy"])
        yc, yg, gc, gg = _run_both(mods, x, gy)
        _close(yg, yc, 1.0, f"{name} y")
        # and against the fp32 fixture itself (adds the operand rounding: sqrt(K) * 2^-8 relative)
        assert (yg.cpu() - torch.from_numpy(g[f"{name}__y"])).abs().max() < 0.05
        for a, b, what in zip(gg, gc, ("gx", "gw", "gb")):
            _close(a, b, 3.0, f"{name} {what}")
        # (no direct comparison of gradients with the fp32 fixture: rounding the operands flips the
        #  LeakyReLU gate of pre-activations near zero, a x10 change of that pixel's contribution)
        gw_ref = torch.from_numpy(g[f"{name}__gw"])
        assert ((gg[1].cpu() - gw_ref).norm() / gw_ref.norm()).item() < 0.05


def test_first_layer_7x7_s2_and_pool(golden):
    g = golden("layers_small.npz")
    w = torch.from_numpy(g["c7x7s2__w"])
    conv = nn.Conv2d(3, 64, 7, 2, 3)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(torch.from_numpy(g["c7x7s2__b"]))
    from yolo import engine
    import copy
    mods_c = nn.Sequential(conv, nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2))
    mods_g = copy.deepcopy(mods_c).cuda()
    x = torch.from_numpy(g["c7x7s2__x"])
    yc = bf16_faithful(mods_c)(_bf(x))
    plan = engine.Plan.from_modules(list(mods_g), 3, True)
    yg = engine.run_plan(plan, x.cuda(), False)
    _close(yg, yc, 1.0, "7x7s2+pool y")
    gy = torch.randn(yc.shape, generator=torch.Generator().manual_seed(3))
    yc.backward(gy)
    yg.backward(gy.cuda())
    _close(mods_g[0].weight.grad, conv.weight.grad, 3.0, "7x7s2 gw")
    _close(mods_g[0].bias.grad, conv.bias.grad, 3.0, "7x7s2 gb")


def test_maxpool_exact(golden):
    """bf16 inputs that are exactly representable -> forward pool is exact; backward routes to the first max."""
    from yolo import engine
    g = golden("layers_small.npz")
    x = _bf(torch.from_numpy(g["pool__x"]))
    conv = nn.Conv2d(32, 32, 1)
    with torch.no_grad():
        conv.weight.zero_()
        conv.weight[:, :, 0, 0] = torch.eye(32)
        conv.bias.zero_()
    mods = nn.Sequential(conv, nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2))
    yc, yg, gc, gg = _run_both(mods, x, torch.from_numpy(g["pool__gy"]))
    assert torch.equal(yg.cpu(), yc)
    _close(gg[0], gc[0], 1.0, "pool gx")


def test_conv_pool_conv_chain_and_tails():
    """odd sizes: pixel-tile tails (M not a multiple of 128), Cout=192 (3 x 64 tiles), stride-2 in the middle."""
    torch.manual_seed(0)
    mods = nn.Sequential(
        nn.Conv2d(64, 192, 3, 1, 1), nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2),
        nn.Conv2d(192, 128, 1), nn.LeakyReLU(0.1),
        nn.Conv2d(128, 256, 3, 2, 1), nn.LeakyReLU(0.1),
        nn.Conv2d(256, 64, 3, 1, 1), nn.LeakyReLU(0.1))
    x = torch.randn(3, 64, 20, 28)
    yc0 = mods(x)
    gy = torch.randn(yc0.shape)
    yc, yg, gc, gg = _run_both(mods, x, gy)
    _close(yg, yc, 8.0, "chain y")
    for i, (a, b) in enumerate(zip(gg, gc)):
        _close(a, b, 16.0, f"chain grad {i}", frac=0.01)


def test_fc_head_split_k():
    """Flatten -> Linear(K=6272) (split-K atomics) -> LeakyReLU -> Dropout(eval) -> Linear(1470 outputs: ragged tile)."""
    torch.manual_seed(1)
    mods = nn.Sequential(
        nn.Conv2d(64, 128, 3, 1, 1), nn.LeakyReLU(0.1),
        nn.Flatten(), nn.Linear(128 * 7 * 7, 512), nn.LeakyReLU(0.1), nn.Dropout(0.5), nn.Linear(512, 1470))
    mods.eval()
    x = torch.randn(5, 64, 7, 7)
    gy = torch.randn(5, 1470)
    yc, yg, gc, gg = _run_both(mods, x, gy)
    _close(yg, yc, 4.0, "fc y")
    for i, (a, b) in enumerate(zip(gg, gc)):
        _close(a, b, 8.0, f"fc grad {i}", frac=0.01)


def test_dropout_training_mask_consistency():
    """training mode: the mask drawn in forward is the one applied in backward (grad zero where output zero)."""
    from yolo import engine
    torch.manual_seed(2)
    mods = nn.Sequential(nn.Flatten(), nn.Linear(64 * 4 * 4, 256), nn.LeakyReLU(0.1), nn.Dropout(0.5), nn.Linear(256, 24)).cuda()
    conv = nn.Sequential(nn.Conv2d(64, 64, 1), nn.LeakyReLU(0.1)).cuda()
    plan = engine.Plan.from_modules(list(conv) + list(mods), 64, False)
    x = torch.randn(4, 64, 4, 4, device="cuda")
    y = engine.run_plan(plan, x, True)
    y.sum().backward()
    gw2 = mods[4].weight.grad            # (24, 256): column k is zero iff unit k was dropped for every sample
    dropped_cols = (gw2.abs().sum(0) == 0).sum().item()
    assert 0 < dropped_cols < 256 * 0.3   # P(all 4 samples dropped) = 1/16
    assert torch.isfinite(y).all()


def test_layout_roundtrip():
    from yolo._hip import lib, check, ptr, stream
    x = torch.randn(2, 40, 9, 11, device="cuda")
    from yolo.engine import Act
    a = Act(2, 9, 11, 40, 1, x.device)
    check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), 2, 40, 9, 11, a.p, 40, 1, 1, stream()))
    back = torch.empty_like(x)
    check(lib().yolo_nhwc_bf16_to_nchw_f32(a.p, 2, 40, 9, 11, 1, ptr(back), stream()))
    assert torch.equal(back, _bf(x))
    v = a.view()
    assert v[:, 0].abs().sum() == 0 and v[:, -1].abs().sum() == 0 and v[:, :, 0].abs().sum() == 0 and v[:, :, -1].abs().sum() == 0


def test_multi_layer_pack_and_unpack_are_exact():
    """yolo_pack_conv_weights_multi / yolo_unpack_conv_wgrads_multi == pure index permutations of the OIHW tensor."""
    import ctypes
    from yolo._hip import ConvPackItem, ConvUnpackItem, check, lib, stream
    torch.manual_seed(3)
    shapes = [(64, 64, 3), (128, 192, 1), (192, 64, 3), (256, 128, 3), (64, 128, 1)]
    ws = [torch.randn(co, ci, k, k, device="cuda") for co, ci, k in shapes]
    wf = [torch.empty(co, k, k, ci, dtype=torch.bfloat16, device="cuda") for co, ci, k in shapes]
    wd = [torch.empty(ci, k, k, co, dtype=torch.bfloat16, device="cuda") for co, ci, k in shapes]
    items = [ConvPackItem(w.data_ptr(), f.data_ptr(), d.data_ptr() if i != 1 else None, co, ci, k, k)
             for i, (w, f, d, (co, ci, k)) in enumerate(zip(ws, wf, wd, shapes))]
    check(lib().yolo_pack_conv_weights_multi((ConvPackItem * len(items))(*items), len(items), stream()))
    for i, (w, f, d) in enumerate(zip(ws, wf, wd)):
        assert torch.equal(f, w.permute(0, 2, 3, 1).to(torch.bfloat16)), f"forward operand {i}"
        if i != 1:
            assert torch.equal(d, w.flip(2, 3).permute(1, 2, 3, 0).to(torch.bfloat16)), f"data-gradient operand {i}"
    packed = [torch.randn(co, k, k, ci, device="cuda") for co, ci, k in shapes]
    out = [torch.empty(co, ci, k, k, device="cuda") for co, ci, k in shapes]
    it2 = [ConvUnpackItem(a.data_ptr(), b.data_ptr(), co, ci, k, k) for a, b, (co, ci, k) in zip(packed, out, shapes)]
    check(lib().yolo_unpack_conv_wgrads_multi((ConvUnpackItem * len(it2))(*it2), len(it2), stream()))
    for a, b in zip(packed, out):
        assert torch.equal(b, a.permute(0, 3, 1, 2))
    # rejected, not silently mis-packed
    bad = ConvPackItem(ws[0].data_ptr(), wf[0].data_ptr(), None, 60, 64, 3, 3)
    assert lib().yolo_pack_conv_weights_multi((ConvPackItem * 1)(bad), 1, stream()) != 0


def test_stem_forward_kernel_matches_generic_igemm():
    """yolo_conv_stem7_fwd (patch staged once, weights in registers) vs the generic row-segment implicit GEMM: same
    bf16-rounded operands, fp32 accumulation in a different order -> equal to a bf16 ulp; with and without the fused pool."""
    from yolo import engine
    torch.manual_seed(12)
    mods = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3), nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2)).cuda().eval()
    x = torch.randn(3, 3, 96, 128, device="cuda")          # output 48 x 64 -> 6 x 4 tiles per image
    outs = {}
    for stem in (True, False):
        for fuse in (True, False):
            engine.STEM_KERNEL, engine.FUSE_POOL = stem, fuse
            try:
                plan = engine.Plan.from_modules(list(mods), 3, True)
                with torch.no_grad():
                    outs[(stem, fuse)] = engine.run_plan(plan, x, False)
            finally:
                engine.STEM_KERNEL, engine.FUSE_POOL = True, True
    assert torch.equal(outs[(True, True)], outs[(True, False)]), "fused pool must equal conv-then-pool bit for bit"
    _close(outs[(True, True)], outs[(False, True)], 1.0, "stem kernel vs igemm")
    ref = mods.cpu()(_bf(x.cpu()))
    _close(outs[(True, True)], ref, 2.0, "stem kernel vs torch")


def test_stem_wgrad_direct_kernel():
    """yolo_wgrad_stem7 (7x7/s2 stem, unfolding in the LDS read addresses) vs fp32 torch on bf16-rounded operands:
    several tiles per workgroup, N > 1, and bit-reproducible (fixed-order partial sums)."""
    from yolo._hip import lib, check, ptr, stream
    from yolo.engine import Act
    torch.manual_seed(9)
    N, H, W = 3, 64, 96                      # output 32 x 48 -> 4 x 3 tiles per image
    x = torch.randn(N, 3, H, W, device="cuda")
    dy = torch.randn(N, 64, H // 2, W // 2, device="cuda") * 0.25
    xa = Act(N, H, W, 4, 3, x.device)
    check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, 3, H, W, xa.p, 4, 3, 3, stream()))
    ga = Act(N, H // 2, W // 2, 64, 1, x.device)
    check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(dy), N, 64, H // 2, W // 2, ga.p, 64, 1, 1, stream()))
    outs = []
    for G in (768, 5):                       # 5 workgroups: every workgroup walks over several tiles
        part = torch.full((G * 14400,), float("nan"), device="cuda")
        dw = torch.full((64, 3, 7, 7), float("nan"), device="cuda")
        db = torch.full((64,), float("nan"), device="cuda")
        check(lib().yolo_wgrad_stem7(xa.p, ga.p, N, H // 2, W // 2, xa.img_stride, xa.row_stride, ga.img_stride, ga.row_stride, ga.interior_off(),
                                     ptr(dw), ptr(db), ptr(part), part.numel(), stream()))
        outs.append((dw, db))
    xr = _bf(x).double().cpu().requires_grad_(True)
    w = torch.zeros(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(64, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w, b, stride=2, padding=3).backward(_bf(dy).double().cpu())
    for dw, db in outs:
        torch.testing.assert_close(dw.cpu().double(), w.grad, rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(db.cpu().double(), b.grad, rtol=1e-4, atol=2e-3)
    dw2 = torch.empty_like(outs[0][0]); db2 = torch.empty_like(outs[0][1])
    part = torch.empty((768 * 14400,), device="cuda")
    check(lib().yolo_wgrad_stem7(xa.p, ga.p, N, H // 2, W // 2, xa.img_stride, xa.row_stride, ga.img_stride, ga.row_stride, ga.interior_off(),
                                 ptr(dw2), ptr(db2), ptr(part), part.numel(), stream()))
    assert torch.equal(dw2, outs[0][0]) and torch.equal(db2, outs[0][1])
    assert lib().yolo_wgrad_stem7(xa.p, ga.p, N, 30, 48, xa.img_stride, xa.row_stride, ga.img_stride, ga.row_stride, ga.interior_off(),
                                  ptr(dw2), ptr(db2), ptr(part), part.numel(), stream()) != 0


@pytest.mark.parametrize("Cin", [192, 64, 128, 512])
def test_wgrad_kernel_variants_agree_bit_for_bit(Cin):
    """yolo_wgrad: the 128x128 kernel, the 256x128 3-stage kernel and its staggered two-phase form reduce every output over
    the pixels in the same order (16-pixel MFMA sub-steps), so with one pixel range per tile they must agree bit for bit --
    weights and fused bias gradient; ragged pixel count, 3x3 taps, several tiles."""
    import ctypes
    from yolo._hip import WgradDesc, check, lib, ptr, stream
    from yolo.engine import Act
    torch.manual_seed(21)
    N, H, W, Cout = 3, 9, 11, 512          # Cin = 64: the 128 x 128 kernel packs two taps per tile (pair_taps), the others do not
    x = Act(N, H, W, Cin, 1, torch.device("cuda")); dy = Act(N, H, W, Cout, 1, torch.device("cuda"))
    x.interior().copy_(torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16))
    dy.interior().copy_((torch.randn(N, H, W, Cout, device="cuda") * 0.1).to(torch.bfloat16))
    outs = []
    for variant in (1, 2, 3, 4, 5, 6):       # 5: 256 x 256 tile, register-pipelined loop (wgrad_pipe.hip); 6: four waves of 128 x 128 (wgrad_wide.hip)
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        wd = WgradDesc(dy.slots, dy.px_stride, x.px_stride, Cout, Cin, 3, 3, 1, x.row_stride, 1, 0, variant)
        check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dw), ptr(db), stream()))
        outs.append((dw, db))
    ref = torch.einsum("nhwo,nhwkc->okc", dy.interior().float(),
                       torch.stack([x.view()[:, ky: ky + H, kx: kx + W, :].float() for ky in range(3) for kx in range(3)], dim=3)).view(Cout, 3, 3, Cin)
    torch.testing.assert_close(outs[0][0], ref, rtol=2e-3, atol=2e-3)
    for dw, db in outs[1:]:
        assert torch.equal(dw, outs[0][0]) and torch.equal(db, outs[0][1])
    # the pipelined kernels in pixel-geometry mode (interior pixels only) and with the library's two-segment schedule (atomics)
    for variant in (5, 6):
        for split in (1, 0):
            dw = torch.zeros((Cout, 3, 3, Cin), device="cuda")
            db = torch.zeros(Cout, device="cuda")
            wd = WgradDesc(N * H * W, dy.px_stride, x.px_stride, Cout, Cin, 3, 3, 1, x.row_stride, split, 0, variant, W, H, dy.Hp * dy.Wp, dy.Wp, 1, dy.halo * dy.Wp + dy.halo)
            check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dw), ptr(db), stream()))
            torch.testing.assert_close(dw, outs[0][0], rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(db, outs[0][1], rtol=1e-4, atol=1e-4)
    # slab mode: partial tiles stored + summed in pixel-range order -- no zero fill needed, bit-reproducible, equal to the atomics' result
    # up to the order of the sum; uniform split (3 ranges) and the library's two-segment schedule; both kernels produce the same partial tiles
    slab_runs = {}
    for variant in (5, 6):
        for split in (3, 0):
            wd = WgradDesc(N * H * W, dy.px_stride, x.px_stride, Cout, Cin, 3, 3, 1, x.row_stride, split, 0, variant, W, H, dy.Hp * dy.Wp, dy.Wp, 1, dy.halo * dy.Wp + dy.halo)
            need = ctypes.c_long(0)
            check(lib().yolo_wgrad_slab_floats(ctypes.byref(wd), ctypes.byref(need)))
            assert need.value > 0 and need.value % (256 * 256) == 0
            slabs = torch.full((need.value,), float("nan"), device="cuda")
            wd.slabs, wd.slab_floats = slabs.data_ptr(), slabs.numel()
            runs = []
            for _ in range(2):
                dw = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
                db = torch.zeros(Cout, device="cuda")
                check(lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dw), ptr(db), stream()))
                runs.append(dw)
            torch.testing.assert_close(runs[0], outs[0][0], rtol=1e-4, atol=1e-4)
            assert torch.equal(runs[0], runs[1])
            slab_runs[(variant, split)] = runs[0]
            wd.slab_floats = need.value - 1
            assert lib().yolo_wgrad(ctypes.byref(wd), x.p, dy.p, ptr(dw), ptr(db), stream()) != 0      # scratch too small: refused
    for split in (3, 0):
        assert torch.equal(slab_runs[(5, split)], slab_runs[(6, split)])


def test_stride2_data_gradient_by_parity_classes():
    """3x3 / stride-2 conv backward: four parity-class convs over the non-zero gradient slots (engine.STRIDE2_CLASSES) vs the
    plain data gradient over the zero-stuffed buffer vs the bf16-faithful CPU reference."""
    from yolo import engine
    torch.manual_seed(8)
    mods = nn.Sequential(nn.Conv2d(64, 128, 3, 1, 1), nn.LeakyReLU(0.1), nn.Conv2d(128, 256, 3, 2, 1), nn.LeakyReLU(0.1),
                         nn.Conv2d(256, 64, 3, 1, 1), nn.LeakyReLU(0.1)).eval()
    x = torch.randn(3, 64, 12, 10)
    gy = torch.randn(3, 64, 6, 5)
    import copy
    res = {}
    for flag in (True, False):
        engine.STRIDE2_CLASSES = flag
        try:
            res[flag] = _run_both(copy.deepcopy(mods), x, gy)      # fresh copy: the CPU side accumulates .grad
        finally:
            engine.STRIDE2_CLASSES = True
    yc, yg, gc, gg = res[True]
    names = ["dx"] + [n for n, _ in mods.named_parameters()]
    for n, a, b in zip(names, gg, gc):
        _close(a, b, 6.0, f"classes vs cpu: {n}", frac=0.01)
    for n, a, b in zip(names, gg, res[False][3]):
        _close(a, b, 2.0, f"classes vs zero-stuffed: {n}", frac=0.005)


def test_fc_dgrad_behind_flatten_matches_cpu():
    """Linear behind nn.Flatten: the data gradient goes through yolo_wgrad (transposed product) + yolo_fc_dgrad_to_nhwc."""
    torch.manual_seed(5)
    mods = nn.Sequential(nn.Conv2d(32, 64, 3, padding=1), nn.LeakyReLU(0.1), nn.Flatten(), nn.Linear(64 * 5 * 6, 192), nn.LeakyReLU(0.1),
                         nn.Dropout(0.5), nn.Linear(192, 70)).eval()
    x = torch.randn(5, 32, 5, 6)
    gy = torch.randn(5, 70)
    yc, yg, gc, gg = _run_both(mods, x, gy)
    _close(yg, yc, 2.0, "forward")
    names = ["dx"] + [n for n, _ in mods.named_parameters()]
    for n, a, b in zip(names, gg, gc):
        _close(a, b, 3.0, n, frac=0.01)


def test_tuner_drops_a_candidate_that_computes_something_else(monkeypatch):
    """tools/tune_plans.py's path (config.AUTOTUNE): the tuner times the candidate plans of a problem and, before the fastest enters the table, compares
    its output buffer with the default plan's.  Here every candidate but the default plan is made to leave zeros behind: whatever their times, the
    default plan must be the one that is kept, and it must reproduce the un-sabotaged result."""
    from yolo import engine, plans
    torch.manual_seed(21)
    conv = nn.Conv2d(128, 256, 3, 1, 1).cuda()
    x = torch.randn(4, 128, 28, 28, device="cuda")
    saved = dict(plans._TUNED)
    real_run = plans._run_plan_igemm
    sabotaged, default = [], []

    def run(L_, d, plan, inp, w, bias, aux, out, st, what):
        real_run(L_, d, plan, inp, w, bias, aux, out, st, what)
        default.append(plans._default_plan(d))
        if sabotaged and plan != default[-1]:
            torch.cuda.synchronize()
            addr = out.value if hasattr(out, "value") else int(out)
            torch.as_tensor(plans._RawDevice(addr, d.N * d.out_img_stride, False), device="cuda").zero_()

    try:
        plans._TUNED.clear()
        engine.AUTOTUNE, engine.TUNE_LOG = True, []
        with torch.no_grad():
            ref = engine.run_plan(engine.Plan.from_modules([conv, nn.LeakyReLU(0.1)], 128, False), x, False).float().clone()
        (key, winner, top), = engine.TUNE_LOG
        assert plans._TUNED[key] == winner and top[0][0] == winner
        # again, with every candidate but the default plan sabotaged
        plans._TUNED.clear()
        engine.TUNE_LOG = []
        sabotaged.append(winner)
        monkeypatch.setattr(plans, "_run_plan_igemm", run)
        with torch.no_grad():
            plan2 = engine.Plan.from_modules([conv, nn.LeakyReLU(0.1)], 128, False)
            engine.run_plan(plan2, x, False)
            monkeypatch.setattr(plans, "_run_plan_igemm", real_run)
            got = engine.run_plan(plan2, x, False).float()
        (key2, winner2, _), = engine.TUNE_LOG
        assert key2 == key and winner2 == default[-1] and plans._TUNED[key] == winner2
        _close(got, ref, 2.0, "the kept plan")
    finally:
        engine.AUTOTUNE, engine.TUNE_LOG = False, None
        plans._TUNED.clear()
        plans._TUNED.update(saved)


@pytest.mark.parametrize("n,hw", [(1, 14), (2, 14), (1, 7), (4, 7), (3, 9)])
def test_few_pixel_deep_k_layers_split_their_k_range(n, hw):
    """small batches on the 14x14 / 7x7 maps: a problem without a table entry and < 2048 pixels under K >= 2304 runs as K ranges stored as slabs
    (plans._default_plan, CFG.SMALL_SPLIT) -- forward and data gradient, stride 1 and 2, equal to the bf16-faithful CPU chain, bit-reproducible,
    and within rounding of the plain single launch."""
    import copy
    from yolo import engine, plans
    torch.manual_seed(11)
    mods = nn.Sequential(
        nn.Conv2d(256, 512, 3, 1, 1), nn.LeakyReLU(0.1),
        nn.Conv2d(512, 512, 3, 1, 1), nn.LeakyReLU(0.1),
        nn.Conv2d(512, 1024, 3, 2, 1), nn.LeakyReLU(0.1))     # (its data gradient: four parity-class convs, the 2x2-tap one with K = 4096)
    x = torch.randn(n, 256, hw, hw)
    ho = (hw + 1) // 2
    gy = torch.randn(n, 1024, ho, ho)
    res = {}
    for on in (True, False):
        engine.SMALL_SPLIT = on
        engine.BORROW_PLANS = False          # the rule itself, not a plan measured for the same layer shape at another batch size
        saved = dict(plans._TUNED)
        try:
            for k in [k for k in plans._TUNED if k[0] == n and k[1] in (hw, ho)]:
                del plans._TUNED[k]          # the deterministic default, not a shipped entry of the same shape
            res[on] = _run_both(copy.deepcopy(mods), x, gy)        # (a fresh CPU copy: .grad accumulates)
            mine = {k: v for k, v in plans._TUNED.items() if k[0] == n and k[1] in (hw, ho) and k[5] >= 256}
            split = [k for k, v in mine.items() if v[0] == "slabs"]
            assert (len(split) >= 4) if on else (not split), mine     # three forward problems (two for the first layer without dx ...) + data gradients
            if on:
                again = _run_both(copy.deepcopy(mods), x, gy)
                assert torch.equal(again[1], res[on][1]) and torch.equal(again[3][0], res[on][3][0])
        finally:
            engine.SMALL_SPLIT = True
            engine.BORROW_PLANS = True
            plans._TUNED.clear()
            plans._TUNED.update(saved)
    yc, yg, gc, gg = res[True]
    _close(yg, yc, 3.0, "y")
    names = ["dx"] + [k for k, _ in mods.named_parameters()]
    for nme, a, b in zip(names, gg, gc):
        _close(a, b, 12.0, nme, frac=0.01)
    _close(res[True][1], res[False][1].float(), 2.0, "split vs plain: y")
    # (three LeakyReLU gates deep a few units flip under the other summation order: up to 2-3 % of the elements move by more than three roundings,
    # the relative L2 difference stays below 1 %, see _close)
    for nme, a, b in zip(names, res[True][3], res[False][3]):
        _close(a, b.float(), 3.0, "split vs plain: " + nme, frac=0.04)


@pytest.mark.parametrize("n,hw", [(1, 14), (1, 7), (5, 9)])
def test_k_range_slabs_of_one_layer_equal_the_fp64_product(n, hw):
    """one 3x3 conv 512 -> 512 on a few pixels, no activation behind it: every K-range count the default rule and the tuner may pick (2 .. 32
    ranges of 64 x 128 / 128 x 64 tiles) against the fp64 product of the same bf16 operands -- the slab sum is fp32 with ONE rounding at the
    end, so its error is that of the plain launch (half a bf16 ulp + fp32 summation noise), not larger."""
    from yolo import engine
    from yolo._hip import lib, check, ptr, stream
    torch.manual_seed(12)
    conv = nn.Conv2d(512, 512, 3, 1, 1).cuda()
    plan = engine.Plan.from_modules([conv], 512, False)
    x = torch.randn(n, 512, hw, hw, device="cuda")
    with torch.no_grad():
        engine.run_plan(plan, x, False)
    key, ws = plan._workspace(n, x.shape, x.device, False)
    a_in, a_out = ws["in"], ws["acts"][0]
    check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), n, 512, hw, hw, a_in.p, 512, 1, 1, stream()))
    L = plan.layers[0]
    wf, _ = plan._pack(0, False)
    d = plan._conv_desc(L, a_in, a_out)
    ref = torch.nn.functional.conv2d(x.bfloat16().double(), conv.weight.detach().bfloat16().double(), conv.bias.detach().double(), padding=1)
    ref = ref.permute(0, 2, 3, 1)

    def run(pl):
        a_out.t.zero_()
        engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(wf), ptr(L.bias.detach()), None, a_out.p, stream(), "test")
        return a_out.interior().clone()

    def err(t):
        return ((t.double() - ref).norm() / ref.norm()).item()

    e_plain = err(run((5, 1)))
    assert e_plain < 2.0 ** -8, e_plain                      # rms of a round-to-nearest bf16 store: 2^-9 / sqrt(3) relative ... with margin
    for hint in (3, 4, 5):
        for S in (2, 4, 8, 16, 32):
            got = run(("slabs", hint, S, 0))
            assert err(got) <= 1.05 * e_plain, (hint, S, err(got), e_plain)
            assert torch.equal(run(("slabs", hint, S, 0)), got), (hint, S)


@pytest.mark.parametrize("hint", [1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, (14, 196), (14, 49), (12, 196), (5, 98), 15, (15, 196), (15, 49), 16, (16, 112), 17, (17, 196), 18, (18, 100)])
def test_every_tile_configuration(hint):
    """the same 3x3 / 1x1 chain through each yolo_igemm tile configuration (128x128, 256x128 8-wave
    3-stage ring, 128x64, 64x128, 256x208 with the uneven staggered split), forward and data-gradient, with ragged pixel
    and channel tiles; (hint, tile_px) also limits the pixels per tile (yolo_igemm_desc.tile_px)."""
    from yolo import engine
    hint, engine.TILE_PX = hint if isinstance(hint, tuple) else (hint, 0)
    torch.manual_seed(4)
    mods = nn.Sequential(
        nn.Conv2d(64, 256, 3, 1, 1), nn.LeakyReLU(0.1),
        nn.Conv2d(256, 320, 1), nn.LeakyReLU(0.1),
        nn.Conv2d(320, 512, 3, 1, 1), nn.LeakyReLU(0.1))
    x = torch.randn(3, 64, 13, 17)
    gy = torch.randn(3, 512, 13, 17)
    engine.TILE_HINT = hint
    try:
        yc, yg, gc, gg = _run_both(mods, x, gy)
    finally:
        engine.TILE_HINT, engine.TILE_PX = 0, 0
    _close(yg, yc, 6.0, f"hint {hint} y")
    for i, (a, b) in enumerate(zip(gg, gc)):
        _close(a, b, 12.0, f"hint {hint} grad {i}", frac=0.01)


def test_pixel_range_launches_tile_the_output_exactly():
    """yolo_igemm_desc.px_begin/px_end: [0, cut) + [cut, M) with ONE configuration is bit-identical to the single launch
    (cuts on and off tile boundaries); a different small-tile configuration for the tail stays within tolerance --
    the launch plans the autotuner composes (engine._run_plan_igemm)."""
    import ctypes
    from yolo import engine
    from yolo._hip import lib, check, ptr, stream
    torch.manual_seed(6)
    conv = nn.Conv2d(64, 256, 3, 1, 1).cuda()
    plan = engine.Plan.from_modules([conv, nn.LeakyReLU(0.1)], 64, False)
    x = torch.randn(5, 64, 20, 24, device="cuda")
    with torch.no_grad():
        engine.run_plan(plan, x, False)             # packs the weights, builds the workspace
    key, ws = plan._workspace(5, x.shape, x.device, False)
    a_in, a_out = ws["in"], ws["acts"][0]
    check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), 5, 64, 20, 24, a_in.p, 64, 1, 1, stream()))
    L = plan.layers[0]
    wf, _ = plan._pack(0, False)
    d = plan._conv_desc(L, a_in, a_out)
    M = 5 * 20 * 24

    def run(pl):
        a_out.t.zero_()
        engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(wf), ptr(L.bias.detach()), None, a_out.p, stream(), "test")
        return a_out.interior().clone()

    for hint in (5, 11, 12):
        ref = run((hint, 1))
        for cut in (256, 1024, 1000, M - 1):
            assert torch.equal(run((hint, 1, cut, hint)), ref), (hint, cut)
        mixed = run((hint, 1, 1280, 3))
        _close(mixed.permute(0, 3, 1, 2), ref.permute(0, 3, 1, 2).float(), 2.0, f"hint {hint} + tail 3")
        assert torch.equal(mixed.view(-1, 256)[:1280], ref.view(-1, 256)[:1280])
        sk = run(("splitk", hint if hint != 12 else 3, 3))     # split-K + finishing pass: fp32 atomics, so not bit-identical
        _close(sk.permute(0, 3, 1, 2), ref.permute(0, 3, 1, 2).float(), 2.0, f"split-K vs hint {hint}")
        # split-K into slabs + fixed-order reduce: within rounding of the single launch and bit-reproducible run to run
        for S in (2, 4):
            sl = run(("slabs", hint, S, 0))
            _close(sl.permute(0, 3, 1, 2), ref.permute(0, 3, 1, 2).float(), 2.0, f"slabs {S} vs hint {hint}")
            for _ in range(3):
                assert torch.equal(run(("slabs", hint, S, 0)), sl), (hint, S)
    # tiles limited to tile_px pixels (hint 14: uneven staggered split) against the plain 128x128 launch
    ref = run((5, 1))
    for pl in (("tile", 14, 1, 0), ("tile", 14, 1, 196), ("tile", 14, 1, 100), ("tile", 12, 1, 196), ("tile", 14, 1, 196, 3, 4000),
               ("slabs", 14, 3, 196), ("tile", 15, 1, 0), ("tile", 15, 1, 196), ("tile", 15, 2, 100), ("tile", 15, 1, 196, 3, 4000), ("slabs", 15, 3, 196)):
        got = run(pl)
        _close(got.permute(0, 3, 1, 2), ref.permute(0, 3, 1, 2).float(), 2.0, f"plan {pl}")
        assert torch.equal(run(pl), got), pl
    assert torch.equal(run(("tile", 14, 1, 196)), run(("tile", 14, 1, 0)))    # same K order per output: bit-identical
    assert torch.equal(run(("tile", 15, 1, 196)), run(("tile", 14, 1, 196)))  # the pipelined loop adds in the same order too
    d.px_begin, d.px_end = 10, 5
    assert lib().yolo_igemm(ctypes.byref(d), a_in.p, ptr(wf), ptr(L.bias.detach()), None, a_out.p, stream()) != 0
    plan._release(key, ws)


def test_start_skew_changes_nothing_but_time():
    """yolo_igemm_desc.skew_phases / skew_step: first-round workgroups of the 8-wave configurations start a few thousand
    cycles apart -- the output must be bit-identical to the plain launch (more than 256 workgroups, so the skew is live)."""
    from yolo import engine
    from yolo._hip import lib, ptr, stream
    torch.manual_seed(8)
    conv = nn.Conv2d(64, 256, 3, 1, 1).cuda()
    plan = engine.Plan.from_modules([conv, nn.LeakyReLU(0.1)], 64, False)
    x = torch.randn(16, 64, 64, 66, device="cuda")
    with torch.no_grad():
        engine.run_plan(plan, x, False)
    key, ws = plan._workspace(16, x.shape, x.device, False)
    a_in, a_out = ws["in"], ws["acts"][0]
    from yolo._hip import check
    check(lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), 16, 64, 64, 66, a_in.p, 64, 1, 1, stream()))
    L = plan.layers[0]
    wf, _ = plan._pack(0, False)
    d = plan._conv_desc(L, a_in, a_out)

    def run(pl):
        a_out.t.zero_()
        engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(wf), ptr(L.bias.detach()), None, a_out.p, stream(), "test")
        return a_out.interior().clone()

    for hint in (11, 12):
        ref = run((hint, 1))
        for ph, step in ((3, 4000), (5, 20000)):
            assert torch.equal(run(("skew", hint, 1, ph, step)), ref), (hint, ph, step)
    assert d.skew_phases == 0 and d.skew_step == 0
    d.skew_phases = -1
    import ctypes
    assert lib().yolo_igemm(ctypes.byref(d), a_in.p, ptr(wf), ptr(L.bias.detach()), None, a_out.p, stream()) != 0
    d.skew_phases = 0
    plan._release(key, ws)


def test_fused_pool_epilogue_equals_separate_pool():
    """inference path: conv+LeakyReLU+MaxPool as one launch must equal conv, then pool (bit for bit:
    max and the bf16 rounding commute because rounding is monotone)."""
    from yolo import engine
    torch.manual_seed(5)
    mods = nn.Sequential(nn.Conv2d(64, 192, 3, 1, 1), nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2),
                         nn.Conv2d(192, 64, 1), nn.LeakyReLU(0.1)).cuda().eval()
    plan = engine.Plan.from_modules(list(mods), 64, False)
    x = torch.randn(2, 64, 16, 32, device="cuda")
    with torch.no_grad():
        y_fused = engine.run_plan(plan, x, False)
        engine.FUSE_POOL = False
        try:
            y_sep = engine.run_plan(plan, x, False)
        finally:
            engine.FUSE_POOL = True
    assert torch.equal(y_fused, y_sep)
    # and the 7x7/s2 stem followed by its pool (BK=32 configuration)
    stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3), nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2)).cuda().eval()
    plan = engine.Plan.from_modules(list(stem), 3, True)
    x = torch.randn(2, 3, 64, 64, device="cuda")
    with torch.no_grad():
        a = engine.run_plan(plan, x, False)
        engine.FUSE_POOL = False
        try:
            b = engine.run_plan(plan, x, False)
        finally:
            engine.FUSE_POOL = True
    assert torch.equal(a, b)


@pytest.mark.parametrize("W,cin,cout", [(112, 64, 192), (56, 64, 320), (28, 128, 512), (56, 128, 128)])
def test_pool_fused_into_the_pipelined_tiles(W, cin, cout):
    """conv -> LeakyReLU -> MaxPool2d(2,2) through the 224-pixel tiles of the register-pipelined kernels (tile_hint 16 / 18):
    rows of 112 (two half rows per pixel group), 56 and 28 pixels -- inference (pooled map only) and training (pool2 = 2: the
    un-pooled activation is written as well) equal the un-fused launches bit for bit, forward AND backward."""
    from yolo import engine
    torch.manual_seed(9)
    mods = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2), nn.Conv2d(cout, 64, 1), nn.LeakyReLU(0.1)).cuda()
    plan = engine.Plan.from_modules(list(mods), cin, False)
    H = 8 if W != 28 else 28
    x = torch.randn(3, cin, H, W, device="cuda")
    L = plan.layers[0]
    L.Hout, L.Wout = H, W
    assert engine.Plan._pool_fusable(L)
    with torch.no_grad():
        y_fused = engine.run_plan(plan, x, False)
        engine.FUSE_POOL = False
        try:
            y_sep = engine.run_plan(plan, x, False)
        finally:
            engine.FUSE_POOL = True
    assert torch.equal(y_fused, y_sep)

    def train_pass():
        for m in mods.parameters():
            m.grad = None
        xx = x.clone().requires_grad_(True)
        y = engine.run_plan(plan, xx, True)
        y.backward(torch.ones_like(y) / y.numel())
        return y.detach().clone(), xx.grad.clone(), [m.grad.clone() for m in mods.parameters()]

    ya, gxa, ga = train_pass()
    engine.FUSE_POOL = False
    try:
        yb, gxb, gb = train_pass()
    finally:
        engine.FUSE_POOL = True
    assert torch.equal(ya, yb) and torch.equal(gxa, gxb)
    for a, b in zip(ga, gb):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-7)      # weight gradients: fp32 atomics order only


@pytest.mark.parametrize("cin,cout,epi,stride", [(64, 256, "add", 1), (256, 64, "relu", 1), (128, 512, "add", 1), (256, 128, "gate", 1), (64, 64, "bias", 1),
                                                 (128, 192, "none", 1), (256, 512, "bias", 2), (64, 128, "none", 2), (512, 256, "relu", 1), (192, 128, "gate", 1),
                                                 (512, 64, "none", 1)])
def test_streaming_1x1_kernel(cin, cout, epi, stride):
    """yolo_igemm tile_hint 19 (igemm_stream.hip): thin-K pointwise conv with every epilogue it takes (bias, bias + ReLU, bias + residual
    + ReLU as in a ResNet bottleneck, LeakyReLU' gate of a data gradient, none) against the fp32 product of the bf16 operands, and
    against the tiled kernel (tile_hint 10) to one bf16 ulp; halo-1 buffers, 3 x 24 x 40 pixels."""
    import ctypes
    from yolo._hip import IgemmDesc, check, lib, ptr, stream, EPI_BIAS, EPI_BIAS_LRELU, EPI_MUL_DLRELU, EPI_NONE, EPI_BIAS_ADD_LRELU
    from yolo.engine import Act
    torch.manual_seed(cin + cout)
    dev = torch.device("cuda")
    N, H, W = 3, 24, 40                        # output grid; the input grid is stride times as large (1x1 / stride 2 = ResNet's downsample conv)
    x = Act(N, H * stride, W * stride, cin, 1, dev)
    x.interior().copy_(torch.randn(N, H * stride, W * stride, cin, device=dev).to(torch.bfloat16))
    aux = Act(N, H, W, cout, 1, dev)
    aux.interior().copy_(torch.randn(N, H, W, cout, device=dev).to(torch.bfloat16))
    w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device=dev)
    code, slope = {"add": (EPI_BIAS_ADD_LRELU, 0.0), "relu": (EPI_BIAS_LRELU, 0.0), "gate": (EPI_MUL_DLRELU, 0.1), "bias": (EPI_BIAS, 1.0), "none": (EPI_NONE, 1.0)}[epi]
    outs = {}
    for hint in (19, 10):
        y = Act(N, H, W, cout, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, H, W
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = x.img_stride, x.row_stride, x.px_stride, x.interior_off()
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = stride, 1, 1, cin, cout
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = y.img_stride, y.row_stride, y.px_stride, y.interior_off()
        d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = aux.img_stride, aux.row_stride, aux.px_stride, aux.interior_off()
        d.epilogue, d.slope, d.out_fp32, d.split_k, d.tile_hint, d.tile_order = code, slope, 0, 1, hint, 1
        acc = None
        if epi == "none":          # BatchNorm statistics of the stored outputs (yolo_igemm_desc.bn_stats), as the ResNet trunk in training mode asks for
            from yolo._hip import BN_ACC_REPLICAS
            acc = torch.zeros(BN_ACC_REPLICAS * 2 * cout, dtype=torch.float64, device=dev)
            d.bn_stats = acc.data_ptr()
        check(lib().yolo_igemm(ctypes.byref(d), x.p, ptr(w), ptr(b) if epi in ("add", "relu", "bias") else None,
                               aux.p if epi in ("add", "gate") else None, y.p, stream()), f"igemm hint {hint}")
        outs[hint] = y
        if acc is not None:
            st = acc.view(BN_ACC_REPLICAS, 2, cout).sum(0).cpu()
            stored = y.interior().double().cpu().reshape(-1, cout)
            torch.testing.assert_close(st[0], stored.sum(0), rtol=1e-5, atol=1e-3)
            torch.testing.assert_close(st[1], stored.pow(2).sum(0), rtol=1e-5, atol=1e-3)
    z = torch.einsum("nhwc,oc->nhwo", x.interior()[:, ::stride, ::stride, :].float(), w.float())
    a = aux.interior().float()
    if epi == "add":
        ref = torch.relu(z + b + a)
    elif epi == "relu":
        ref = torch.relu(z + b)
    elif epi == "gate":
        ref = z * torch.where(a > 0, 1.0, 0.1)
    elif epi == "bias":
        ref = z + b
    else:
        ref = z
    got = outs[19].interior().float()
    _close(got.cpu(), _bf(ref.cpu()), 1.0, f"streaming 1x1 {cin}->{cout} {epi} vs fp32")
    _close(got.cpu(), outs[10].interior().float().cpu(), 1.0, f"streaming 1x1 {cin}->{cout} {epi} vs tile_hint 10")
    halo = outs[19].view().clone()
    halo[:, 1: 1 + H, 1: 1 + W, :] = 0
    assert not bool(halo.any()), "the halo must stay zero"


def _persist_problem(N, cin, cout, k, H, W, epi, pool=0, stride=1):
    """one yolo_igemm problem on random data: (descriptor, input, weight panel, bias, aux, output Act)"""
    from yolo import engine
    from yolo._hip import EPI_BIAS_ADD_LRELU, EPI_BIAS_LRELU, EPI_MUL_DLRELU, EPI_NONE, IgemmDesc
    g = torch.Generator(device="cuda").manual_seed(1000 * cin + cout + k)
    pad = (k - 1) // 2
    a_in = engine.Act(N, H, W, cin, 1, "cuda")
    a_in.interior().copy_(torch.randn(N, H, W, cin, device="cuda", generator=g))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    a_out = engine.Act(N, Ho // 2 if pool else Ho, Wo // 2 if pool else Wo, cout, 1, "cuda")
    aux = engine.Act(N, Ho, Wo, cout, 1, "cuda")
    aux.interior().copy_(torch.randn(N, Ho, Wo, cout, device="cuda", generator=g))
    w = (torch.randn(cout, k, k, cin, device="cuda", generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda", generator=g)
    d = IgemmDesc()
    d.N, d.Ho, d.Wo = N, Ho, Wo
    d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a_in.img_stride, a_in.row_stride, a_in.px_stride, a_in.interior_off(pad)
    d.stride, d.KH, d.KW, d.tap_len, d.Cout = stride, k, k, cin, cout
    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = a_out.img_stride, a_out.row_stride, a_out.px_stride, a_out.interior_off()
    d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = aux.img_stride, aux.row_stride, aux.px_stride, aux.interior_off()
    d.epilogue = {"lrelu": EPI_BIAS_LRELU, "gate": EPI_MUL_DLRELU, "none": EPI_NONE, "add": EPI_BIAS_ADD_LRELU}[epi]
    d.slope, d.out_fp32, d.split_k, d.pool2 = 0.1, 0, 1, pool
    return d, a_in, w, b, aux, a_out


@pytest.mark.parametrize("cin,cout,k,epi,tpx", [(64, 512, 3, "lrelu", 196), (64, 512, 3, "lrelu", 208), (192, 256, 1, "lrelu", 196), (256, 512, 1, "gate", 196),
                                                 (128, 256, 3, "gate", 208), (64, 192, 3, "lrelu", 196), (512, 256, 1, "none", 100), (64, 256, 3, "lrelu", 224),
                                                 (256, 1024, 1, "add", 208), (192, 256, 1, "add", 224), (128, 128, 3, "lrelu", 196), (64, 64, 3, "gate", 208),
                                                 (256, 512, 3, "lrelu", -2), (512, 256, 1, "lrelu", -1)])
def test_persistent_kernel_equals_the_pipelined_one(cin, cout, k, epi, tpx):
    """igemm_persist.hip (tile_hint 20 / 21): several tiles per workgroup in ONE software pipeline (61 440 pixels: 296-615 pixel tiles x
    1-2 channel tiles for 256 workgroups), table of the next tile built under the K loop, epilogue straight out of the accumulator
    registers -- the same MFMAs in the same order as tile_hint 15 / 16, so every output is bit-identical; also run twice (no
    run-to-run differences: the duplicate stores of a tile's idle pixel slots write identical bits)."""
    from yolo import engine
    from yolo._hip import lib, ptr, stream
    if tpx == -2:        # a stride-2 conv (ResNet's downsample / 3x3 s2 layers run these plans), ragged last tile
        d, a_in, w, b, aux, a_out = _persist_problem(8, cin, cout, k, 80, 96, epi, stride=2)
        tpx = 196
    elif tpx == -1:      # fewer tiles than CUs: one image, 5 pixel tiles -- every workgroup has ONE tile, most CUs none
        d, a_in, w, b, aux, a_out = _persist_problem(1, cin, cout, k, 28, 28, epi)
        tpx = 196
    else:
        d, a_in, w, b, aux, a_out = _persist_problem(8, cin, cout, k, 80, 96, epi)
    hint = 21 if tpx == 224 else 20

    def run(pl):
        a_out.t.fill_(7.0)
        engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(w), ptr(b) if epi in ("lrelu", "add") else None, aux.p if epi in ("gate", "add") else None, a_out.p, stream(), "test")
        return a_out.interior().clone()

    ref = run(("tile", 16 if tpx == 224 else 15, 1, 0 if tpx == 224 else tpx))
    got = run(("tile", hint, 1, 0 if tpx == 224 else tpx))
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    assert torch.equal(run(("tile", hint, 2, 0 if tpx == 224 else tpx)), ref)          # the other tile order
    if cout % 256 == 0 and epi in ("lrelu", "none", "gate") and hint == 20 and (k * k * cin // 32) % 4 == 0 and k * k * cin // 32 >= 24:
        # K ranges as extra tiles, partial tiles stored as fp32 slabs, fixed-order sum in yolo_igemm_finish: equal to the slab form of the
        # pipelined kernel bit for bit (same MFMAs per range, same order of the ranges), and within rounding of the un-split launch
        sl = run(("slabs", 20, 2, tpx))
        assert torch.equal(sl, run(("slabs", 15, 2, tpx)))
        assert torch.equal(sl, run(("slabs", 20, 2, tpx)))
        _close(sl.permute(0, 3, 1, 2), ref.permute(0, 3, 1, 2).float(), 2.0, "K slabs vs one launch")
    assert torch.equal(a_out.view()[:, 0], torch.full_like(a_out.view()[:, 0], 7.0))   # the halo is never written


@pytest.mark.parametrize("W,H,cin,cout", [(112, 16, 64, 192), (56, 56, 128, 256), (28, 28, 256, 512)])
def test_persistent_kernel_fuses_the_pool(W, H, cin, cout):
    """tile_hint 21 with pool2 = 1: the 2x2 windows sit on lane quads (address table), maximum by DPP quad permutes, one 8-byte store per
    lane and column -- bit-identical to the pooled epilogue of tile_hint 16 (through LDS)."""
    from yolo import engine
    from yolo._hip import lib, ptr, stream
    N = 16 if W == 28 else 8
    d, a_in, w, b, aux, a_out = _persist_problem(N, cin, cout, 3, H, W, "lrelu", pool=1)

    def run(pl):
        a_out.t.fill_(7.0)
        engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(w), ptr(b), None, a_out.p, stream(), "test")
        return a_out.interior().clone()

    ref = run(("tile", 16, 1, 0))
    got = run(("tile", 21, 1, 0))
    assert torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    assert torch.equal(a_out.view()[:, 0], torch.full_like(a_out.view()[:, 0], 7.0))
    # pool2 = 3 (training): pooled map + 2-bit arg-max codes, one uint16 per (pooled pixel, 8 channels) -- map AND codes bit for bit
    d.pool2 = 3
    codes = torch.zeros(a_out.t.numel() // 8, dtype=torch.int16, device="cuda")

    def run3(pl):
        a_out.t.fill_(7.0)
        codes.fill_(-1)
        engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(w), ptr(b), ptr(codes), a_out.p, stream(), "test")
        return a_out.interior().clone(), codes.clone()

    ref_y, ref_c = run3(("tile", 16, 1, 0))
    got_y, got_c = run3(("tile", 21, 1, 0))
    assert torch.equal(ref_y, ref), "pool2 = 3 stores the same pooled map as pool2 = 1"
    assert torch.equal(got_y, ref_y)
    assert torch.equal(got_c, ref_c), int((got_c != ref_c).sum())
    assert int((ref_c != -1).sum()) >= a_out.interior().numel() // 8 - 64   # the interior codes were written (0xffff is a legal code: a few), the rest was not
    assert int((ref_c != -1).sum()) <= a_out.interior().numel() // 8


def test_persistent_kernels_draw_their_tiles_while_others_hold_the_chip():
    """The tile queue of igemm_persist.hip: a workgroup's first tile is static, the others are drawn from per-XCD counters, so a launch
    stays correct (and work-conserving) when its workgroups are NOT all resident at once.  Two persistent launches of different
    problems run at the same time on two streams -- each needs a CU per workgroup (141 KB of LDS), so they take the chip from each
    other in whatever order the dispatcher decides -- 40 times over; every output must equal the launch's result when it ran alone.
    (Also crosses the 1024-slot counter ring's reuse: > 1100 persistent launches in this test.)"""
    from yolo import engine
    from yolo._hip import lib, ptr, stream
    pa = _persist_problem(8, 64, 512, 3, 80, 96, "lrelu")          # 628 tiles: 2-3 per workgroup
    pb = _persist_problem(8, 256, 512, 1, 80, 96, "gate")          # another kernel instantiation (data-gradient epilogue)
    plan = ("tile", 20, 1, 196)

    def launch(prob, st):
        d, a_in, w, b, aux, a_out = prob
        epi_bias = d.epilogue in (1, 2, 4)          # YOLO_EPI_BIAS, _BIAS_LRELU, _BIAS_ADD_LRELU
        engine._run_plan_igemm(lib(), d, plan, a_in.p, ptr(w), ptr(b) if epi_bias else None, aux.p if d.epilogue in (3, 4) else None, a_out.p, st, "test")

    refs = []
    for prob in (pa, pb):
        prob[5].t.fill_(7.0)
        launch(prob, stream())
        torch.cuda.synchronize()
        refs.append(prob[5].interior().clone())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    import ctypes
    for rep in range(40):
        pa[5].t.fill_(7.0)
        pb[5].t.fill_(7.0)
        torch.cuda.synchronize()
        for k in range(7):                          # several launches deep on both streams: the two kernels keep meeting
            launch(pa, ctypes.c_void_p(s1.cuda_stream))
            launch(pb, ctypes.c_void_p(s2.cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(pa[5].interior(), refs[0]), rep
        assert torch.equal(pb[5].interior(), refs[1]), rep
    # 600 more single launches: every slot of the counter ring has been used and reset at least once
    for k in range(600):
        launch(pa, stream())
    torch.cuda.synchronize()
    assert torch.equal(pa[5].interior(), refs[0])


@pytest.mark.parametrize("N,H,W,epi", [(2, 32, 48, "lrelu"), (3, 16, 16, "none"), (16, 64, 48, "none"), (64, 112, 112, "lrelu")])
def test_small_channel_3x3_kernel_equals_the_tiled_one(N, H, W, epi):
    """conv_c64.hip (tile_hint 22): 3x3 / stride-1 conv of 64 -> 64 channels with the weight panel resident in LDS and the input patch of a 16 x 16
    tile staged once (ResNet-50's first-stage 3x3 convs: src/yolo/models.py:131-176).  K blocks in the implicit GEMM's order, the same MFMA:
    bit-identical to the tiled kernel (tile_hint 4), also at the real size (64 x 112 x 112: 3136 tiles over 256 workgroups, double-buffered patches);
    the halo of the output buffer stays untouched; shapes it does not take are refused."""
    from yolo import engine
    from yolo._hip import HipUnsupported, lib, ptr, stream
    d, a_in, w, b, aux, a_out = _persist_problem(N, 64, 64, 3, H, W, epi)
    if epi == "lrelu" and N == 2:
        d.slope = 0.0            # ReLU: what the folded BatchNorm + ReLU of the ResNet trunk runs

    def run(pl):
        a_out.t.fill_(7.0)
        engine._run_plan_igemm(lib(), d, pl, a_in.p, ptr(w), ptr(b) if epi == "lrelu" else None, None, a_out.p, stream(), "test")
        return a_out.interior().clone()

    ref = run((4, 1))
    got = run((22, 1))
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    assert torch.equal(a_out.view()[:, 0], torch.full_like(a_out.view()[:, 0], 7.0)) and torch.equal(a_out.view()[:, :, 0], torch.full_like(a_out.view()[:, :, 0], 7.0))
    if N == 2:
        d2, a2, w2, b2, _, o2 = _persist_problem(2, 64, 64, 3, 24, 48, "lrelu")          # 24 rows: not a multiple of 16
        with pytest.raises(HipUnsupported):
            engine._run_plan_igemm(lib(), d2, (22, 1), a2.p, ptr(w2), ptr(b2), None, o2.p, stream(), "test")
        d3, a3, w3, b3, _, o3 = _persist_problem(2, 128, 64, 3, 32, 48, "lrelu")         # 128 input channels
        with pytest.raises(HipUnsupported):
            engine._run_plan_igemm(lib(), d3, (22, 1), a3.p, ptr(w3), ptr(b3), None, o3.p, stream(), "test")
    if epi == "none":
        # yolo_igemm_desc.bn_stats (the raw conv of the ResNet trunk in training mode): per-channel sum and sum of squares of the values AS STORED, spread over
        # the accumulator replicas; same output bits
        from yolo._hip import BN_ACC_REPLICAS
        acc = torch.zeros(BN_ACC_REPLICAS * 2 * 64, dtype=torch.float64, device="cuda")
        d.bn_stats = acc.data_ptr()
        try:
            got_s = run((22, 1))
        finally:
            d.bn_stats = None
        assert torch.equal(got_s, ref)
        tot = acc.view(BN_ACC_REPLICAS, 2, 64).sum(0)
        y = ref.double().reshape(-1, 64)
        assert torch.allclose(tot[0], y.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(tot[1], (y * y).sum(0), rtol=1e-5, atol=1e-3)
