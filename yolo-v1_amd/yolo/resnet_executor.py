"""Executor of the ResNet-50 trunk (``yolo.resnet.resnet50_trunk``; reference src/yolo/models.py:131-176) on the same kernels: inference with
BatchNorm folded, the frozen trunk in training mode (batch statistics), and the trainable trunk's forward / backward."""

from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _hip
from ._hip import (EPI_BIAS, EPI_BIAS_LRELU, EPI_NONE, ConvPackItem, ConvUnpackItem, IgemmDesc, PoolDesc, WgradDesc, check, ptr)
from .config import CONFIG as CFG
from .executor import Plan
from .plans import igemm_call
from .runtime import RT, Act, _igemm, _on_side_stream, _round_up, _timed

# ====================================================================================================
# ResNet-50 trunk, inference only (BatchNorm folded into the conv that precedes it)
# ====================================================================================================
class ResNetPlan:
    """Inference executor for ``yolo.resnet.resnet50_trunk`` on the same kernels: every conv+BN(+ReLU) is
    one yolo_igemm (BN folded into the bf16 weights and an fp32 bias at pack time), the residual add + ReLU
    of a bottleneck is the epilogue of its last 1x1 conv (YOLO_EPI_BIAS_ADD_LRELU with slope 0), the stem's
    MaxPool2d(3,2,1) is yolo_maxpool3s2_fwd.  ``forward_batch_stats`` runs the same trunk with BatchNorm in training mode
    (batch statistics: conv with the raw weights, then yolo_batchnorm_train_fwd) for the FROZEN backbone of a training run;
    ``forward_train`` / ``backward_train`` are the trainable trunk of the reference's default run (src/train.py:144)."""

    def __init__(self, trunk: nn.Sequential):
        self.trunk = trunk
        self._packed = None
        self._raw = None
        self._bn_scratch = None
        self._bufs: dict = {}
        self.trace = None            # tests: a list that backward_train fills with per-block gradients

    # -- BN folding: y = gamma * (conv(x) - mean) / sqrt(var + eps) + beta
    @staticmethod
    def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d):
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = conv.weight.detach().float() * scale.view(-1, 1, 1, 1)
        b = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        if conv.bias is not None:
            b = b + conv.bias.detach().float() * scale
        return w.contiguous(), b.contiguous()

    def _pack_all(self):
        ver = tuple(int(p._version) for p in self.trunk.parameters()) + tuple(int(b._version) for b in self.trunk.buffers())
        if self._packed is not None and self._packed[0] == ver:
            return self._packed[1]
        st = RT.stream()
        out = {}

        def pack(name, conv, bn, first=False):
            w, b = self._fold(conv, bn)
            co, ci, k, _ = w.shape
            if first:
                wf = torch.empty((co, 7, 8, 4), dtype=torch.bfloat16, device=w.device)
                check(RT.lib().yolo_pack_conv_weight(ptr(w), co, 3, 7, 7, 4, 8, ptr(wf), None, st), "pack stem")
            else:
                wf = torch.empty((co, k, k, ci), dtype=torch.bfloat16, device=w.device)
                check(RT.lib().yolo_pack_conv_weight(ptr(w), co, ci, k, k, ci, k, ptr(wf), None, st), "pack")
            out[name] = (wf, b, conv)

        pack("stem", self.trunk[0], self.trunk[1], first=True)
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                pack((li, bi, 1), blk.conv1, blk.bn1)
                pack((li, bi, 2), blk.conv2, blk.bn2)
                pack((li, bi, 3), blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    pack((li, bi, "d"), blk.downsample[0], blk.downsample[1])
        self._packed = (ver, out)
        return out

    def _pack_raw(self):
        """bf16 operands of the UN-folded conv weights (batch-statistics mode: BatchNorm cannot be folded)."""
        ver = tuple(int(p._version) for n, p in self.trunk.named_parameters() if p.dim() == 4)
        if self._raw is not None and self._raw[0] == ver:
            return self._raw[1]
        st = RT.stream()
        out = {}

        def pack(name, conv, bn, first=False):
            w = conv.weight.detach().float().contiguous()
            co, ci, k, _ = w.shape
            if first:
                wf = torch.empty((co, 7, 8, 4), dtype=torch.bfloat16, device=w.device)
                check(RT.lib().yolo_pack_conv_weight(ptr(w), co, 3, 7, 7, 4, 8, ptr(wf), None, st), "pack stem")
            else:
                wf = torch.empty((co, k, k, ci), dtype=torch.bfloat16, device=w.device)
                check(RT.lib().yolo_pack_conv_weight(ptr(w), co, ci, k, k, ci, k, ptr(wf), None, st), "pack")
            out[name] = (wf, None, conv, bn)

        pack("stem", self.trunk[0], self.trunk[1], first=True)
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                pack((li, bi, 1), blk.conv1, blk.bn1)
                pack((li, bi, 2), blk.conv2, blk.bn2)
                pack((li, bi, 3), blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    pack((li, bi, "d"), blk.downsample[0], blk.downsample[1])
        self._raw = (ver, out)
        return out

    def _bn_train(self, a: Act, bn: nn.BatchNorm2d, relu: bool, residual: Act | None, dev, st, out: Act | None = None, save: torch.Tensor | None = None,
                  stats_ready: bool = False, frozen: bool = False):
        """BatchNorm with batch statistics (+ residual, + ReLU) in place on the conv output, running statistics updated.
        frozen: eval() mode with gradients -- normalise with the running statistics as they are, update nothing."""
        C = a.C
        acc, ss = self._scratch(dev)
        if C > 2048 or bn.weight is None or not bn.track_running_stats:
            raise NotImplementedError("batch-statistics BatchNorm: affine layers with running statistics and C <= 2048")
        mom = 0.1 if bn.momentum is None else bn.momentum
        check(RT.lib().yolo_batchnorm_train_fwd(a.p, a.N, a.H, a.W, C, a.halo, ptr(bn.weight.detach()), ptr(bn.bias.detach()), float(bn.eps), float(mom),
                                             ptr(bn.running_mean), ptr(bn.running_var), residual.p if residual is not None else None,
                                             residual.halo if residual is not None else 0, 1 if relu else 0, ptr(acc), ptr(ss),
                                             out.p if out is not None else None, out.halo if out is not None else 0,
                                             ptr(save) if save is not None else None, 2 if frozen else (1 if stats_ready else 0), st), "batchnorm_train_fwd")
        if not frozen:
            bn.num_batches_tracked += 1

    def _scratch(self, dev):
        if self._bn_scratch is None or self._bn_scratch[0].device != dev:
            self._bn_scratch = (torch.zeros(_hip.BN_ACC_REPLICAS * 2 * 2048, dtype=torch.float64, device=dev), torch.empty(2 * 2048, dtype=torch.float32, device=dev))
        return self._bn_scratch

    def _conv_bn_train(self, tag, a_in: Act, packed, N, relu: bool, residual: Act | None, dev, st):
        wf, _, conv, bn = packed
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (a_in.H + 2 * p - k) // s + 1, (a_in.W + 2 * p - k) // s + 1
        a_out = self._act(tag, N, Ho, Wo, conv.out_channels, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.in_off = a_in.interior_off(p)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = s, k, k, conv.in_channels, conv.out_channels
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = a_out.img_stride, a_out.row_stride, a_out.px_stride, a_out.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        d.bn_stats = self._scratch(dev)[0].data_ptr() if CFG.BN_STATS_IN_CONV else None      # the conv's epilogue accumulates BatchNorm's sums
        with _timed(str(tag), "igemm", 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * k * k):
            igemm_call(d, a_in.p, ptr(wf), None, None, a_out.p, st, f"igemm {tag}")
        self._bn_train(a_out, bn, relu, residual, dev, st, stats_ready=CFG.BN_STATS_IN_CONV)
        return a_out

    @_hip.device_guard
    def forward_batch_stats(self, x: torch.Tensor) -> torch.Tensor:
        """the trunk with its BatchNorm layers in TRAINING mode (batch statistics, running statistics updated) -- the frozen
        backbone of the reference's default training run (trainer.py:49).  Forward only: no gradient flows into the trunk."""
        _hip.require_cuda(x)
        st = RT.stream()
        pk = self._pack_raw()
        N, _, H, W = x.shape
        dev = x.device
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        a = self._act("in", N, H, W, 4, 3, dev)
        check(RT.lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, 3, H, W, a.p, 4, 3, 3, st), "nchw->nhwc4")
        wf, _, conv, bn = pk["stem"]
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        s1 = self._act("stem", N, Ho, Wo, 64, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a.img_stride, a.row_stride, a.px_stride, 0
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 2, 7, 1, 32, 64
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = s1.img_stride, s1.row_stride, s1.px_stride, s1.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        _igemm(RT.lib(), d, a.p, ptr(wf), None, None, s1.p, st, "igemm stem")
        self._bn_train(s1, bn, True, None, dev, st)
        Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        cur = self._act("pool", N, Hq, Wq, 64, 1, dev)
        pd = PoolDesc(N, Ho, Wo, 64, 1, 1)
        check(RT.lib().yolo_maxpool3s2_fwd(ctypes.byref(pd), s1.p, cur.p, st), "maxpool3s2")
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                idn = cur if blk.downsample is None else self._conv_bn_train((li, bi, "d"), cur, pk[(li, bi, "d")], N, False, None, dev, st)
                t = self._conv_bn_train((li, bi, 1), cur, pk[(li, bi, 1)], N, True, None, dev, st)
                t = self._conv_bn_train((li, bi, 2), t, pk[(li, bi, 2)], N, True, None, dev, st)
                cur = self._conv_bn_train((li, bi, 3), t, pk[(li, bi, 3)], N, True, idn, dev, st)
        out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
        check(RT.lib().yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        return out

    # ------------------------------------------------------------------ trainable trunk (forward keeps z, backward)
    def _pack_train(self):
        """bf16 forward AND data-gradient operands of the raw conv weights (the stem needs no data gradient); refreshed with
        yolo_pack_conv_weights_multi, 32 layers per launch, whenever a weight changed"""
        ver = tuple(int(p._version) for n, p in self.trunk.named_parameters() if p.dim() == 4)
        if getattr(self, "_train_pk", None) is not None and self._train_pk[0] == ver:
            return self._train_pk[1]
        st = RT.stream()
        out = self._train_pk[1] if getattr(self, "_train_pk", None) is not None else {}
        items = []

        def pack(name, conv, bn, first=False):
            w = conv.weight.detach()
            if w.dtype != torch.float32 or not w.is_contiguous():
                w = w.float().contiguous()
            co, ci, k, _ = w.shape
            if name in out:
                wf, wd = out[name][0], out[name][1]
            elif first:
                wf, wd = torch.empty((co, 7, 8, 4), dtype=torch.bfloat16, device=w.device), None
            else:
                wf = torch.empty((co, k, k, ci), dtype=torch.bfloat16, device=w.device)
                wd = torch.empty((ci, k, k, co), dtype=torch.bfloat16, device=w.device)
            if first:
                check(RT.lib().yolo_pack_conv_weight(ptr(w), co, 3, 7, 7, 4, 8, ptr(wf), None, st), "pack stem")
            elif co % 64 == 0 and ci % 64 == 0:
                items.append((ConvPackItem(w.data_ptr(), wf.data_ptr(), wd.data_ptr(), co, ci, k, k), w))
            else:
                check(RT.lib().yolo_pack_conv_weight(ptr(w), co, ci, k, k, ci, k, ptr(wf), ptr(wd), st), "pack")
            out[name] = (wf, wd, conv, bn)

        pack("stem", self.trunk[0], self.trunk[1], first=True)
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                pack((li, bi, 1), blk.conv1, blk.bn1)
                pack((li, bi, 2), blk.conv2, blk.bn2)
                pack((li, bi, 3), blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    pack((li, bi, "d"), blk.downsample[0], blk.downsample[1])
        for i in range(0, len(items), 32):
            tab = (ConvPackItem * len(items[i: i + 32]))(*[it[0] for it in items[i: i + 32]])
            check(RT.lib().yolo_pack_conv_weights_multi(tab, len(items[i: i + 32]), st), "pack_conv_weights_multi")
        self._train_pk = (ver, out)
        return out

    def _unit_fwd(self, tag, a_in: Act, packed, N, relu: bool, residual: Act | None, stats: torch.Tensor, dev, st, frozen: bool = False):
        """conv -> z (kept) -> BatchNorm(batch statistics; frozen: running statistics) [+ residual] [ReLU] -> y; returns the record the backward needs."""
        wf, wd, conv, bn = packed
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (a_in.H + 2 * p - k) // s + 1, (a_in.W + 2 * p - k) // s + 1
        z = self._act((tag, "z"), N, Ho, Wo, conv.out_channels, 1, dev)
        y = self._act((tag, "y"), N, Ho, Wo, conv.out_channels, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.in_off = a_in.interior_off(p)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = s, k, k, conv.in_channels, conv.out_channels
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = z.img_stride, z.row_stride, z.px_stride, z.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        in_conv = CFG.BN_STATS_IN_CONV and not frozen
        d.bn_stats = self._scratch(dev)[0].data_ptr() if in_conv else None      # the conv's epilogue accumulates BatchNorm's sums
        with _timed(str(tag), "igemm", 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * k * k):
            igemm_call(d, a_in.p, ptr(wf), None, None, z.p, st, f"igemm {tag}")
        self._bn_train(z, bn, relu, residual, dev, st, out=y, save=stats, stats_ready=in_conv, frozen=frozen)
        return {"tag": tag, "conv": conv, "bn": bn, "x": a_in, "z": z, "y": y, "relu": relu, "res": residual is not None, "stats": stats, "wd": wd,
                "k": k, "s": s, "p": p}

    @_hip.device_guard
    def forward_train(self, x: torch.Tensor, frozen: bool = False):
        """Training-mode forward of a TRAINABLE trunk (the reference's default run, src/train.py:144: ResNetBackbone(freeze=False)):
        as forward_batch_stats, but every unit keeps its conv output z, its activation y and the batch mean / invstd.
        frozen: the trunk is in eval() mode and gradients are wanted -- every BatchNorm normalises with its running statistics
        (aten batch_norm(training=False)) and updates nothing; the backward pass then has no batch terms.
        Returns (out, saved).  One forward may be in flight per plan (the buffers are reused step to step)."""
        _hip.require_cuda(x)
        st = RT.stream()
        pk = self._pack_train()
        N, _, H, W = x.shape
        dev = x.device
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        nstat = 4 * (64 + sum(u.num_features for u in self.trunk.modules() if isinstance(u, nn.BatchNorm2d)))
        if getattr(self, "_stats", None) is None or self._stats.numel() < nstat or self._stats.device != dev:
            self._stats = torch.empty(nstat, dtype=torch.float32, device=dev)
        cursor = [0]

        def stat(C):
            t = self._stats[cursor[0]: cursor[0] + 4 * C]
            cursor[0] += 4 * C
            return t

        a = self._act("in", N, H, W, 4, 3, dev)
        check(RT.lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, 3, H, W, a.p, 4, 3, 3, st), "nchw->nhwc4")
        wf, _, conv, bn = pk["stem"]
        z0 = self._act(("stem", "z"), N, Ho, Wo, 64, 1, dev)
        y0 = self._act(("stem", "y"), N, Ho, Wo, 64, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a.img_stride, a.row_stride, a.px_stride, 0
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 2, 7, 1, 32, 64
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = z0.img_stride, z0.row_stride, z0.px_stride, z0.interior_off()
        d.epilogue, d.slope = EPI_NONE, 1.0
        _igemm(RT.lib(), d, a.p, ptr(wf), None, None, z0.p, st, "igemm stem")
        stem = {"tag": "stem", "conv": conv, "bn": bn, "x": a, "z": z0, "y": y0, "relu": True, "res": False, "stats": stat(64)}
        self._bn_train(z0, bn, True, None, dev, st, out=y0, save=stem["stats"], frozen=frozen)
        Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        cur = self._act("pool", N, Hq, Wq, 64, 1, dev)
        pd = PoolDesc(N, Ho, Wo, 64, 1, 1)
        check(RT.lib().yolo_maxpool3s2_fwd(ctypes.byref(pd), y0.p, cur.p, st), "maxpool3s2")
        blocks = []
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                ud = None
                idn = cur
                if blk.downsample is not None:
                    ud = self._unit_fwd((li, bi, "d"), cur, pk[(li, bi, "d")], N, False, None, stat(blk.downsample[1].num_features), dev, st, frozen)
                    idn = ud["y"]
                u1 = self._unit_fwd((li, bi, 1), cur, pk[(li, bi, 1)], N, True, None, stat(blk.bn1.num_features), dev, st, frozen)
                u2 = self._unit_fwd((li, bi, 2), u1["y"], pk[(li, bi, 2)], N, True, None, stat(blk.bn2.num_features), dev, st, frozen)
                u3 = self._unit_fwd((li, bi, 3), u2["y"], pk[(li, bi, 3)], N, True, idn, stat(blk.bn3.num_features), dev, st, frozen)
                blocks.append((li, bi, u1, u2, u3, ud))
                cur = u3["y"]
        out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
        check(RT.lib().yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        self._train_gen = getattr(self, "_train_gen", 0) + 1
        return out, {"N": N, "dev": dev, "stem": stem, "blocks": blocks, "out": cur, "gen": self._train_gen, "frozen": frozen}

    def backward_train(self, saved, gout: torch.Tensor) -> dict:
        """gradients of every trunk parameter for the forward recorded in `saved`: {parameter: fp32 gradient}."""
        L_ = RT.lib()
        st = RT.stream()
        N, dev = saved["N"], saved["dev"]
        if saved["gen"] != self._train_gen:
            raise RuntimeError("ResNetPlan: a later training forward has reused this forward's activation buffers -- call backward() "
                               "before the next forward of the same backbone (one forward in flight per plan)")
        acc, _ = self._bn_scratch
        if getattr(self, "_coef", None) is None or self._coef.device != dev:
            self._coef = torch.empty(3 * 2048, dtype=torch.float32, device=dev)
            self._zero_bias = torch.zeros(2048, dtype=torch.float32, device=dev)
        grads: dict = {}
        convs = [u["conv"] for b in saved["blocks"] for u in b[2:] if u is not None]
        offs, tot = {}, 0
        for c in convs:
            offs[id(c)] = tot
            tot += _round_up(c.weight.numel(), 64)
        scratch = torch.zeros(tot, dtype=torch.float32, device=dev)
        pending = []
        # weight gradients (and their unpack passes) on the low-priority second stream, beside the BatchNorm-backward / data-gradient
        # chain (Plan.backward does the same): the chain's HBM-bound BatchNorm passes and the MFMA-bound weight gradients mix well
        main_t = torch.cuda.current_stream(dev)
        side_t = Plan._side_stream(dev) if CFG.WGRAD_STREAM else None

        def flush():
            with _on_side_stream(main_t, side_t):
                _flush()

        def _flush():
            for i in range(0, len(pending), 32):
                items = [ConvUnpackItem(dwp.data_ptr(), dw.data_ptr(), c.out_channels, c.in_channels, c.kernel_size[0], c.kernel_size[1])
                         for (c, dwp, dw) in pending[i: i + 32]]
                check(L_.yolo_unpack_conv_wgrads_multi((ConvUnpackItem * len(items))(*items), len(items), RT.stream()), "unpack_conv_wgrads_multi")
            pending.clear()

        def bn_bwd(u, dy: Act, store_masked: bool) -> Act:
            """dz of unit u from the gradient dy wrt its output, in the geometry of the conv's INPUT grid (zero-stuffed for stride 2)"""
            z, y, bn, s = u["z"], u["y"], u["bn"], u.get("s", 1)
            C = z.C
            if u["tag"] == "stem" or s == 1:
                dz = self._act((u["tag"], "dz"), N, z.H, z.W, C, 1, dev)
                strides = (dz.img_stride, dz.row_stride, dz.px_stride, dz.interior_off())
            else:
                xin = u["x"]
                dz = self._act((u["tag"], "dz"), N, xin.H, xin.W, C, 1, dev)
                strides = (dz.img_stride, s * dz.row_stride, s * dz.px_stride, dz.interior_off())
            dg, db = torch.empty_like(bn.weight, dtype=torch.float32), torch.empty_like(bn.bias, dtype=torch.float32)
            from_z = u["relu"] and not u["res"]        # conv -> BN -> ReLU: the mask is recomputed from z, y is not read
            check(L_.yolo_batchnorm_bwd(dy.p, dy.halo, y.p if (u["relu"] and not from_z) else None, y.halo, z.p, z.halo, N, z.H, z.W, C,
                                        ptr(bn.weight.detach()), ptr(u["stats"]), dz.p, strides[0], strides[1], strides[2], strides[3],
                                        1 if store_masked else 0, (1 if from_z else 0) | (2 if saved.get("frozen") else 0), ptr(dg), ptr(db), ptr(acc),
                                        ptr(self._coef), st),
                  f"batchnorm_bwd {u['tag']}")
            grads[bn.weight], grads[bn.bias] = dg, db
            return dz

        def wgrad(u, dz: Act):
            conv, xin, k, s, p = u["conv"], u["x"], u["k"], u["s"], u["p"]
            Hout, Wout = u["z"].H, u["z"].W
            o = offs[id(conv)]
            dwp = scratch[o: o + conv.weight.numel()]
            if Hout >= 2 and Wout >= 2 and (s > 1 or dz.Hp * dz.Wp >= 1.12 * Hout * Wout):
                wd = WgradDesc(N * Hout * Wout, dz.px_stride, xin.px_stride, conv.out_channels, conv.in_channels, k, k, p, xin.row_stride, 0, 0, 0,
                               Wout, Hout, dz.Hp * dz.Wp, dz.Wp * s, s, dz.halo * dz.Wp + dz.halo)
            else:
                wd = WgradDesc(dz.slots, dz.px_stride, xin.px_stride, conv.out_channels, conv.in_channels, k, k, p, xin.row_stride, 0, 0)
            dw = torch.empty_like(conv.weight, dtype=torch.float32)
            with _on_side_stream(main_t, side_t) as wst:
                with _timed(f"{u['tag']}.wgrad", "wgrad", 2.0 * N * Hout * Wout * conv.out_channels * conv.in_channels * k * k):
                    check(L_.yolo_wgrad(ctypes.byref(wd), xin.p, dz.p, ptr(dwp), None, wst), f"wgrad {u['tag']}")
            grads[conv.weight] = dw
            pending.append((conv, dwp, dw))

        def dgrad(u, dz: Act, add: Act | None) -> Act:
            conv, xin, k, p = u["conv"], u["x"], u["k"], u["p"]
            g = self._act((u["tag"], "gx"), N, xin.H, xin.W, conv.in_channels, 1, dev)
            d = IgemmDesc()
            d.N, d.Ho, d.Wo = N, xin.H, xin.W
            d.in_img_stride, d.in_row_stride, d.in_px_stride = dz.img_stride, dz.row_stride, dz.px_stride
            d.in_off = dz.interior_off(k - 1 - p)
            d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, k, k, conv.out_channels, conv.in_channels
            d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = g.img_stride, g.row_stride, g.px_stride, g.interior_off()
            d.slope, d.out_fp32, d.split_k = 1.0, 0, 1
            aux, bias = None, None
            d.epilogue = EPI_NONE
            if add is not None:
                d.epilogue = _hip.EPI_BIAS_ADD_LRELU          # slope 1: out = conv + 0 + aux
                d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = add.img_stride, add.row_stride, add.px_stride, add.interior_off()
                aux, bias = add.p, ptr(self._zero_bias)
            with _timed(f"{u['tag']}.dgrad", "igemm", 2.0 * N * xin.H * xin.W * conv.out_channels * conv.in_channels * k * k):
                igemm_call(d, dz.p, ptr(u["wd"]), bias, aux, g.p, st, f"dgrad {u['tag']}")
            return g

        out = saved["out"]
        gout = gout.detach()
        if gout.dtype != torch.float32 or not gout.is_contiguous():
            gout = gout.float().contiguous()
        cur_g = self._act(("g", "out"), N, out.H, out.W, out.C, 1, dev)
        check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(gout), N, out.C, out.H, out.W, cur_g.p, out.C, 1, 1, st), "gout nchw->nhwc")
        for (li, bi, u1, u2, u3, ud) in reversed(saved["blocks"]):
            if self.trace is not None:
                self.trace.append(((li, bi), "gout", cur_g.interior().float().permute(0, 3, 1, 2).contiguous()))
            dz3 = bn_bwd(u3, cur_g, True)                 # cur_g becomes g * [out > 0]: what the identity branch receives
            wgrad(u3, dz3)
            g_t2 = dgrad(u3, dz3, None)
            dz2 = bn_bwd(u2, g_t2, False)
            wgrad(u2, dz2)
            g_t1 = dgrad(u2, dz2, None)
            dz1 = bn_bwd(u1, g_t1, False)
            wgrad(u1, dz1)
            if ud is not None:
                dzd = bn_bwd(ud, cur_g, False)
                wgrad(ud, dzd)
                g_idn = dgrad(ud, dzd, None)
            else:
                g_idn = cur_g
            cur_g = dgrad(u1, dz1, g_idn)
            if self.trace is not None:
                self.trace.append(((li, bi), "gx", cur_g.interior().float().permute(0, 3, 1, 2).contiguous()))
            if len(pending) >= 24:
                flush()
        flush()
        # stem: MaxPool2d(3,2,1) backward -> BatchNorm/ReLU backward -> the direct 7x7 weight-gradient kernel
        stem = saved["stem"]
        y0, xin = stem["y"], stem["x"]
        g_y0 = self._act(("stem", "gy"), N, y0.H, y0.W, 64, 1, dev)
        pd = PoolDesc(N, y0.H, y0.W, 64, y0.halo, cur_g.halo)
        check(L_.yolo_maxpool3s2_bwd(ctypes.byref(pd), y0.p, cur_g.p, g_y0.p, g_y0.halo, st), "maxpool3s2_bwd")
        dz0 = bn_bwd(stem, g_y0, False)
        conv = stem["conv"]
        dw = torch.empty_like(conv.weight, dtype=torch.float32)
        part = getattr(self, "_stem_part", None)
        if part is None or part.device != dev:
            part = self._stem_part = torch.empty((768 * 14400,), dtype=torch.float32, device=dev)
            self._stem_db = torch.empty(64, dtype=torch.float32, device=dev)
        if y0.H % 8 == 0 and y0.W % 16 == 0:
            check(L_.yolo_wgrad_stem7(xin.p, dz0.p, N, y0.H, y0.W, xin.img_stride, xin.row_stride, dz0.img_stride, dz0.row_stride, dz0.interior_off(),
                                      ptr(dw), ptr(self._stem_db), ptr(part), part.numel(), st), "wgrad_stem7")
        else:
            # stem maps that the direct kernel's 8 x 16-pixel tiles do not cover (inputs that are not (16k) x (32k) pixels): the generic weight-gradient
            # kernel over a row-unfolded copy of the input (7 kernel rows x 8 columns x 4 channels per output pixel), as Plan.backward does
            xcol = self._act(("stem", "xcol"), N, y0.H, y0.W, 7 * 32, 1, dev)
            check(L_.yolo_im2col_rows(xin.p, xin.img_stride, xin.row_stride, xin.px_stride, 2, 7, 32, N, y0.H, y0.W, 1, xcol.p, st), "im2col_rows")
            dwp = torch.zeros(64 * 7 * 8 * 4, dtype=torch.float32, device=dev)
            wd = WgradDesc(dz0.slots, dz0.px_stride, xcol.px_stride, 64, 7 * 32, 1, 1, 0, xcol.row_stride, max(1, min(1024, dz0.slots // 4096)), 0)
            check(L_.yolo_wgrad(ctypes.byref(wd), xcol.p, dz0.p, ptr(dwp), None, st), "wgrad stem")
            check(L_.yolo_unpack_conv_wgrad(ptr(dwp), 64, 3, 7, 7, 4, 8, ptr(dw), 0, st), "unpack stem")
        grads[conv.weight] = dw
        if side_t is not None:
            main_t.wait_stream(side_t)          # every weight gradient is final before the pass returns
        return grads

    def _act(self, key, N, H, W, C, halo, dev):
        k = (key, N, H, W, C, halo, str(dev))
        a = self._bufs.get(k)
        if a is None:
            a = Act(N, H, W, C, halo, dev)
            self._bufs[k] = a
        return a

    def _conv(self, tag, a_in: Act, packed, N, relu: bool, residual: Act | None, dev, st):
        wf, b, conv = packed
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (a_in.H + 2 * p - k) // s + 1, (a_in.W + 2 * p - k) // s + 1
        a_out = self._act(tag, N, Ho, Wo, conv.out_channels, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.in_off = a_in.interior_off(p)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = s, k, k, conv.in_channels, conv.out_channels
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = a_out.img_stride, a_out.row_stride, a_out.px_stride, a_out.interior_off()
        d.slope = 0.0 if relu else 1.0
        aux = None
        if residual is not None:
            d.epilogue = _hip.EPI_BIAS_ADD_LRELU
            d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = residual.img_stride, residual.row_stride, residual.px_stride, residual.interior_off()
            aux = residual.p
        else:
            d.epilogue = EPI_BIAS_LRELU if relu else EPI_BIAS
        with _timed(str(tag), "igemm", 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * k * k):
            igemm_call(d, a_in.p, ptr(wf), ptr(b), aux, a_out.p, st, f"igemm {tag}")
        return a_out

    @_hip.device_guard
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N,3,H,W) fp32 on the device -> (N,2048,H/32,W/32) fp32."""
        _hip.require_cuda(x)
        st = RT.stream()
        pk = self._pack_all()
        N, _, H, W = x.shape
        dev = x.device
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        a = self._act("in", N, H, W, 4, 3, dev)
        check(RT.lib().yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, 3, H, W, a.p, 4, 3, 3, st), "nchw->nhwc4")
        # stem: 7x7/s2 (+BN+ReLU) as the row-segment implicit GEMM, then MaxPool2d(3,2,1)
        wf, b, conv = pk["stem"]
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        s1 = self._act("stem", N, Ho, Wo, 64, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, Ho, Wo
        d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = a.img_stride, a.row_stride, a.px_stride, 0
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 2, 7, 1, 32, 64
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = s1.img_stride, s1.row_stride, s1.px_stride, s1.interior_off()
        d.epilogue, d.slope = EPI_BIAS_LRELU, 0.0
        if CFG.STEM_KERNEL and Ho % 8 == 0 and Wo % 16 == 0:
            with _timed("stem", "stem", 2.0 * N * Ho * Wo * 64 * 147):
                check(RT.lib().yolo_conv_stem7_fwd(a.p, ptr(wf), ptr(b), N, Ho, Wo, a.img_stride, a.row_stride, 0.0, 0, s1.p, s1.img_stride, s1.row_stride,
                                                s1.interior_off(), None, 0, 0, 0, st), "conv_stem7_fwd")
        else:
            with _timed("stem", "igemm", 2.0 * N * Ho * Wo * 64 * 147):
                _igemm(RT.lib(), d, a.p, ptr(wf), ptr(b), None, s1.p, st, "igemm stem")
        Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        cur = self._act("pool", N, Hq, Wq, 64, 1, dev)
        pd = PoolDesc(N, Ho, Wo, 64, 1, 1)
        check(RT.lib().yolo_maxpool3s2_fwd(ctypes.byref(pd), s1.p, cur.p, st), "maxpool3s2")
        for li in range(4, 8):
            for bi, blk in enumerate(self.trunk[li]):
                idn = cur if blk.downsample is None else self._conv((li, bi, "d"), cur, pk[(li, bi, "d")], N, False, None, dev, st)
                t = self._conv((li, bi, 1), cur, pk[(li, bi, 1)], N, True, None, dev, st)
                t = self._conv((li, bi, 2), t, pk[(li, bi, 2)], N, True, None, dev, st)
                cur = self._conv((li, bi, 3), t, pk[(li, bi, 3)], N, True, idn, dev, st)
        out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
        check(RT.lib().yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        return out
