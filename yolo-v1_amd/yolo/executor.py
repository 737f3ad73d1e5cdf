"""Layer-plan executor of the YOLOv1 networks: runs a conv / pool / FC stack of the reference's models on the HIP kernels.

A *plan* is built once from the PyTorch-layout modules (``nn.Conv2d``/``nn.LeakyReLU``/``nn.MaxPool2d`` inside ``backbone.features`` / ``head`` -- those
modules stay the owners of the fp32 parameters so that ``state_dict`` keys and shapes are the reference's).  On a device tensor the
modules' ``forward`` is bypassed and the plan drives libyolo_hip.so:

  * activations: zero-haloed NHWC bf16 buffers with a guard band (``runtime.Act``), allocated once per (batch, mode) and reused; producers only ever
    write the interior, so halos stay zero;
  * weights: bf16 panels re-packed from the fp32 masters only when a parameter's version changes;
  * forward = one yolo_igemm per conv / Linear (+ pool / flatten helpers), launched with the problem's plan (``plans.igemm_call``);
  * backward = per conv one yolo_wgrad (on a second stream) + one yolo_igemm data-gradient whose epilogue applies the previous LeakyReLU's
    derivative (or a pool backward).

Everything is enqueued on the current PyTorch stream (and the plan's side stream); there is no host synchronisation.  Switches: ``plan.cfg`` (an
``EngineConfig``) if set, else the process-wide ``config.CONFIG``."""

from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import _hip
from ._hip import (EPI_BIAS, EPI_BIAS_LRELU, EPI_MUL_DLRELU, EPI_NONE, ConvPackItem, ConvUnpackItem, IgemmDesc, PoolDesc, WgradDesc, check, ptr)
from .config import CONFIG as CFG
from .plans import igemm_call
from .runtime import RT, Act, _attach_wgrad_slabs, _EventSlot, _igemm, _on_side_stream, _round_up, _timed

@dataclass
class Layer:
    kind: str                      # conv | pool | flatten | fc
    name: str = ""
    Cout: int = 0
    Cin: int = 0
    K: int = 1
    stride: int = 1
    pad: int = 0
    lrelu: bool = False
    dropout: float = 0.0
    drop_mod: object = None     # the nn.Dropout behind a Linear: its p is read at every forward, as stock torch does (a plan outlives `model.head[3].p = 0.0`)
    weight: nn.Parameter | None = None
    bias: nn.Parameter | None = None
    first: bool = False            # the 3-channel 7x7/s2 stem (NHWC4 input, row-segment taps)
    # geometry, filled by Plan._shape
    Hin: int = 0
    Win: int = 0
    Hout: int = 0
    Wout: int = 0


class Plan:
    """Executable plan for [conv|pool]* [flatten fc*]? ."""

    SLOPE = 0.1

    def __init__(self, layers: list[Layer], in_channels: int, input_is_image: bool, S: int | None = None):
        self.layers = layers
        self.in_channels = in_channels
        self.input_is_image = input_is_image
        self.params: list[nn.Parameter] = []
        for L in layers:
            if L.kind in ("conv", "fc"):
                self.params += [L.weight, L.bias]
        self._pf: dict[int, tuple] = {}
        self._pd: dict[int, tuple] = {}
        self._pfb: dict[int, tuple] = {}
        self._pd2: dict[int, tuple] = {}
        self._ws: dict[tuple, list] = {}
        self.grad_norm_sq: dict = {}   # id(weight) -> ((data_ptr, shape) of the gradient, its version, device double |g|^2) left by the last backward pass
        self.debug_keep = False      # tests: True = keep the last workspace (activations + gradients) for inspection AND store the
                                     # un-pooled activations; "codes" = keep the workspace of the product path (pooled maps + arg-max codes)
        self.last = None
        # optional persistent gradient arena (data-parallel training): one flat fp32 buffer holding every
        # parameter gradient in the order backward PRODUCES them (last layer first), so that finished
        # gradients form a growing contiguous prefix that can be all-reduced while backward continues
        self.arena = None
        self.arena_views: dict[int, tuple] = {}
        self.on_grad_ready = None    # callback(lo, hi): arena[lo:hi] (elements) is final -- called with the PRODUCING stream current
        self.on_backward_done = None # callback(): every gradient is final and the current stream has waited for all of them
        self.on_stream_wait = None   # callback(waiter, waited): hipStream_t handles; `waiter` now waits for everything queued on `waited`
        self.cfg = None              # an EngineConfig of this plan's own (None: the process-wide config.CONFIG)
        self.params_ready = _EventSlot()      # event behind a background update of the Linear layers (forward waits in front of them)
        self.owner = None            # weakref to the nn.Module whose layers this plan runs (models.*.hip_plan sets it)

    @property
    def c(self):
        """the switches this plan runs with"""
        return self.cfg if self.cfg is not None else CFG

    def attach_grad_arena(self, device) -> torch.Tensor:
        """Allocate the gradient arena; backward then writes gradients into it, assigns ``p.grad`` views and
        returns no gradients to autograd (gradients are OVERWRITTEN each backward: no accumulation)."""
        order = [li for li in reversed(range(len(self.layers))) if self.layers[li].kind in ("conv", "fc")]
        off = 0
        wv, bv = {}, {}
        for li in order:                          # every view starts on a 256-B boundary (float4 kernels)
            n = self.layers[li].weight.numel()
            wv[li] = (off, off + n, _round_up(off + n, 64))
            off = _round_up(off + n, 64)
        self._arena_w_end = off
        for li in order:
            n = self.layers[li].bias.numel()
            bv[li] = (off, off + n)
            off = _round_up(off + n, 64)
        self.arena = torch.zeros(off, dtype=torch.float32, device=device)
        self.arena_views = {li: (self.arena[wv[li][0]:wv[li][1]].view_as(self.layers[li].weight),
                                 self.arena[bv[li][0]:bv[li][1]].view_as(self.layers[li].bias), wv[li][0], wv[li][2])
                            for li in order}
        return self.arena

    @staticmethod
    def _side_stream(dev) -> "torch.cuda.Stream":
        """the second stream of the backward pass (weight gradients): one per device, shared by all plans (kept outside the plan
        objects, which are deep-copied with their modules)"""
        key = torch.device(dev).index
        if key not in RT._SIDE_STREAMS:
            RT._SIDE_STREAMS[key] = RT.STREAMS.side(dev, CFG.SIDE_LOW)
        return RT._SIDE_STREAMS[key]

    def _layer_done(self, li: int):
        if self.arena is not None and self.on_grad_ready is not None:
            _, _, a, b = self.arena_views[li]
            self.on_grad_ready(a, b)

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def from_modules(mods, in_channels: int, input_is_image: bool) -> "Plan":
        """mods: flat list of nn.Module (Conv2d, LeakyReLU, MaxPool2d, Flatten, Linear, Dropout)."""
        layers: list[Layer] = []
        i = 0
        mods = list(mods)
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, nn.Conv2d):
                k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
                if m.kernel_size[0] != m.kernel_size[1] or m.groups != 1 or m.dilation != (1, 1) or m.bias is None:
                    raise ValueError(f"unsupported conv {m}")
                act = isinstance(nxt, nn.LeakyReLU)
                if act and abs(nxt.negative_slope - Plan.SLOPE) > 1e-12:
                    raise ValueError("only LeakyReLU(0.1) is fused")
                first = (m.in_channels == 3 and k == 7 and s == 2 and p == 3)
                if not first and not ((k == 3 and p == 1) or (k == 1 and p == 0)) or (not first and s not in (1, 2)):
                    raise ValueError(f"unsupported conv geometry {m}")
                if not first and (m.in_channels % 32 or m.out_channels % 8):
                    raise ValueError(f"unsupported channel counts {m}")
                layers.append(Layer("conv", Cout=m.out_channels, Cin=m.in_channels, K=k, stride=s, pad=p, lrelu=act,
                                    weight=m.weight, bias=m.bias, first=first))
                i += 2 if act else 1
            elif isinstance(m, nn.MaxPool2d):
                ks = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
                st = m.stride if isinstance(m.stride, int) else m.stride[0]
                if ks != 2 or st != 2:
                    raise ValueError("only MaxPool2d(2,2)")
                layers.append(Layer("pool"))
                i += 1
            elif isinstance(m, nn.Flatten):
                layers.append(Layer("flatten"))
                i += 1
            elif isinstance(m, nn.Linear):
                if m.in_features % 64 or m.bias is None:
                    raise ValueError(f"unsupported Linear {m}: in_features must be a multiple of 64")
                act = isinstance(nxt, nn.LeakyReLU)
                j = i + (2 if act else 1)
                drop, drop_mod = 0.0, None
                if j < len(mods) and isinstance(mods[j], nn.Dropout):
                    drop, drop_mod = mods[j].p, mods[j]
                    j += 1
                layers.append(Layer("fc", Cout=m.out_features, Cin=m.in_features, lrelu=act, dropout=drop, drop_mod=drop_mod, weight=m.weight, bias=m.bias))
                i = j
            else:
                raise ValueError(f"unsupported module in plan: {m}")
        return Plan(layers, in_channels, input_is_image)

    # ------------------------------------------------------------------ weights
    # bf16 operand copies of the fp32 masters, cached per layer and keyed on (tensor version, storage):
    #   _pf[li] = (key, forward operand)        conv: [Cout][KH][KW][Cin]      Linear: [O][K] (the master's layout)
    #   _pd[li] = (key, data-gradient operand)  conv: [Cin][KH][KW][Cout] flipped   Linear: [K][ld(O)] (only the
    #             small Linear layers; the big one in front of nn.Flatten uses the forward copy, see backward)
    @staticmethod
    def _wkey(w):
        return (w._version, w.data_ptr())

    def _multi_ok(self, L: Layer) -> bool:
        return L.kind == "conv" and not L.first and L.Cout % 64 == 0 and L.Cin % 64 == 0 and L.K * L.K <= 9

    def _src(self, L: Layer):
        wsrc = L.weight.detach()
        if wsrc.dtype != torch.float32 or not wsrc.is_contiguous():
            wsrc = wsrc.float().contiguous()
        return wsrc

    def _pack_all(self, need_dgrad: bool):
        """Refresh every stale conv operand of the plan in ONE launch (yolo_pack_conv_weights_multi)."""
        items, keep, done = [], [], []
        for li, L in enumerate(self.layers):
            if not self._multi_ok(L):
                continue
            key = self._wkey(L.weight)
            f, d = self._pf.get(li), self._pd.get(li)
            want_f = f is None or f[0] != key
            want_d = need_dgrad and li > 0 and (d is None or d[0] != key)
            if not (want_f or want_d):
                continue
            dev = L.weight.device
            wsrc = self._src(L)
            keep.append(wsrc)
            wf = wd = None
            if want_f:
                wf = f[1] if f is not None else torch.empty((L.Cout, L.K, L.K, L.Cin), dtype=torch.bfloat16, device=dev)
            if want_d:
                wd = d[1] if d is not None else torch.empty((L.Cin, L.K, L.K, L.Cout), dtype=torch.bfloat16, device=dev)
            items.append(ConvPackItem(wsrc.data_ptr(), wf.data_ptr() if wf is not None else None, wd.data_ptr() if wd is not None else None,
                                      L.Cout, L.Cin, L.K, L.K))
            done.append((li, key, wf, wd))
        if items:
            tab = (ConvPackItem * len(items))(*items)
            check(RT.lib().yolo_pack_conv_weights_multi(tab, len(items), RT.stream()), "pack_conv_weights_multi")
            for li, key, wf, wd in done:
                if wf is not None:
                    self._pf[li] = (key, wf)
                if wd is not None:
                    self._pd[li] = (key, wd)

    def _pack(self, li: int, need_dgrad: bool):
        """(forward operand, data-gradient operand | None) of layer li, refreshed if the master changed."""
        L = self.layers[li]
        key = self._wkey(L.weight)
        f, d = self._pf.get(li), self._pd.get(li)
        ok_f = f is not None and f[0] == key
        ok_d = d is not None and d[0] == key
        if ok_f and (ok_d or not need_dgrad):
            return f[1], (d[1] if ok_d else None)
        dev = L.weight.device
        wsrc = self._src(L)
        wf = f[1] if f is not None else None
        wd = d[1] if d is not None else None
        if L.kind == "conv":
            if L.first:
                if wf is None:
                    wf = torch.empty((L.Cout, 7, 8, 4), dtype=torch.bfloat16, device=dev)
                check(RT.lib().yolo_pack_conv_weight(ptr(wsrc), L.Cout, 3, 7, 7, 4, 8, ptr(wf), None, RT.stream()), "pack_conv_weight")
                self._pf[li] = (key, wf)
                return wf, None
            if wf is None:
                wf = torch.empty((L.Cout, L.K, L.K, L.Cin), dtype=torch.bfloat16, device=dev)
            if need_dgrad and wd is None:
                wd = torch.empty((L.Cin, L.K, L.K, L.Cout), dtype=torch.bfloat16, device=dev)
            check(RT.lib().yolo_pack_conv_weight(ptr(wsrc), L.Cout, L.Cin, L.K, L.K, L.Cin, L.K, None if ok_f else ptr(wf),
                                              ptr(wd) if (need_dgrad and not ok_d) else None, RT.stream()), "pack_conv_weight")
        else:
            if not ok_f:
                if wf is None:
                    wf = torch.empty((L.Cout, L.Cin), dtype=torch.bfloat16, device=dev)
                check(RT.lib().yolo_cast_f32_to_bf16(ptr(wsrc), wsrc.numel(), ptr(wf), RT.stream()), "cast fc weight")
            if need_dgrad and not ok_d:
                ld = _round_up(L.Cout, 32)
                if wd is None:
                    wd = torch.zeros((L.Cin, ld), dtype=torch.bfloat16, device=dev)
                check(RT.lib().yolo_transpose_f32_to_bf16(ptr(wsrc), L.Cout, L.Cin, ptr(wd), ld, RT.stream()), "transpose")
        self._pf[li] = (key, wf)
        if need_dgrad:
            self._pd[li] = (key, wd)
            return wf, wd
        return wf, (wd if ok_d else None)

    def _stride2_panels(self, li: int, wdg: torch.Tensor) -> dict:
        """data-gradient operands of a stride-2 3x3 conv by input-pixel parity: slices of the flipped panel
        wd[ci][ky'][kx'][co] (ky' = 2 - ky): parity 0 uses ky' = 1, parity 1 uses ky' = 0 (tap offset 0) and 2 (offset 1)."""
        L = self.layers[li]
        key = self._wkey(L.weight)
        hit = self._pd2.get(li)
        if hit is not None and hit[0] == key:
            return hit[1]
        sel = {0: slice(1, 2), 1: slice(0, 3, 2)}      # basic slices: one strided copy per class, no gather kernels
        panels = {}
        for py in (0, 1):
            for px in (0, 1):
                panels[(py, px)] = wdg[:, sel[py], sel[px], :].contiguous()
        self._pd2[li] = (key, panels)
        return panels

    def _pack_fc_blocked(self, li: int, hwc=None):
        """inference operand of a Linear layer: bf16 [O/128][K/64][128][64] panels (contiguous 16-KB stage reads; the
        plain [O][K] copy that training shares with the optimizer streams ~15 % slower).  ``hwc = (C, HW)``: K axis permuted from
        nn.Flatten's (c, hw) order to (hw, c) -- the layer then reads the dense NHWC conv output directly (self.c.FLATTEN_FREE)."""
        L = self.layers[li]
        key = (self._wkey(L.weight), hwc)
        hit = self._pfb.get(li)
        if hit is not None and hit[0] == key:
            return hit[1]
        wsrc = self._src(L)
        wb = hit[1] if hit is not None else torch.empty((_round_up(L.Cout, 128) * L.Cin,), dtype=torch.bfloat16, device=L.weight.device)
        if hwc is not None:
            check(RT.lib().yolo_pack_fc_weight_blocked_hwc(ptr(wsrc), L.Cout, hwc[0], hwc[1], ptr(wb), RT.stream()), "pack_fc_blocked_hwc")
        else:
            check(RT.lib().yolo_pack_fc_weight_blocked(ptr(wsrc), L.Cout, L.Cin, ptr(wb), RT.stream()), "pack_fc_blocked")
        self._pfb[li] = (key, wb)
        return wb

    def fc_biases(self):
        """bias parameters of the Linear layers on the device (updated together with their weights, yolo.optim.Adam.attach_plan)"""
        return [L.bias for L in self.layers if L.kind == "fc" and L.bias is not None and L.bias.is_cuda]

    def bf16_shadows(self):
        """[(param, bf16 forward operand with the master's layout, callback)] for the Linear layers: an optimizer
        that writes bf16(p) into the operand while it updates p calls ``callback(p)`` afterwards
        (yolo.optim.Adam.attach_plan), which saves the 822 MB + 411 MB re-cast of the big Linear per step."""
        out = []
        for li, L in enumerate(self.layers):
            if L.kind != "fc" or not L.weight.is_cuda:
                continue
            wf, _ = self._pack(li, False)

            def fresh(p, li=li, wf=wf):
                self._pf[li] = (self._wkey(p), wf)
            out.append((L.weight, wf, fresh))
        return out

    # ------------------------------------------------------------------ workspace
    def _workspace(self, N: int, x_shape, device, train: bool):
        key = (N, tuple(x_shape[1:]), str(device), train)
        pool = self._ws.setdefault(key, [])
        if pool:
            ws = pool.pop()
            self._apply_geom(ws)
            return key, ws
        ws = {"acts": [], "grads": {}, "misc": {}, "geom": {}}
        C, H, W = x_shape[1], x_shape[2], x_shape[3]
        if self.layers and self.layers[0].kind == "conv" and self.layers[0].first:
            a = Act(N, H, W, 4, 3, device)
        else:
            a = Act(N, H, W, C, 1, device)
        ws["in"] = a
        cur = a
        flat = None
        for li, L in enumerate(self.layers):
            if L.kind == "conv":
                L.Hin, L.Win = cur.H, cur.W
                L.Hout = (cur.H + 2 * L.pad - L.K) // L.stride + 1
                L.Wout = (cur.W + 2 * L.pad - L.K) // L.stride + 1
                # inference, conv -> nn.Flatten -> Linear: the conv writes a dense NHWC map (no halo) that the Linear layer reads as it
                # lies, through weight panels with a permuted K axis -- no flatten pass (self.c.FLATTEN_FREE)
                dense = (not train and self.c.FLATTEN_FREE and li + 2 < len(self.layers) and self.layers[li + 1].kind == "flatten"
                         and self.layers[li + 2].kind == "fc" and L.Cout % 8 == 0 and (L.Cout * L.Hout * L.Wout) % 64 == 0)
                cur = Act(N, L.Hout, L.Wout, L.Cout, 0 if dense else 1, device)
            elif L.kind == "pool":
                L.Hin, L.Win = cur.H, cur.W
                cur = Act(N, cur.H // 2, cur.W // 2, cur.C, 1, device)
            elif L.kind == "flatten":
                flat = torch.empty((N, cur.C * cur.H * cur.W), dtype=torch.bfloat16, device=device)
                cur = flat
            elif L.kind == "fc":
                feat = flat.shape[1] if (flat is not None and cur is flat) else None
                if feat is not None and feat != L.Cin:
                    # the reference raises here too (stock nn.Linear): e.g. a 224x224 batch into the 448x448 head
                    raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({N}x{feat} and {L.Cin}x{L.Cout}): input of "
                                       f"{tuple(x_shape[2:])} pixels does not match the Linear layer behind nn.Flatten")
                cur = None  # allocated per call (tiny)
            ws["acts"].append(cur)
            ws["geom"][li] = (L.Hin, L.Win, L.Hout, L.Wout)
        return key, ws

    def _apply_geom(self, ws):
        """the layer geometry (Hin, Win, Hout, Wout) belongs to a workspace, not to the plan: a plan may serve several input
        sizes, and a forward at another size may run between a training forward and its backward.  Every entry point that
        reads ``L.Hin`` .. ``L.Wout`` calls this first with the workspace it is about to use."""
        for li, g in ws["geom"].items():
            L = self.layers[li]
            L.Hin, L.Win, L.Hout, L.Wout = g

    def _release(self, key, ws):
        self._ws.setdefault(key, []).append(ws)

    # ------------------------------------------------------------------ descriptors
    def _conv_desc(self, L: Layer, a_in: Act, a_out: Act) -> IgemmDesc:
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = a_in.N, L.Hout, L.Wout
        d.in_img_stride, d.in_row_stride, d.in_px_stride = a_in.img_stride, a_in.row_stride, a_in.px_stride
        d.stride = L.stride
        if L.first:
            d.in_off = 0
            d.KH, d.KW, d.tap_len = 7, 1, 32
        else:
            d.in_off = a_in.interior_off(L.pad)
            d.KH, d.KW, d.tap_len = L.K, L.K, L.Cin
        d.Cout = L.Cout
        d.out_img_stride, d.out_row_stride, d.out_px_stride = a_out.img_stride, a_out.row_stride, a_out.px_stride
        d.out_off = a_out.interior_off()
        d.epilogue = EPI_BIAS_LRELU if L.lrelu else EPI_BIAS
        d.slope = self.SLOPE
        d.out_fp32 = 0
        d.split_k = 1
        d.tile_hint, d.tile_px = self.c.TILE_HINT, self.c.TILE_PX
        return d

    @staticmethod
    def _pool_fusable(L: Layer) -> bool:
        """conv -> LeakyReLU -> MaxPool2d(2,2) as one launch: the library's 8 x 16-patch pooled epilogue (224^2 and 112^2
        maps), or the pipelined kernels' 224-pixel tiles of whole row pairs (rows of 112, 56 or 28 pixels)"""
        if L.Cout % 8:
            return False
        if L.Hout % 8 == 0 and L.Wout % 16 == 0:
            return True
        nk = L.K * L.K * L.Cin // 32
        return (not L.first and L.Cin % 32 == 0 and nk % 2 == 0 and nk >= 4 and L.Wout in (112, 56, 28) and L.Hout % 2 == 0
                and (L.Hout * L.Wout) % 112 == 0)

    def _few_tiles(self, L: Layer, N: int) -> bool:
        """small batches: a deep-K conv in front of a pool whose pooled kernel would have work for less than a quarter of the chip (batch 1: conv18 = 4
        pixel tiles x 4 channel tiles under K = 4608, 83 us) runs un-fused -- K ranges as slabs over the whole chip (plans._default_plan / the tuner's
        slab candidates; the pooled epilogue has no split form) and the pool as its own small pass (config.SMALL_SPLIT)"""
        tiles = ((N * L.Hout * L.Wout + 223) // 224) * ((L.Cout + 255) // 256)
        return bool(self.c.SMALL_SPLIT and not L.first and tiles < 64 and L.K * L.K * L.Cin >= 2304 and L.Cin % 64 == 0)

    # ------------------------------------------------------------------ forward
    @_hip.device_guard
    def forward(self, x: torch.Tensor, train: bool, drop_training: bool, u8_size=None):
        """x: NCHW fp32 device tensor -- or, with ``u8_size = (H, W)``, decoded uint8 images [N][h][w][3] that
        yolo_preprocess_u8 resizes + normalises straight into the stem's NHWC4 input buffer (no fp32 NCHW round trip).
        Returns (out, saved) -- out is (N, O) fp32 if the plan ends with an fc layer, else NCHW fp32 features."""
        L_ = RT.lib()
        st = RT.stream()
        N = x.shape[0]
        dev = x.device
        x = x.detach()
        stem_f32 = False
        if u8_size is not None:
            from . import preprocess as _pp
            key, ws = self._workspace(N, (N, 3, u8_size[0], u8_size[1]), dev, train)
            self._pack_all(train)
            a = ws["in"]
            if not (a.C == 4 and a.halo == 3):
                raise ValueError("uint8 input needs a plan that starts with the 7x7/s2 stem")
            _pp.preprocess_u8_into(x, u8_size, a)
        else:
            if x.dim() != 4 or x.shape[1] != self.in_channels:
                raise RuntimeError(f"expected input of shape (N, {self.in_channels}, H, W), got {tuple(x.shape)}")
            if x.dtype != torch.float32 or not x.is_contiguous():
                x = x.float().contiguous()
            key, ws = self._workspace(N, x.shape, dev, train)
            self._pack_all(train)
            a = ws["in"]
            L0 = self.layers[0]
            # inference: the stem kernel reads the caller's NCHW fp32 batch itself (the patch is converted on its way into LDS); training
            # keeps the NHWC4 copy, which the stem's weight gradient reads
            stem_f32 = (not train and self.c.STEM_F32_INPUT and self.c.STEM_KERNEL and a.C == 4 and a.halo == 3 and L0.kind == "conv" and L0.first and L0.Cout == 64
                        and L0.Hout % 8 == 0 and L0.Wout % 16 == 0 and 2 * L0.Hout == x.shape[2] and 2 * L0.Wout == x.shape[3]
                        and ws["acts"][0].C == 64 and L0.bias.dtype == torch.float32)      # = the conditions of the stem-kernel branch below
            if stem_f32:
                pass
            elif a.C == 4 and a.halo == 3:
                check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, x.shape[1], x.shape[2], x.shape[3], a.p, 4, 3, 3, st), "nchw->nhwc4")
            else:
                check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(x), N, x.shape[1], x.shape[2], x.shape[3], a.p, a.C, 1, 1, st), "nchw->nhwc")
        cur = a
        fc_saved = {}
        out = None
        skip_pool = False
        # training, conv -> LeakyReLU -> MaxPool2d(2,2): the fused epilogue stores the pooled map and, per pooled element, the 2-bit
        # window position of the maximum; the backward pass needs nothing else of the un-pooled activation (debug_keep: the tests'
        # teacher-forced checks read that activation, so it is written instead)
        self.grad_norm_sq.clear()
        hwc = None
        codes_mode = train and self.c.POOL_CODES and self.debug_keep is not True      # (debug_keep = "codes": keep the workspace of the product path)
        ws["codes"] = set()
        for li, L in enumerate(self.layers):
            nxt = ws["acts"][li]
            if L.kind == "conv":
                wf, _ = self._pack(li, train)
                # inference: conv -> LeakyReLU -> MaxPool2d(2,2) as ONE launch when the conv output tiles into
                # 8 x 16 pixel patches (the first two layers: 224^2 and 112^2); training keeps the un-pooled
                # activation, which the backward pass needs
                fuse = (not train and self.c.FUSE_POOL and li + 1 < len(self.layers) and self.layers[li + 1].kind == "pool"
                        and self._pool_fusable(L) and not self._few_tiles(L, N))
                if fuse:
                    nxt = ws["acts"][li + 1]
                d = self._conv_desc(L, cur, nxt)
                d.pool2 = 1 if fuse else 0
                b = L.bias.detach()
                if L.first and self.c.STEM_KERNEL and L.Cout == 64 and L.Hout % 8 == 0 and L.Wout % 16 == 0 and nxt.C == 64 and b.dtype == torch.float32:
                    # dedicated stem kernel: input patch staged once per 8x16 tile, weights in registers; in training the
                    # following MaxPool2d is fused as well, with the un-pooled activation written next to the pooled one
                    dual = (train and self.c.FUSE_POOL and li + 1 < len(self.layers) and self.layers[li + 1].kind == "pool")
                    full = nxt if dual else None
                    dst = ws["acts"][li + 1] if dual else nxt
                    codes = self._codes(ws, li, dst) if (dual and codes_mode) else None
                    with _timed(f"conv{li}" + ("+pool" if (fuse or dual) else ""), "stem", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        if stem_f32 and li == 0:
                            check(L_.yolo_conv_stem7_fwd_f32(ptr(x), ptr(wf), ptr(b), N, x.shape[2], x.shape[3], self.SLOPE if L.lrelu else 1.0,
                                                             1 if fuse else 0, dst.p, dst.img_stride, dst.row_stride, dst.interior_off(), None, 0, 0, 0, st),
                                  "conv_stem7_fwd_f32")
                        elif codes is not None:       # pooled map + arg-max codes: the 411 MB un-pooled activation (batch 64) is never written
                            check(L_.yolo_conv_stem7_fwd(cur.p, ptr(wf), ptr(b), N, L.Hout, L.Wout, cur.img_stride, cur.row_stride,
                                                         self.SLOPE if L.lrelu else 1.0, 3, dst.p, dst.img_stride, dst.row_stride,
                                                         dst.interior_off(), ptr(codes), 0, 0, 0, st), "conv_stem7_fwd")
                        else:
                            check(L_.yolo_conv_stem7_fwd(cur.p, ptr(wf), ptr(b), N, L.Hout, L.Wout, cur.img_stride, cur.row_stride,
                                                         self.SLOPE if L.lrelu else 1.0, 1 if (fuse or dual) else 0, dst.p, dst.img_stride, dst.row_stride,
                                                         dst.interior_off(), full.p if dual else None, full.img_stride if dual else 0,
                                                         full.row_stride if dual else 0, full.interior_off() if dual else 0, st), "conv_stem7_fwd")
                    cur = dst
                    skip_pool = fuse or dual
                    continue
                # training: the same fused pool, with the un-pooled activation written too (pool2 = 2)
                dual = (train and self.c.FUSE_POOL and not fuse and li + 1 < len(self.layers) and self.layers[li + 1].kind == "pool"
                        and self._pool_fusable(L) and not L.first)
                if dual:
                    full, pooled = nxt, ws["acts"][li + 1]
                    d = self._conv_desc(L, cur, pooled)
                    codes = self._codes(ws, li, pooled) if codes_mode else None
                    if codes is not None:
                        d.pool2 = 3
                        auxp = ptr(codes)
                    else:
                        d.pool2 = 2
                        d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = full.img_stride, full.row_stride, full.px_stride, full.interior_off()
                        auxp = full.p
                    with _timed(f"conv{li}+pool", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        igemm_call(d, cur.p, ptr(wf), ptr(b), auxp, pooled.p, st, f"igemm conv{li}")
                    cur = pooled
                    skip_pool = True
                    continue
                with _timed(f"conv{li}" + ("+pool" if fuse else ""), "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                    igemm_call(d, cur.p, ptr(wf), ptr(b), None, nxt.p, st, f"igemm conv{li}")
                cur = nxt
                skip_pool = fuse
            elif L.kind == "pool" and skip_pool:
                skip_pool = False
            elif L.kind == "pool":
                pd = PoolDesc(N, cur.H, cur.W, cur.C, cur.halo, nxt.halo)
                with _timed(f"pool{li}", "maxpool2_fwd"):
                    check(L_.yolo_maxpool2_fwd(ctypes.byref(pd), cur.p, nxt.p, st), "maxpool")
                cur = nxt
            elif L.kind == "flatten":
                if not train and isinstance(cur, Act) and cur.halo == 0 and cur.halo_hi == 0 and self.c.FLATTEN_FREE:
                    hwc = (cur.C, cur.H * cur.W)            # the next Linear layer takes (hw, c)-ordered panels
                    cur = cur.t.view(N, -1)
                    continue
                check(L_.yolo_nhwc_bf16_to_nchw_bf16(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(nxt), st), "flatten")
                cur = nxt
            elif L.kind == "fc":
                self.params_ready.wait(dev)      # yolo.optim.Adam(overlap): the Linear layers' update of the last step runs on a second stream
                if train:
                    wf, _ = self._pack(li, False)
                else:
                    wf = self._pack_fc_blocked(li, hwc)
                    hwc = None
                xin = cur  # (N, K) bf16
                K = L.Cin
                d = IgemmDesc()
                d.N, d.Ho, d.Wo = N, 1, 1
                d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = xin.shape[1], 0, xin.shape[1], 0
                d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 1, 1, K, L.Cout
                d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = L.Cout, 0, L.Cout, 0
                d.slope = self.SLOPE
                d.out_fp32 = 1
                d.w_blocked = 0 if train else 1
                last = (li == len(self.layers) - 1)
                nk = K // 64
                # blocked panels (inference) run the 3-stage weight-stream kernel: 32 co-tiles x 32 splits = two full rounds of 512 slots
                splits = max(1, min(32 if d.w_blocked else 48, nk // 16)) if K >= 4096 else 1
                b = L.bias.detach()
                if splits > 1:
                    # every K split STORES its partial [N][Cout] result as a slab; the finishing pass adds the slabs in fixed
                    # order -> the forward is bit-reproducible (fp32 atomics of 32 splits were not) and needs no zero fill
                    acc = RT._splitk_scratch(splits * N * L.Cout, zero=False)
                    d.epilogue, d.split_k, d.split_slabs = EPI_NONE, splits, 1
                    with _timed(f"fc{li}", "igemm", 2.0 * N * L.Cout * L.Cin):
                        _igemm(L_, d, ptr(xin), ptr(wf), None, None, ptr(acc), st, f"igemm fc{li}")
                    yb = torch.empty((N, L.Cout), dtype=torch.bfloat16, device=dev) if not last else None
                    yf = torch.empty((N, L.Cout), dtype=torch.float32, device=dev) if last else None
                    check(L_.yolo_bias_lrelu_rows_slabs(ptr(acc), splits, ptr(b), N, L.Cout, self.SLOPE if L.lrelu else 1.0, ptr(yb), ptr(yf), st),
                          "bias_lrelu_rows")
                else:
                    yf = torch.empty((N, L.Cout), dtype=torch.float32, device=dev)
                    d.epilogue, d.split_k = (EPI_BIAS_LRELU if L.lrelu else EPI_BIAS), 1
                    with _timed(f"fc{li}", "igemm", 2.0 * N * L.Cout * L.Cin):
                        _igemm(L_, d, ptr(xin), ptr(wf), ptr(b), None, ptr(yf), st, f"igemm fc{li}")
                    yb = None
                    if not last:
                        yb = torch.empty((N, L.Cout), dtype=torch.bfloat16, device=dev)
                        check(L_.yolo_cast_f32_to_bf16(ptr(yf), yf.numel(), ptr(yb), st), "cast")
                mask = None
                y_act = yb
                if L.drop_mod is not None:
                    L.dropout = float(L.drop_mod.p)          # (the backward of this forward reads L.dropout)
                if not last and L.dropout > 0 and drop_training:
                    mask = (torch.rand((N, L.Cout), device=dev) >= L.dropout).to(torch.uint8)
                    yd = torch.empty_like(yb)
                    check(L_.yolo_dropout_bf16(ptr(yb), ptr(mask), 1.0 / max(1.0 - L.dropout, 1e-12) if L.dropout < 1.0 else 0.0, yb.numel(), ptr(yd), st), "dropout")
                    cur = yd
                else:
                    cur = yb
                fc_saved[li] = (xin, y_act, mask)
                if last:
                    out = yf
        if out is None:
            out = torch.empty((N, cur.C, cur.H, cur.W), dtype=torch.float32, device=dev)
            check(L_.yolo_nhwc_bf16_to_nchw_f32(cur.p, N, cur.C, cur.H, cur.W, cur.halo, ptr(out), st), "nhwc->nchw")
        saved = (key, ws, fc_saved, N, dev) if train else None
        if not train:
            self._release(key, ws)
        return out, saved

    # ------------------------------------------------------------------ backward
    @staticmethod
    def _codes(ws, li: int, pooled: Act) -> torch.Tensor:
        """arg-max codes of the pool behind conv layer li: uint16 per (pooled pixel, 8 channels), indexed like the pooled map / 8"""
        c = ws["misc"].get(("codes", li))
        if c is None:
            c = torch.empty(pooled.t.numel() // 8, dtype=torch.int16, device=pooled.t.device)
            ws["misc"][("codes", li)] = c
        ws["codes"].add(li)
        return c

    def _grad_buf(self, ws, li: int, N, dev) -> Act:
        """gradient wrt the (post-activation-derivative) output of conv layer li, in the geometry
        yolo_wgrad's flat indexing needs (= the layer's INPUT geometry; zero-stuffed for stride 2)."""
        g = ws["grads"].get(li)
        if g is None:
            L = self.layers[li]
            if L.stride == 1 or L.first:
                g = Act(N, L.Hout, L.Wout, L.Cout, 1, dev)
            else:
                g = Act(N, L.Hin, L.Win, L.Cout, 1, dev)
            ws["grads"][li] = g
        return g

    def _grad_out_strides(self, L: Layer, g: Act):
        """(img, row, px, off) strides a producer uses to write layer L's output gradient into g."""
        if L.stride == 1 or L.first:
            return g.img_stride, g.row_stride, g.px_stride, g.interior_off()
        return g.img_stride, 2 * g.row_stride, 2 * g.px_stride, g.interior_off()

    @staticmethod
    def _wgrad_desc(L: Layer, g: Act, xin: Act, N: int) -> WgradDesc:
        """yolo_wgrad problem of conv layer L over N images of the gradient buffer g / the input buffer xin"""
        # kernel variant: the 256 x 256 pipelined kernel (5) on the big deep layers, where the in-process A/B measured it 10-19 %
        # faster (56x56 256 -> 512, 28x28 512 -> 1024, 14x14 1024 -> 1024: tools/time_wgrad.py); the 128 x 128 kernel (0) elsewhere
        px = N * L.Hout * L.Wout
        deep = (L.Cout >= 512 and L.Cin >= 256 and px >= 40000) or (L.Cout >= 1024 and L.Cin >= 1024 and px >= 12000)
        variant = 5 if (CFG.WGRAD_PIPE and L.K == 3 and L.stride == 1 and deep) else 0
        flat = False
        if CFG.WGRAD_PIPE and CFG.WGRAD_WIDE and L.K == 3 and L.stride == 1:
            # variant 6 (tools/time_wgrad.py, each launch alone, same box): 56x56 256 -> 512 0.542 -> 0.484 ms, 28x28 512 -> 1024 0.499 -> 0.488,
            # 112x112 64 -> 192 (four taps per 256-column tile) 0.360 (128 x 128 kernel) -> 0.303; NOT on 14x14 1024 -> 1024 (0.297 vs 0.287).
            # From 56x56 up it reduces over every slot of the zero-haloed buffer (<= 7 % more pixels) and saves the pixel -> slot arithmetic.
            if (deep and L.Cin < 1024) or (L.Cin == 64 and L.Cout >= 192 and px >= 500000):
                variant = 6
                flat = L.Hout >= 56 and L.Wout >= 56
        geo_ok = L.Hout >= 2 and L.Wout >= 2
        geo = variant >= 5 or (geo_ok and (L.stride > 1 or g.Hp * g.Wp >= 1.12 * L.Hout * L.Wout))
        choice = CFG.WGRAD_CHOICE.get((N, L.Hout, L.Wout, L.Cout, L.Cin, L.K, L.stride)) if CFG.WGRAD_CHOICE else None
        if choice is not None and L.K == 3 and L.stride == 1 and CFG.WGRAD_PIPE and (int(choice[0]) != 6 or CFG.WGRAD_WIDE):
            # a choice measured inside the training step (tools/search_wgrad.py): which kernel, and whether it walks every slot of the zero-haloed buffer
            variant, flat = int(choice[0]), bool(choice[1])
            geo = geo_ok and not flat
        if not flat and geo:
            return WgradDesc(N * L.Hout * L.Wout, g.px_stride, xin.px_stride, L.Cout, L.Cin, L.K, L.K, L.pad, xin.row_stride, 0, 0, variant,
                             L.Wout, L.Hout, g.Hp * g.Wp, g.Wp * L.stride, L.stride, g.halo * g.Wp + g.halo)
        return WgradDesc(N * g.Hp * g.Wp, g.px_stride, xin.px_stride, L.Cout, L.Cin, L.K, L.K, L.pad, xin.row_stride, 0, 0, variant)

    def backward(self, saved, gout: torch.Tensor, need_gx: bool):
        """gout: gradient of the plan output (same shape as forward's out).  Returns (gx or None, [param grads])."""
        L_ = RT.lib()
        st = RT.stream()
        key, ws, fc_saved, N, dev = saved
        self._apply_geom(ws)
        if self.arena is not None:
            self.arena[self._arena_w_end:].zero_()      # bias gradients are accumulated with atomics
        grads: dict[int, tuple] = {}
        nl = len(self.layers)
        # two zero-filled fp32 scratch areas for the whole pass: the packed conv weight gradients (targets of yolo_wgrad's atomics; a buffer
        # of the workspace, cleared ON THE SIDE STREAM, where its first user runs: 80 MB of fill off the data-gradient chain) and,
        # without an arena, the bias gradients (fresh every pass: they are handed to autograd) -- two fills instead of ~50
        offs, tot = {}, 0
        for i, L in enumerate(self.layers):
            if L.kind == "conv":
                offs[("w", i)] = tot
                tot += _round_up(L.Cout * 7 * 8 * 4 if L.first else L.Cout * L.K * L.K * L.Cin, 64)
        btot = 0
        if self.arena is None:
            for i, L in enumerate(self.layers):
                if L.kind in ("conv", "fc"):
                    offs[("b", i)] = btot
                    btot += _round_up(L.Cout, 64)
        scratch = ws["misc"].get("wgrad_scratch")
        if scratch is None or scratch.numel() < tot:
            scratch = ws["misc"]["wgrad_scratch"] = torch.empty(max(tot, 1), dtype=torch.float32, device=dev)
        bscratch = torch.zeros(max(btot, 1), dtype=torch.float32, device=dev)

        def grad_tensors(i):
            L = self.layers[i]
            if self.arena is not None:
                dw, db, _, _ = self.arena_views[i]
                return dw, db
            o = offs[("b", i)]
            return torch.empty_like(L.weight, dtype=torch.float32), bscratch[o: o + L.Cout]

        # packed -> OIHW conversion of finished conv gradients is deferred and done for several layers per
        # launch (yolo_unpack_conv_wgrads_multi); gradients become final (and are announced) at the flush
        pending: list[tuple] = []
        stem_dpool = None            # pooled gradient handed straight to the stem's weight-gradient kernel (pool backward fused there)

        def flush():
            items = [ConvUnpackItem(dwp.data_ptr(), dw.data_ptr(), L.Cout, L.Cin, L.K, L.K) for (i, L, dwp, dw) in pending if self._multi_ok(L)]
            if items:
                check(L_.yolo_unpack_conv_wgrads_multi((ConvUnpackItem * len(items))(*items), len(items), RT.stream()), "unpack_conv_wgrads_multi")
            for (i, L, dwp, dw) in pending:
                if not self._multi_ok(L):
                    if L.first:
                        check(L_.yolo_unpack_conv_wgrad(ptr(dwp), L.Cout, 3, 7, 7, 4, 8, ptr(dw), 0, RT.stream()), "unpack")
                    else:
                        check(L_.yolo_unpack_conv_wgrad(ptr(dwp), L.Cout, L.Cin, L.K, L.K, L.Cin, L.K, ptr(dw), 0, RT.stream()), "unpack")
                self._layer_done(i)
            pending.clear()

        gout = gout.detach()
        if gout.dtype != torch.float32 or not gout.is_contiguous():
            gout = gout.float().contiguous()

        # The data gradients form the chain every later layer waits for; a layer's weight gradient only needs that layer's output
        # gradient and is first read by the optimizer.  The conv weight gradients (and their unpack passes / gradient-ready
        # callbacks) therefore go to a second stream: their atomic epilogues, partial last rounds and prologues -- phases in which a
        # kernel leaves the matrix cores idle -- overlap with the data-gradient kernels of the layers below, workgroup by workgroup.
        main_t = RT.STREAMS.current(dev)
        side_t = self._side_stream(dev) if self.c.WGRAD_STREAM else None

        def _on_side():
            return _on_side_stream(main_t, side_t, self.on_stream_wait if self.arena is not None else None)

        with _on_side():
            scratch.zero_()
        fc_keep: list = []

        # what each layer's input activation is
        def input_of(li):
            return ws["in"] if li == 0 else ws["acts"][li - 1]

        # g_cur: gradient flowing into the output of layer li (representation depends on kind)
        g_flat = None       # fp32 (N, K) gradient wrt an fc layer's output / flatten output
        g_act: Act | None = None   # Act gradient wrt a conv/pool output (already through LeakyReLU')
        li = nl - 1
        if self.layers[li].kind == "fc":
            g_flat = gout.reshape(N, -1)
        else:
            # plan ends with feature maps (NCHW fp32 gradient): last layer is a conv(+lrelu) or a pool
            L = self.layers[li]
            y = ws["acts"][li]
            graw = Act(N, y.H, y.W, y.C, 1, dev)
            check(L_.yolo_nchw_f32_to_nhwc_bf16(ptr(gout), N, y.C, y.H, y.W, graw.p, y.C, 1, 1, st), "gout->nhwc")
            if L.kind == "conv":
                g = self._grad_buf(ws, li, N, dev)
                self._apply_dlrelu_into(graw, y, L, g, st)
                g_act = g
            else:
                assert L.kind == "pool", "plans end with fc, conv or pool"
                g_act = graw

        while li >= 0:
            L = self.layers[li]
            if L.kind == "fc":
                xin, y_act, mask = fc_saved[li]
                last = (li == nl - 1)
                ldg = _round_up(L.Cout, 32)
                gb = torch.empty((N, ldg), dtype=torch.bfloat16, device=dev)
                # through dropout + LeakyReLU of THIS layer's output (none for the last layer)
                check(L_.yolo_scale_rows_to_bf16(ptr(g_flat), ptr(mask), ((1.0 / (1.0 - L.dropout)) if L.dropout < 1.0 else 0.0) if mask is not None else 1.0,
                                                 ptr(y_act) if (L.lrelu and not last) else None, self.SLOPE, N, L.Cout, ldg, ptr(gb), st), "scale_rows")
                # weight / bias gradient, native [O][K] layout
                dw, db = grad_tensors(li)
                wd = WgradDesc(N, ldg, L.Cin, L.Cout, L.Cin, 1, 1, 0, 0, 1, 0)
                nsq = None
                if L.Cout * L.Cin >= self.c.FC_NORM_IN_WGRAD and L.Cin % 4 == 0:
                    # the kernel that stores this gradient also sums its squares: the optimizer's global-norm pass (clip_grad_norm_) then
                    # need not read the 822 MB of the Linear behind nn.Flatten again (yolo.optim.grad_norm_sq, `known`)
                    nsq = torch.zeros((), dtype=torch.float64, device=dev)
                    wd.dw_sumsq = nsq.data_ptr()
                if self.c.FC_WGRAD_SIDE and side_t is not None:
                    # HBM-bound both: the weight gradient of the Linear behind nn.Flatten STORES 822 MB (4.1 TB/s alone), its data gradient READS
                    # the 411 MB of weights (2.7 TB/s alone); side by side they share the memory system instead of taking turns
                    fc_keep.append(gb)          # (a temporary of the main stream's allocator that the second stream reads: alive until the streams join)
                    with _on_side() as wst:
                        with _timed(f"fc{li}.wgrad", "wgrad", 2.0 * N * L.Cout * L.Cin):
                            check(L_.yolo_wgrad(ctypes.byref(wd), ptr(xin), ptr(gb), ptr(dw), ptr(db), wst), f"wgrad fc{li}")
                        self._layer_done(li)
                else:
                    with _timed(f"fc{li}.wgrad", "wgrad", 2.0 * N * L.Cout * L.Cin):
                        check(L_.yolo_wgrad(ctypes.byref(wd), ptr(xin), ptr(gb), ptr(dw), ptr(db), st), f"wgrad fc{li}")
                    self._layer_done(li)
                if nsq is not None:
                    # (no reference to dw itself: autograd takes the gradient over without a copy only while nobody else holds it)
                    self.grad_norm_sq[id(L.weight)] = ((dw.data_ptr(), tuple(dw.shape)), dw._version, nsq)
                grads[li] = (dw, db)
                # data gradient
                need_prev = li > 0 or need_gx
                behind_flatten = li >= 2 and self.layers[li - 1].kind == "flatten" and self.layers[li - 2].kind in ("conv", "pool")
                if need_prev and behind_flatten:
                    # the Linear behind nn.Flatten (205 M weights): reduce over the OUTPUT features with the
                    # weight-gradient kernel -- both operands are strided along the reduction axis there, which
                    # is exactly how W[o][k] and g^T[o][n] lie in memory -- and read the forward bf16 copy of W:
                    #   dxT[k][n] = sum_o W[o][k] * gT[o][n]
                    Lc = self.layers[li - 2]
                    y = ws["acts"][li - 2]
                    wf, _ = self._pack(li, False)
                    ldn = _round_up(N, 8)
                    gT = torch.zeros((L.Cout, ldn), dtype=torch.bfloat16, device=dev)
                    check(L_.yolo_transpose_bf16(ptr(gb), N, L.Cout, ldg, ptr(gT), ldn, st), "transpose g")
                    dxT = torch.zeros((L.Cin, N), dtype=torch.float32, device=dev)
                    wd = WgradDesc(L.Cout, L.Cin, ldn, L.Cin, N, 1, 1, 0, 0, 0, 1)   # split 0: library's schedule (0.095 vs 0.135 ms with 3 ranges)
                    with _timed(f"fc{li}.dgrad", "wgrad", 2.0 * N * L.Cout * L.Cin):
                        check(L_.yolo_wgrad(ctypes.byref(wd), ptr(gT), ptr(wf), ptr(dxT), None, st), f"dgrad fc{li}")
                    if Lc.kind == "conv":
                        assert Lc.stride == 1, "nn.Flatten is expected after a stride-1 conv or a pool"
                        g = self._grad_buf(ws, li - 2, N, dev)
                        yact = y.p if Lc.lrelu else None
                    else:
                        g = ws["misc"].get("graw_flat")
                        if g is None:
                            g = Act(N, y.H, y.W, y.C, 1, dev)
                            ws["misc"]["graw_flat"] = g
                        yact = None
                    check(L_.yolo_fc_dgrad_to_nhwc(ptr(dxT), N, y.C, y.H, y.W, 1, yact, self.SLOPE, g.p, st), "fc_dgrad_to_nhwc")
                    g_act = g
                    g_flat = None
                    li -= 2          # nn.Flatten is done as well
                    continue
                if need_prev:
                    _, wt = self._pack(li, True)
                    d = IgemmDesc()
                    d.N, d.Ho, d.Wo = N, 1, 1
                    d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = ldg, 0, ldg, 0
                    d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, 1, 1, ldg, L.Cin
                    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = L.Cin, 0, L.Cin, 0
                    d.epilogue, d.slope, d.out_fp32, d.split_k = EPI_NONE, self.SLOPE, 1, 1
                    gprev = torch.empty((N, L.Cin), dtype=torch.float32, device=dev)
                    with _timed(f"fc{li}.dgrad", "igemm", 2.0 * N * L.Cout * L.Cin):
                        _igemm(L_, d, ptr(gb), ptr(wt), None, None, ptr(gprev), st, f"dgrad fc{li}")
                    g_flat = gprev
                li -= 1
            elif L.kind == "flatten":
                raise AssertionError("nn.Flatten is handled together with the Linear layer behind it")
            elif L.kind == "pool":
                # g_act = gradient wrt the pooled output; produce gradient wrt the conv in front
                lc = li - 1
                Lc = self.layers[lc]
                assert Lc.kind == "conv" and Lc.lrelu, "MaxPool2d is expected right after conv+LeakyReLU"
                yfull = ws["acts"][lc]
                if (lc == 0 and Lc.first and self.c.STEM_POOL_BWD_FUSED and Lc.Cout == 64 and Lc.Hout % 8 == 0 and Lc.Wout % 16 == 0 and g_act.halo == 1
                        and not need_gx):      # (a gradient wrt the input image needs the stem's output gradient as a tensor)
                    # the stem's weight-gradient kernel rebuilds this pool's (+ LeakyReLU's) backward per tile from the
                    # activation and the pooled gradient: the 224x224x64 gradient buffer is never written or read
                    stem_dpool = g_act
                    li -= 1
                    continue
                g = self._grad_buf(ws, lc, N, dev)
                pd = PoolDesc(N, yfull.H, yfull.W, yfull.C, 1, 1)
                with _timed(f"pool{li}.bwd", "maxpool2_bwd"):
                    if lc in ws.get("codes", ()):
                        ypool = ws["acts"][li]
                        assert (ypool.Hp, ypool.Wp, ypool.C, ypool.halo) == (g_act.Hp, g_act.Wp, g_act.C, g_act.halo)
                        check(L_.yolo_maxpool2_bwd_codes(ctypes.byref(pd), ypool.p, ptr(ws["misc"][("codes", lc)]), g_act.p, self.SLOPE, g.p, st), "maxpool_bwd_codes")
                    else:
                        check(L_.yolo_maxpool2_bwd_lrelu(ctypes.byref(pd), yfull.p, g_act.p, self.SLOPE, g.p, st), "maxpool_bwd")
                g_act = g
                li -= 1
            elif L.kind == "conv":
                g = g_act  # dZ of this layer, flat-geometry buffer
                xin = input_of(li)
                # ---- weight + bias gradient
                dw, db = grad_tensors(li)
                with _on_side() as wst:
                    o = offs[("w", li)]
                    stem_direct = L.first and L.Cout == 64 and L.Hout % 8 == 0 and L.Wout % 16 == 0
                    if stem_direct:
                        part = ws["misc"].get("stem_part")
                        if part is None:
                            part = torch.empty((768 * 14400,), dtype=torch.float32, device=dev)
                            ws["misc"]["stem_part"] = part
                        with _timed(f"conv{li}.wgrad", "wgrad", 2.0 * N * L.Hout * L.Wout * L.Cout * 147):
                            if stem_dpool is not None and 0 in ws.get("codes", ()):
                                yp = ws["acts"][1]
                                assert (yp.Hp, yp.Wp, yp.C, yp.halo) == (stem_dpool.Hp, stem_dpool.Wp, stem_dpool.C, stem_dpool.halo)
                                check(L_.yolo_wgrad_stem7_codes(xin.p, yp.p, ptr(ws["misc"][("codes", 0)]), N, L.Hout, L.Wout, xin.img_stride, xin.row_stride,
                                                                stem_dpool.p, stem_dpool.img_stride, stem_dpool.row_stride, stem_dpool.interior_off(),
                                                                self.SLOPE if L.lrelu else 1.0, ptr(dw), ptr(db), ptr(part), part.numel(), wst), "wgrad_stem7_codes")
                            elif stem_dpool is not None:
                                yf = ws["acts"][0]
                                check(L_.yolo_wgrad_stem7_pooled(xin.p, yf.p, N, L.Hout, L.Wout, xin.img_stride, xin.row_stride, yf.img_stride, yf.row_stride,
                                                                 yf.interior_off(), stem_dpool.p, stem_dpool.img_stride, stem_dpool.row_stride,
                                                                 stem_dpool.interior_off(), self.SLOPE if L.lrelu else 1.0, ptr(dw), ptr(db), ptr(part),
                                                                 part.numel(), wst), "wgrad_stem7_pooled")
                            else:
                                check(L_.yolo_wgrad_stem7(xin.p, g.p, N, L.Hout, L.Wout, xin.img_stride, xin.row_stride, g.img_stride, g.row_stride,
                                                          g.interior_off(), ptr(dw), ptr(db), ptr(part), part.numel(), wst), "wgrad_stem7")
                        grads[li] = (dw, db)
                        flush()
                        self._layer_done(li)
                    elif L.first:
                        xcol = ws["misc"].get("xcol")
                        if xcol is None:
                            xcol = Act(N, L.Hout, L.Wout, 7 * 32, 1, dev)
                            ws["misc"]["xcol"] = xcol
                        check(L_.yolo_im2col_rows(xin.p, xin.img_stride, xin.row_stride, xin.px_stride, 2, 7, 32, N, L.Hout, L.Wout, 1, xcol.p, wst), "im2col_rows")
                        dwp = scratch[o: o + L.Cout * 7 * 8 * 4]
                        split = max(1, min(1024, g.slots // 4096))
                        wd = WgradDesc(g.slots, g.px_stride, xcol.px_stride, L.Cout, 7 * 32, 1, 1, 0, xcol.row_stride, split, 0)
                        with _timed(f"conv{li}.wgrad", "wgrad", 2.0 * N * L.Hout * L.Wout * L.Cout * 147):
                            check(L_.yolo_wgrad(ctypes.byref(wd), xcol.p, g.p, ptr(dwp), ptr(db), wst), "wgrad conv0")
                    else:
                        dwp = scratch[o: o + L.Cout * L.K * L.K * L.Cin]
                        # reduce over the layer's OUTPUT pixels only (not over every slot of the zero-haloed -- for stride 2
                        # zero-stuffed -- gradient buffer, whose geometry the input buffer shares slot for slot)
                        # (measured: worth it from 28x28 down and for stride 2; at 56x56 and above the halo is < 8 % of the slots
                        # and the per-row coordinate arithmetic costs more than it saves)
                        wd = self._wgrad_desc(L, g, xin, N)
                        if self.c.WGRAD_SLABS and wd.variant >= 5:
                            _attach_wgrad_slabs(L_, wd, dev)
                        with _timed(f"conv{li}.wgrad", "wgrad", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                            check(L_.yolo_wgrad(ctypes.byref(wd), xin.p, g.p, ptr(dwp), ptr(db), wst), f"wgrad conv{li}")
                    if not stem_direct:
                        grads[li] = (dw, db)
                        pending.append((li, L, dwp, dw))
                        if li == 0 or sum(t[2].numel() for t in pending) >= (16 << 20):
                            flush()
                # ---- data gradient
                if li == 0:
                    if side_t is not None:
                        main_t.wait_stream(side_t)       # every weight gradient is final before anything that follows the backward pass
                        if self.arena is not None and self.on_stream_wait is not None:
                            self.on_stream_wait(main_t.cuda_stream, side_t.cuda_stream)
                    gx = None
                    if need_gx:
                        gx = self._stem_dgrad(li, g, N, dev, st) if L.first else self._dgrad_to_input(li, g, N, dev, st)
                    if self.debug_keep:
                        self.last = (ws, fc_saved)
                    else:
                        self._release(key, ws)
                    if self.arena is not None:
                        if self.on_backward_done is not None:
                            self.on_backward_done()
                        for i in grads:          # hand the views to the optimizer without going through autograd
                            L2 = self.layers[i]
                            if L2.weight.grad is not grads[i][0]:
                                L2.weight.grad = grads[i][0]
                            if L2.bias.grad is not grads[i][1]:
                                L2.bias.grad = grads[i][1]
                        return gx, [None] * (2 * len(grads))
                    return gx, [grads[i][j] for i in sorted(grads) for j in (0, 1)]
                _, wdg = self._pack(li, True)
                prev = self.layers[li - 1]
                d = IgemmDesc()
                d.N, d.Ho, d.Wo = N, L.Hin, L.Win            # gradient grid = this layer's input grid
                d.in_img_stride, d.in_row_stride, d.in_px_stride = g.img_stride, g.row_stride, g.px_stride
                d.in_off = g.interior_off(L.K - 1 - L.pad)
                d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, L.K, L.K, L.Cout, L.Cin
                d.slope, d.out_fp32, d.split_k = self.SLOPE, 0, 1
                d.tile_hint = self.c.TILE_HINT
                if (prev.kind == "conv" and self.c.STRIDE2_CLASSES and L.stride == 2 and L.K == 3 and L.pad == 1 and prev.stride == 1
                        and L.Hin % 2 == 0 and L.Win % 2 == 0):
                    # stride-2 3x3 conv: the gradient buffer g holds dy zero-stuffed to the input grid, and the plain data gradient
                    # spends 3/4 of its MACs on those zeros.  By input-pixel parity (py, px) only the taps ky = 1 (py even) or
                    # ky = 2, 0 (py odd; likewise kx) contribute: four small convs over the NON-ZERO slots (doubled input strides)
                    # with 1, 2, 2 and 4 taps -- 9 taps per 2x2 input pixels instead of 36 -- each writing its parity class of the
                    # previous layer's gradient (doubled output strides).
                    gp = self._grad_buf(ws, li - 1, N, dev)
                    yprev = ws["acts"][li - 1]
                    panels = self._stride2_panels(li, wdg)
                    with _timed(f"conv{li}.dgrad", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        for (py, px), wc in panels.items():
                            dc = IgemmDesc()
                            dc.N, dc.Ho, dc.Wo = N, L.Hin // 2, L.Win // 2
                            dc.in_img_stride, dc.in_row_stride, dc.in_px_stride, dc.in_off = g.img_stride, 2 * g.row_stride, 2 * g.px_stride, g.interior_off()
                            dc.stride, dc.KH, dc.KW, dc.tap_len, dc.Cout = 1, 1 + py, 1 + px, L.Cout, L.Cin
                            dc.slope, dc.out_fp32, dc.split_k, dc.tile_hint = self.SLOPE, 0, 1, self.c.TILE_HINT
                            dc.out_img_stride, dc.out_row_stride, dc.out_px_stride = gp.img_stride, 2 * gp.row_stride, 2 * gp.px_stride
                            dc.out_off = gp.interior_off() + py * gp.row_stride + px * gp.px_stride
                            aux = None
                            dc.epilogue = EPI_NONE
                            if prev.lrelu:
                                dc.epilogue = EPI_MUL_DLRELU
                                dc.aux_img_stride, dc.aux_row_stride, dc.aux_px_stride = yprev.img_stride, 2 * yprev.row_stride, 2 * yprev.px_stride
                                dc.aux_off = yprev.interior_off() + py * yprev.row_stride + px * yprev.px_stride
                                aux = yprev.p
                            igemm_call(dc, g.p, ptr(wc), None, aux, gp.p, st, f"dgrad conv{li} class {py}{px}")
                    g_act = gp
                elif prev.kind == "conv":
                    gp = self._grad_buf(ws, li - 1, N, dev)
                    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = self._grad_out_strides(prev, gp)
                    yprev = ws["acts"][li - 1]
                    if prev.lrelu:
                        d.epilogue = EPI_MUL_DLRELU
                        d.aux_img_stride, d.aux_row_stride, d.aux_px_stride, d.aux_off = yprev.img_stride, yprev.row_stride, yprev.px_stride, yprev.interior_off()
                        aux = yprev.p
                    else:
                        d.epilogue, aux = EPI_NONE, None
                    with _timed(f"conv{li}.dgrad", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        igemm_call(d, g.p, ptr(wdg), None, aux, gp.p, st, f"dgrad conv{li}")
                    g_act = gp
                elif prev.kind == "pool":
                    gp = ws["misc"].get(("gpool", li))
                    if gp is None:
                        gp = Act(N, L.Hin, L.Win, L.Cin, 1, dev)
                        ws["misc"][("gpool", li)] = gp
                    d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = gp.img_stride, gp.row_stride, gp.px_stride, gp.interior_off()
                    d.epilogue = EPI_NONE
                    with _timed(f"conv{li}.dgrad", "igemm", 2.0 * N * L.Hout * L.Wout * L.Cout * L.Cin * L.K * L.K):
                        igemm_call(d, g.p, ptr(wdg), None, None, gp.p, st, f"dgrad conv{li}")
                    g_act = gp
                else:
                    raise AssertionError("conv after flatten/fc")
                li -= 1
        raise AssertionError("unreachable")

    def _apply_dlrelu_into(self, graw: Act, y: Act, L: Layer, g: Act, st):
        """g(interior, possibly zero-stuffed) = graw * lrelu'(y) -- used only at plan ends (rare path)."""
        gi = graw.interior().float()
        if L.lrelu:
            gi = gi * torch.where(y.interior().float() > 0, 1.0, self.SLOPE)
        gi = gi.to(torch.bfloat16)
        if L.stride == 1 or L.first:
            g.interior().copy_(gi)
        else:
            g.interior()[:, 0::2, 0::2, :][:, : gi.shape[1], : gi.shape[2], :].copy_(gi)

    def _stem_dgrad(self, li, g: Act, N, dev, st):
        """gradient wrt the input IMAGE through the 7x7 / stride-2 / pad-3 stem (the reference back-propagates to x in
        tests/test_backbone.py:187-196; training never asks for it, so the step's FLOP count skips this product).  By the parity
        (py, px) of the image pixel (y, x) = (2a + py, 2b + px) only the taps ky = py + 5 - 2 ty (ty = 0 .. 2 + py; likewise kx) meet an
        output pixel, (a + ty - 1, b + tx - 1): four stride-1 correlations over the stem's output gradient g with 3x3, 3x4, 4x3 and 4x4
        taps of 64 channels, each writing its parity class of the image (doubled output strides) -- the scheme of the stride-2 3x3
        layers' data gradient.  The three image channels ride in an 8-channel fp32 NHWC scratch; rows / columns a + 2 past the map fall
        on the zero halo of the next row / image (or the guard band)."""
        L = self.layers[li]
        assert L.first and L.K == 7 and L.stride == 2 and L.pad == 3 and g.halo == 1 and g.C == L.Cout and L.Cout % 64 == 0
        H, W = 2 * L.Hout, 2 * L.Wout
        w = L.weight.detach().float()                                   # [Cout][3][7][7]
        buf = torch.empty((N, H, W, 8), dtype=torch.float32, device=dev)
        L_ = RT.lib()
        for py in (0, 1):
            for px in (0, 1):
                kys = [py + 5 - 2 * t for t in range(3 + py)]
                kxs = [px + 5 - 2 * t for t in range(3 + px)]
                panel = torch.zeros((8, len(kys), len(kxs), L.Cout), dtype=torch.bfloat16, device=dev)
                panel[:3] = w[:, :, kys][:, :, :, kxs].permute(1, 2, 3, 0).to(torch.bfloat16)      # [c][ty][tx][co]
                d = IgemmDesc()
                d.N, d.Ho, d.Wo = N, L.Hout, L.Wout
                d.in_img_stride, d.in_row_stride, d.in_px_stride, d.in_off = g.img_stride, g.row_stride, g.px_stride, g.interior_off(1)
                d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, len(kys), len(kxs), L.Cout, 8
                d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = H * W * 8, 2 * W * 8, 16, (py * W + px) * 8
                d.epilogue, d.slope, d.out_fp32, d.split_k, d.tile_hint = EPI_NONE, self.SLOPE, 1, 1, 4      # 64 x 128 tiles: 8 "channels"
                _igemm(L_, d, g.p, ptr(panel), None, None, ptr(buf), st, f"stem dgrad class {py}{px}")
        return buf[..., :3].permute(0, 3, 1, 2).contiguous()

    def _dgrad_to_input(self, li, g: Act, N, dev, st):
        """data gradient of the first conv of a plan whose input is a feature map (DetectionHead)."""
        L = self.layers[li]
        _, wdg = self._pack(li, True)
        gi = Act(N, L.Hin, L.Win, L.Cin, 1, dev)
        d = IgemmDesc()
        d.N, d.Ho, d.Wo = N, L.Hin, L.Win
        d.in_img_stride, d.in_row_stride, d.in_px_stride = g.img_stride, g.row_stride, g.px_stride
        d.in_off = g.interior_off(L.K - 1 - L.pad)
        d.stride, d.KH, d.KW, d.tap_len, d.Cout = 1, L.K, L.K, L.Cout, L.Cin
        d.out_img_stride, d.out_row_stride, d.out_px_stride, d.out_off = gi.img_stride, gi.row_stride, gi.px_stride, gi.interior_off()
        d.epilogue, d.slope, d.out_fp32, d.split_k = EPI_NONE, self.SLOPE, 0, 1
        _igemm(RT.lib(), d, g.p, ptr(wdg), None, None, gi.p, st, "dgrad input")
        gx = torch.empty((N, L.Cin, L.Hin, L.Win), dtype=torch.float32, device=dev)
        check(RT.lib().yolo_nhwc_bf16_to_nchw_f32(gi.p, N, L.Cin, L.Hin, L.Win, 1, ptr(gx), st), "gx nhwc->nchw")
        return gx

