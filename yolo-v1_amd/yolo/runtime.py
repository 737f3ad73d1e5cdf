"""Workspaces and streams of the engine: zero-haloed activation buffers (``Act``), scratch pools, the second stream of the backward pass and the
event slot of a background optimizer pass, the per-launch timers -- and ``RT``, the one place the executors get the library, the current stream and the
stream factory from (tests of the stream schedule put recording stand-ins there: ``engine.lib = fake`` forwards to ``RT.lib``)."""

from __future__ import annotations

import ctypes

import torch

from . import _hip
from ._hip import WgradDesc, check
from .config import CONFIG as CFG


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def _igemm(L_, d, inp, w, bias, aux, out, st, what):
    CFG.IGEMM_LAUNCHES += 1
    check(L_.yolo_igemm(ctypes.byref(d), inp, w, bias, aux, out, st), what)


class _Streams:
    """where the executors get their streams from (tests of the stream schedule put recording stand-ins here)"""

    @staticmethod
    def current(dev):
        return torch.cuda.current_stream(dev)

    @staticmethod
    def side(dev, low):
        return _hip.side_stream(torch.device(dev), low=low)

    @staticmethod
    def use(s):
        return torch.cuda.stream(s)




class _on_side_stream:
    """``with _on_side_stream(main, side) as st:`` -- work issued inside goes to ``side`` (None: stays on ``main``), behind everything
    queued on ``main`` so far; ``st`` is the hipStream_t to launch on.  The caller joins with ``main.wait_stream(side)``.
    ``note(waiter, waited)``: told about the wait (the gradient reducer keeps track of which stream has seen which)."""

    def __init__(self, main_t, side_t, note=None):
        self.main_t, self.side_t, self.note = main_t, side_t, note

    def __enter__(self):
        if self.side_t is None:
            return ctypes.c_void_p(self.main_t.cuda_stream)
        self.side_t.wait_stream(self.main_t)
        if self.note is not None:
            self.note(self.side_t.cuda_stream, self.main_t.cuda_stream)
        self.ctx = RT.STREAMS.use(self.side_t)
        self.ctx.__enter__()
        return ctypes.c_void_p(self.side_t.cuda_stream)

    def __exit__(self, *exc):
        if self.side_t is not None:
            self.ctx.__exit__(*exc)
        return False


class _timed:
    def __init__(self, tag: str, kernel: str, flops: float = 0.0):
        self.tag, self.kernel, self.flops = tag, kernel, flops

    def __enter__(self):
        if CFG.TIMERS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if CFG.TIMERS is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            CFG.TIMERS.append((self.tag, self.kernel, self.flops, self.e0, e1))
        return False


_SPLITK_SCRATCH: dict = {}


def _splitk_scratch(n: int, zero: bool) -> torch.Tensor:
    """fp32 scratch of n elements on the current device for a split-K launch (allocated once per device and grown on
    demand; the atomics form needs it zero-filled, the slab form does not)"""
    dev = torch.cuda.current_device()
    buf = _SPLITK_SCRATCH.get(dev)
    if buf is None or buf.numel() < n:
        buf = _SPLITK_SCRATCH[dev] = torch.empty(n, dtype=torch.float32, device=torch.device("cuda", dev))
    v = buf[:n]
    if zero:
        v.zero_()
    return v


_WGRAD_SLAB_BUF: dict = {}


def _attach_wgrad_slabs(L_, wd: WgradDesc, dev) -> None:
    """slab mode of the pipelined weight-gradient kernel (yolo_wgrad_desc.slabs): partial tiles as plain stores + a fixed-order sum
    instead of fp32 atomics.  One scratch per device, grown on demand (the launches of one stream use it one after the other)."""
    need = ctypes.c_long(0)
    check(L_.yolo_wgrad_slab_floats(ctypes.byref(wd), ctypes.byref(need)), "wgrad_slab_floats")
    if need.value <= 0:
        return
    key = torch.device(dev).index
    buf = _WGRAD_SLAB_BUF.get(key)
    if buf is None or buf.numel() < need.value:
        buf = _WGRAD_SLAB_BUF[key] = torch.empty(need.value, dtype=torch.float32, device=dev)
    wd.slabs, wd.slab_floats = buf.data_ptr(), buf.numel()


class _EventSlot:
    """holder of the event behind a background optimizer launch (yolo.optim.Adam.attach_plan(overlap=True)).  It lives on the plan
    object (not in a table keyed by id(plan), which outlives garbage collection); a deep copy of a plan starts with an empty slot --
    events do not copy, and the copy's parameters are new tensors nobody updates in the background."""

    def __init__(self):
        self.event = None

    def __deepcopy__(self, memo):
        return _EventSlot()

    def __reduce__(self):
        return (_EventSlot, ())

    def wait(self, dev=None, keep: bool = False):
        """the current stream waits for the pending update; ``keep``: leave the event in place for later readers on other streams"""
        ev = self.event
        if not keep:
            self.event = None
        if ev is not None:
            torch.cuda.current_stream(dev).wait_event(ev)


class Act:
    """Zero-haloed NHWC bf16 activation: [N][H+2h][W+2h][C] plus guard bands of zeros."""

    def __init__(self, N, H, W, C, halo, device, halo_hi=None):
        self.N, self.H, self.W, self.C = N, H, W, C
        self.halo = halo
        self.halo_hi = halo if halo_hi is None else halo_hi
        self.Hp = H + self.halo + self.halo_hi
        self.Wp = W + self.halo + self.halo_hi
        self.px_stride = C
        self.row_stride = self.Wp * C
        self.img_stride = self.Hp * self.Wp * C
        self.slots = N * self.Hp * self.Wp
        guard = _round_up((self.Wp + 2) * C + 64 * 8, 128)
        self.store = torch.zeros(guard + self.slots * C + guard, dtype=torch.bfloat16, device=device)
        self.t = self.store[guard: guard + self.slots * C]

    @property
    def p(self):
        return ctypes.c_void_p(self.t.data_ptr())

    def interior_off(self, shift=0):
        """element offset of logical pixel (-shift, -shift) inside an image"""
        h = self.halo - shift
        return (h * self.Wp + h) * self.C

    def view(self):
        return self.t.view(self.N, self.Hp, self.Wp, self.C)

    def interior(self):
        h = self.halo
        return self.view()[:, h: h + self.H, h: h + self.W, :]




class _Runtime:
    """patchable hooks: the library, the current HIP stream, the stream factory, the split-K scratch pool, the side streams per device"""

    def __init__(self):
        self.lib = _hip.lib
        self.stream = _hip.stream
        self.STREAMS = _Streams()
        self._SIDE_STREAMS: dict = {}
        self._splitk_scratch = _splitk_scratch


RT = _Runtime()
HOOKS = frozenset(("lib", "stream", "STREAMS", "_SIDE_STREAMS", "_splitk_scratch"))
