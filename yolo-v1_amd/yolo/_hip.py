"""ctypes binding of libyolo_hip.so (include/yolo_hip.h) -- the only way Python reaches the GPU kernels.

The library is hand-written HIP for gfx950, built in-tree by ``make -C yolo-v1_amd/csrc`` (or
``__graft_entry__.build()``).  There is NO fallback: if a CUDA/ROCm tensor reaches a module of this
package and the library is missing, :func:`lib` raises.
"""

from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO_HIP_LIB: another build of the same library (tools: the stamped diagnostic build libyolo_hip_diag.so)
LIB_PATH = os.environ.get("YOLO_HIP_LIB") or os.path.join(_HERE, "libyolo_hip.so")
_LIB = None

ABI_VERSION = 2              # YOLO_HIP_ABI_VERSION this binding was written against
BN_ACC_REPLICAS = 16         # YOLO_BN_ACC_REPLICAS (include/yolo_hip.h)
c_int, c_long, c_float, c_double, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class IgemmDesc(ctypes.Structure):
    """struct yolo_igemm_desc (include/yolo_hip.h)."""

    _fields_ = [
        ("N", ctypes.c_int32), ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
        ("in_img_stride", ctypes.c_int64),
        ("in_row_stride", ctypes.c_int32), ("in_px_stride", ctypes.c_int32), ("in_off", ctypes.c_int32),
        ("stride", ctypes.c_int32), ("KH", ctypes.c_int32), ("KW", ctypes.c_int32),
        ("tap_len", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("out_img_stride", ctypes.c_int64),
        ("out_row_stride", ctypes.c_int32), ("out_px_stride", ctypes.c_int32), ("out_off", ctypes.c_int32),
        ("epilogue", ctypes.c_int32), ("slope", ctypes.c_float), ("out_fp32", ctypes.c_int32), ("split_k", ctypes.c_int32),
        ("aux_img_stride", ctypes.c_int64),
        ("aux_row_stride", ctypes.c_int32), ("aux_px_stride", ctypes.c_int32), ("aux_off", ctypes.c_int32),
        ("pool2", ctypes.c_int32),
        ("w_blocked", ctypes.c_int32),
        ("tile_order", ctypes.c_int32),
        ("tile_hint", ctypes.c_int32),
        ("px_begin", ctypes.c_int64), ("px_end", ctypes.c_int64),
        ("skew_phases", ctypes.c_int32), ("skew_step", ctypes.c_int32),
        ("bn_stats", ctypes.c_void_p),
        ("tile_px", ctypes.c_int32), ("split_slabs", ctypes.c_int32),
    ]


class WgradDesc(ctypes.Structure):
    """struct yolo_wgrad_desc (include/yolo_hip.h)."""

    _fields_ = [
        ("P", ctypes.c_int64),
        ("dy_px_stride", ctypes.c_int32), ("x_px_stride", ctypes.c_int32),
        ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("KH", ctypes.c_int32), ("KW", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("x_row_stride", ctypes.c_int64),
        ("split", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("variant", ctypes.c_int32),
        ("geo_W", ctypes.c_int32), ("geo_H", ctypes.c_int32), ("geo_img_slots", ctypes.c_int32), ("geo_row_slots", ctypes.c_int32),
        ("geo_px_slots", ctypes.c_int32), ("geo_slot0", ctypes.c_int32),
        ("dw_sumsq", ctypes.c_void_p),
        ("slabs", ctypes.c_void_p), ("slab_floats", ctypes.c_int64),
    ]


class PoolDesc(ctypes.Structure):
    """struct yolo_pool_desc (include/yolo_hip.h)."""

    _fields_ = [("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
                ("in_halo", ctypes.c_int32), ("out_halo", ctypes.c_int32)]


class AdamTensor(ctypes.Structure):
    """struct yolo_adam_tensor (include/yolo_hip.h)."""

    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("p_bf16", c_void_p), ("n", ctypes.c_long)]


class ConvPackItem(ctypes.Structure):
    """struct yolo_conv_pack_item (include/yolo_hip.h)."""

    _fields_ = [("w", c_void_p), ("wf", c_void_p), ("wd", c_void_p), ("Cout", c_int), ("Cin", c_int), ("KH", c_int), ("KW", c_int)]


class ConvUnpackItem(ctypes.Structure):
    """struct yolo_conv_unpack_item (include/yolo_hip.h)."""

    _fields_ = [("dwp", c_void_p), ("dw", c_void_p), ("Cout", c_int), ("Cin", c_int), ("KH", c_int), ("KW", c_int)]


EPI_NONE, EPI_BIAS, EPI_BIAS_LRELU, EPI_MUL_DLRELU, EPI_BIAS_ADD_LRELU = 0, 1, 2, 3, 4
NMS_INFERENCE, NMS_METRICS = 0, 1

# name -> argtypes ; every symbol include/yolo_hip.h declares (tests check the list against the header)
_SIGS = {
    "yolo_hip_abi_version": [],
    "yolo_debug_stamps": [c_void_p, c_int],
    "yolo_stream_create": [c_int, ctypes.POINTER(c_void_p)],
    "yolo_decode": [c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p],
    "yolo_decode_gt": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "yolo_nms": [c_void_p, c_void_p, c_int, c_int, c_double, c_int, c_void_p, c_void_p, c_void_p],
    "yolo_batchnorm_train_fwd": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p, c_void_p, c_int,
                                 c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "yolo_batchnorm_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_long, ctypes.c_long,
                           ctypes.c_long, ctypes.c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "yolo_map_match": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_double), c_int, c_double, c_double, c_double,
                       c_void_p, c_void_p, c_void_p],
    "yolo_pairwise_iou": [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "yolo_loss_fwd_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "yolo_loss_iou": [c_void_p, c_void_p, c_long, c_void_p, c_void_p],
    "yolo_igemm": [ctypes.POINTER(IgemmDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "yolo_igemm_finish": [ctypes.POINTER(IgemmDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "yolo_wgrad": [ctypes.POINTER(WgradDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "yolo_wgrad_slab_floats": [ctypes.POINTER(WgradDesc), ctypes.POINTER(ctypes.c_long)],
    "yolo_conv_stem7_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_float, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_long,
                            c_int, c_int, c_void_p],
    "yolo_conv_stem7_fwd_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_long, c_int, c_int,
                                c_void_p],
    "yolo_wgrad_stem7": [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "yolo_wgrad_stem7_pooled": [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_long, c_int, c_int, c_void_p, c_long, c_int, c_int, c_float, c_void_p,
                                c_void_p, c_void_p, c_long, c_void_p],
    "yolo_wgrad_stem7_codes": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_void_p, c_long, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                               c_long, c_void_p],
    "yolo_maxpool3s2_fwd": [ctypes.POINTER(PoolDesc), c_void_p, c_void_p, c_void_p],
    "yolo_maxpool3s2_bwd": [ctypes.POINTER(PoolDesc), c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "yolo_maxpool2_fwd": [ctypes.POINTER(PoolDesc), c_void_p, c_void_p, c_void_p],
    "yolo_maxpool2_bwd_lrelu": [ctypes.POINTER(PoolDesc), c_void_p, c_void_p, c_float, c_void_p, c_void_p],
    "yolo_maxpool2_bwd_codes": [ctypes.POINTER(PoolDesc), c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p],
    "yolo_nchw_f32_to_nhwc_bf16": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "yolo_nhwc_bf16_to_nchw_f32": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "yolo_nhwc_bf16_to_nchw_bf16": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "yolo_scale_rows_to_bf16": [c_void_p, c_void_p, c_float, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p],
    "yolo_dropout_bf16": [c_void_p, c_void_p, c_float, c_long, c_void_p, c_void_p],
    "yolo_pack_conv_weight": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "yolo_pack_fc_weight": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "yolo_pack_fc_weight_blocked": [c_void_p, c_int, c_long, c_void_p, c_void_p],
    "yolo_pack_fc_weight_blocked_hwc": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "yolo_unpack_conv_wgrad": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p],
    "yolo_im2col_rows": [c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "yolo_transpose_f32_to_bf16": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "yolo_transpose_bf16": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p],
    "yolo_fc_dgrad_to_nhwc": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p],
    "yolo_cast_f32_to_bf16": [c_void_p, c_long, c_void_p, c_void_p],
    "yolo_cast_bf16_to_f32": [c_void_p, c_long, c_void_p, c_void_p],
    "yolo_preprocess_u8": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                           ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p, c_int, c_void_p, c_void_p],
    "yolo_sumsq_f32": [c_void_p, c_long, c_void_p, c_void_p],
    "yolo_adam_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float, c_long, c_void_p, c_float, c_void_p, c_void_p],
    "yolo_clip_scale_f32": [c_void_p, c_long, c_void_p, c_float, c_void_p],
    "yolo_pack_conv_weights_multi": [ctypes.POINTER(ConvPackItem), c_int, c_void_p],
    "yolo_unpack_conv_wgrads_multi": [ctypes.POINTER(ConvUnpackItem), c_int, c_void_p],
    "yolo_sumsq_f32_multi": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "yolo_adam_step_multi": [ctypes.POINTER(AdamTensor), c_int, c_float, c_float, c_float, c_float, c_float, c_long, c_void_p, c_float, c_void_p, c_void_p],
    "yolo_adam_step_multi_bg": [ctypes.POINTER(AdamTensor), c_int, c_float, c_float, c_float, c_float, c_float, c_long, c_void_p, c_float, c_void_p, c_int, c_void_p],
    "yolo_bias_lrelu_rows": [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "yolo_bias_lrelu_rows_slabs": [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
}


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    """Load libyolo_hip.so once.  Raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the gfx950 HIP library is not built (run `make -C yolo-v1_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`).  This package has no CPU/eager fallback for GPU tensors.")
        L = ctypes.CDLL(LIB_PATH)
        L.yolo_hip_abi_version.argtypes, L.yolo_hip_abi_version.restype = [], c_int
        if L.yolo_hip_abi_version() != ABI_VERSION:
            # descriptors travel by pointer: a library built from another header would read (or leave unread) struct tails
            raise RuntimeError(f"{LIB_PATH} has ABI version {L.yolo_hip_abi_version()}, this package binds version {ABI_VERSION}: rebuild it "
                               "(`make -C yolo-v1_amd/csrc`)")
        for name, args in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.argtypes = args
            fn.restype = c_int
        L.yolo_hip_last_error.argtypes = []
        L.yolo_hip_last_error.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


E_ARG, E_UNSUPPORTED = -1, -2      # YOLO_E_ARG, YOLO_E_UNSUPPORTED (include/yolo_hip.h)


class HipUnsupported(RuntimeError):
    """the library refused a shape / configuration (YOLO_E_UNSUPPORTED) -- nothing was launched"""


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().yolo_hip_last_error().decode(errors="replace")
        raise (HipUnsupported if rc == E_UNSUPPORTED else RuntimeError)(f"libyolo_hip {what} failed (code {rc}): {msg}")


def ptr(t) -> c_void_p:
    """Device pointer of a tensor (None -> NULL)."""
    return c_void_p(None) if t is None else c_void_p(t.data_ptr())


def stream() -> c_void_p:
    """The current PyTorch HIP stream as hipStream_t."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def side_stream(device, low: bool = True) -> "torch.cuda.Stream":
    """a second HIP stream on ``device`` (lowest scheduling priority if ``low``), wrapped for torch's stream / event calls"""
    with torch.cuda.device(device):
        h = c_void_p()
        check(lib().yolo_stream_create(1 if low else 0, ctypes.byref(h)), "yolo_stream_create")
        return torch.cuda.ExternalStream(h.value, device=device)


def require_cuda(*tensors) -> None:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libyolo_hip kernels need device tensors (got a CPU tensor)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"libyolo_hip kernels need all tensors on one device (got {dev} and {t.device})")


def device_guard(fn):
    """Run ``fn`` with the CURRENT device set to the device of its tensor arguments.  The kernels are launched on
    ``torch.cuda.current_stream()`` and scratch buffers are allocated on the current device, so a call with tensors on
    ``cuda:1`` while the current device is 0 (``predict.py --device cuda:1``, ``YOLOInference(model, device=...)``) must
    switch first; tensors on different devices are rejected."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        dev = None
        for a in (*args, *kw.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise RuntimeError(f"{fn.__name__}: tensors on different devices ({dev} and {a.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)

    return wrapper
