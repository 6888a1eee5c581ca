"""ctypes binding of ``libdnanet_hip.so`` (C ABI in ``include/dnanet_hip.h``).

The library is built in-tree by ``dnanet_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is
no CPU fallback: if the shared object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_void_p

import torch

F32, BF16 = 0, 1
PACK_FWD, PACK_DGRAD, PACK_CONVT_FWD, PACK_CONVT_DGRAD = 0, 1, 2, 3
PACK_VEC, PACK_UP_BIAS, PACK_FIRST = 4, 5, 6          # fp32 side images, dn_pack_batched only

# DNANET_HIP_LIB: another build of the same library (A/B runs of scratch/ micro-benchmarks); default = the in-tree build
_LIB_PATH = os.environ.get("DNANET_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdnanet_hip.so")


class DnTensor(Structure):
    _fields_ = [("base", c_void_p), ("planes_total", c_int), ("plane0", c_int), ("channels", c_int)]


class DnTensorX3(Structure):
    """``dn_tensor_x3``: a bf16 tensor kept as hi + lo parts ``lo_off`` planes apart (``lo_off`` = 0: plain bf16)."""
    _fields_ = [("base", c_void_p), ("planes_total", c_int), ("plane0", c_int), ("channels", c_int), ("lo_off", c_int)]


class DnPackJob(Structure):
    """``dn_pack_job`` of include/dnanet_hip.h."""
    _fields_ = [("w", c_void_p), ("scale", c_void_p), ("out", c_void_p), ("cout", c_int), ("cin", c_int), ("taps", c_int),
                ("mode", c_int), ("rows_pad", c_int), ("nchunk", c_int), ("gsize", c_int), ("block0", c_int)]


class HipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `make -C dnanet_amd/csrc` (or __graft_entry__.build()). "
            "dnanet_amd has no CPU fallback.")
    return ctypes.CDLL(_LIB_PATH)


lib = _load()

_P = c_void_p
_SIGS = {
    "dn_last_error": (c_char_p, []),
    "dn_abi_version": (c_int, []),
    "dn_packed_weight_bytes": (c_long, [c_int] * 6),
    "dn_pack_weight": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dn_conv_first": (c_int, [_P, _P, _P, DnTensor] + [c_int] * 8 + [_P]),
    "dn_conv2d": (c_int, [DnTensor, _P, _P, DnTensor] + [c_int] * 7 + [_P]),
    "dn_conv_transpose_1x2": (c_int, [DnTensor, _P, _P, DnTensor, c_int, c_int, c_int, c_int, _P]),
    "dn_conv_transpose_1x2_dgrad": (c_int, [DnTensor, _P, DnTensor, c_int, c_int, c_int, c_int, _P]),
    "dn_conv_transpose_1x2_dgrad_bn_sums": (c_int, [DnTensor, _P, DnTensor, DnTensor, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_long,
                                                    POINTER(c_int), _P]),
    "dn_maxpool_1x2": (c_int, [DnTensor, DnTensor, c_int, c_int, c_int, c_int, _P]),
    "dn_head_1x1": (c_int, [DnTensor, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dn_reduce_blocks": (c_int, [c_long]),
    "dn_bn_train_forward": (c_int, [DnTensor, DnTensor, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float,
                                    _P, _P, _P, _P, _P, c_int, _P]),
    "dn_bn_train_forward_noalloc": (c_int, [DnTensor, DnTensor, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float,
                                            _P, _P, _P, _P, c_int, _P]),
    "dn_bn_partial_totals": (c_int, [_P, c_int, c_int, _P, _P]),
    "dn_bn_train_apply_totals": (c_int, [DnTensor, DnTensor, DnTensor, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float,
                                         _P, _P, _P, _P, _P, c_double, c_int, _P]),
    "dn_bn_train_backward_totals": (c_int, [DnTensor, DnTensor, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "dn_bn_train_backward_apply_totals": (c_int, [DnTensor, DnTensor, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_double,
                                                  _P, _P, _P, c_int, _P]),
    "dn_bn_train_backward": (c_int, [DnTensor, DnTensor, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "dn_maxpool_1x2_backward": (c_int, [DnTensor, DnTensor, DnTensor, DnTensor, c_int, c_int, c_int, c_int, _P]),
    "dn_head_1x1_backward": (c_int, [_P, DnTensor, _P, DnTensor, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dn_channel_sum": (c_int, [DnTensor, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dn_conv2d_wgrad": (c_int, [DnTensor, DnTensor, _P, _P, c_long] + [c_int] * 8 + [_P]),
    "dn_conv_transpose_1x2_wgrad": (c_int, [DnTensor, DnTensor, _P, _P, c_long, c_int, c_int, c_int, c_int, _P]),
    "dn_conv_first_wgrad_blocks": (c_int, [c_int, c_int, c_int]),
    "dn_conv_first_wgrad": (c_int, [_P, DnTensor, _P, _P, c_long] + [c_int] * 6 + [_P]),
    "dn_dice_forward": (c_int, [_P, _P, c_long, _P, _P, _P]),
    "dn_dice_backward": (c_int, [_P, _P, c_long, _P, _P, _P]),
    "dn_adam_step": (c_int, [_P, _P, _P, _P, c_long, c_float, c_float, c_float, c_float, c_float, c_long, c_float, c_int, _P]),
    "dn_adam_step_graph": (c_int, [_P, _P, _P, _P, c_long, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "dn_i16_to_f32": (c_int, [_P, _P, c_long, _P]),
    "dn_pack_jobs_prepare": (c_long, [_P, c_int, c_int]),
    "dn_pack_batched": (c_int, [_P, c_int, c_long, c_int, _P]),
    "dn_rescale_profiles": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "dn_pixel_confusion": (c_int, [_P, _P, c_int, c_long, c_float, c_float, _P, _P]),
    "dn_conv2d_variant": (c_int, [c_int] * 8 + [ctypes.c_char_p, c_int]),
    "dn_conv2d_pool": (c_int, [DnTensor, _P, _P, DnTensor, DnTensor, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dn_conv2d_head_supported": (c_int, [c_int] * 8),
    "dn_conv2d_head": (c_int, [DnTensor, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dn_conv_first_stats": (c_int, [_P, _P, _P, DnTensor, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_long, POINTER(c_int), _P]),
    "dn_conv2d_dgrad_bn_sums": (c_int, [DnTensor, _P, DnTensor, DnTensor, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                        c_long, POINTER(c_int), _P]),
    "dn_bn_train_backward_sums": (c_int, [DnTensor, DnTensor, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "dn_bn_backward_sums_totals": (c_int, [_P, c_int, c_int, _P, _P]),
    "dn_maxpool_1x2_backward_bn_sums": (c_int, [DnTensor, DnTensor, DnTensor, DnTensor, DnTensor, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                                _P, c_long, POINTER(c_int), _P]),
    "dn_head_1x1_backward_bn_sums": (c_int, [_P, DnTensor, _P, DnTensor, _P, _P, _P, DnTensor, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                             _P, c_long, POINTER(c_int), _P]),
    "dn_conv2d_stats": (c_int, [DnTensor, _P, _P, DnTensor, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_long, POINTER(c_int), _P]),
    "dn_bn_train_apply": (c_int, [DnTensor, DnTensor, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P,
                                  c_int, c_int, _P]),
    "dn_bn_train_apply_pool": (c_int, [DnTensor, DnTensor, DnTensor, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P,
                                  c_int, c_int, _P]),
    "dn_conv2d_x3": (c_int, [DnTensorX3, _P, _P, DnTensorX3, POINTER(DnTensorX3)] + [c_int] * 6 + [_P]),
    "dn_conv2d_head_x3": (c_int, [DnTensorX3, _P, _P, c_int, _P, _P, _P, _P] + [c_int] * 6 + [_P]),
    "dn_conv_first_x3": (c_int, [_P, _P, _P, DnTensorX3] + [c_int] * 7 + [_P]),
    "dn_head_1x1_x3": (c_int, [DnTensorX3, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dn_conv_transpose_1x2_x3": (c_int, [DnTensorX3, _P, _P, DnTensorX3, c_int, c_int, c_int, _P]),
    "dn_call_alleles": (c_int, [_P, _P, _P, c_long, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
}


def _bind():
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


_bind()


def exported_symbols():
    return sorted(_SIGS)


def check(rc: int):
    if rc != 0:
        raise HipError(f"libdnanet_hip error {rc}: {lib.dn_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensor required"
    return t.data_ptr()


def epr(dtype: int) -> int:
    return 16 if dtype == BF16 else 8


def torch_dtype(dtype: int):
    return torch.bfloat16 if dtype == BF16 else torch.float32


def channel_pad(c: int) -> int:
    return (c + 31) // 32 * 32


def partial_floats(c: int) -> int:
    return 512 * channel_pad(c) * 2 + 4096


def first_wgrad_partial_floats(cout: int, kh: int, kw: int, blocks: int = 512) -> int:
    """Capacity ``dn_conv_first_wgrad`` needs in its ``partial`` workspace (``blocks`` = its largest grid)."""
    return blocks * kh * kw * channel_pad(cout)


def up_bias2(b: torch.Tensor, dtype: int) -> torch.Tensor:
    """ConvTranspose2d bias [cout] as ``dn_conv_transpose_1x2`` takes it: 2 * channel_pad(cout) floats in the row order of the
    DN_PACK_CONVT_FWD image, [output plane][output column k][channels of the plane] (zero past cout)."""
    gs, e = channel_pad(b.numel()), epr(dtype)
    rows = torch.arange(2 * gs, device=b.device)
    co = (rows // (2 * e)) * e + rows % e
    padded = torch.zeros(gs, dtype=torch.float32, device=b.device)
    padded[:b.numel()] = b.reshape(-1)
    return padded[co].contiguous()


def planes(c: int, dtype: int) -> int:
    e = epr(dtype)
    return (c + e - 1) // e


class StepArena:
    """The device memory of one pass (a training step: forward + backward; or an eval forward) at one batch shape.

    A pass asks for its buffers in the same order every time, so slot i of the arena IS the i-th buffer of the pass:
    the first pass at a shape allocates, every later one gets the same tensors back -- no allocator call, no
    ``hipMalloc``, no ``record_stream`` bookkeeping in steady state, and fixed addresses for HIP-graph replay.  By default
    nothing is shared between slots inside a pass; the price is that a pass holds the sum of its buffers (~130 MB per
    bf16 6-dye profile for a training step), which is what 288 GB of HBM is for.  A caller whose pass runs on ONE stream
    may ``release`` a buffer it is done with: a later ``take`` of the same shape inside the pass gets it back (stream
    order makes that safe; with the weight gradients on a second stream nothing is released).  Releases happen at the
    same points of every pass, so the slot sequence stays the same.  Contents do not survive ``begin()``."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.slots = []
        self.cursor = 0
        self.allocations = 0        # slots ever created (steady state: constant)
        self.passes = 0
        self._free = {}

    def begin(self) -> "StepArena":
        self.cursor = 0
        self.passes += 1
        self._free = {}
        return self

    def release(self, t: torch.Tensor):
        """``t`` (handed out by ``take`` in this pass) may be handed out again; only for single-stream passes."""
        self._free.setdefault((tuple(t.shape), t.dtype), []).append(t)

    def take(self, shape, dtype, zero: bool = False) -> torch.Tensor:
        shape = tuple(int(v) for v in shape)
        free = self._free.get((shape, dtype))
        if free:
            t = free.pop()
            if zero:
                t.zero_()
            return t
        i = self.cursor
        self.cursor += 1
        if i < len(self.slots):
            t = self.slots[i]
            if tuple(t.shape) == shape and t.dtype == dtype:
                if zero:
                    t.zero_()
                return t
            del self.slots[i:]       # the pass took another route than last time (a key the owner did not foresee)
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        self.slots.append(t)
        self.allocations += 1
        return t

    @property
    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.slots)


class PlaneBuffer:
    """A plane-record activation buffer [N][planes][H][W][epr]: slot of a ``StepArena`` (the passes of ``HipUNet``) or a
    tensor of torch's allocator (tests, one-off callers)."""

    def __init__(self, n: int, planes_total: int, h: int, w: int, dtype: int, device, zero: bool = False,
                 arena: "StepArena | None" = None):
        self.n, self.planes_total, self.h, self.w, self.dtype = n, planes_total, h, w, dtype
        shape = (n, planes_total, h, w, epr(dtype))
        if arena is not None:
            self.data = arena.take(shape, torch_dtype(dtype), zero)
        else:
            self.data = (torch.zeros if zero else torch.empty)(shape, dtype=torch_dtype(dtype), device=device)

    def ref(self, channels: int, plane0: int = 0) -> DnTensor:
        assert plane0 + planes(channels, self.dtype) <= self.planes_total
        t = DnTensor(self.data.data_ptr(), self.planes_total, plane0, channels)
        t._owner = self      # a ref keeps its buffer alive (saved activations outlive the forward call)
        return t

    def ref3(self, channels: int, plane0: int = 0, lo_off: int = 0) -> DnTensorX3:
        """hi + lo view (bf16 buffers only): hi planes from ``plane0``, lo planes ``lo_off`` further; 0 = plain tensor."""
        assert self.dtype == BF16 and plane0 + lo_off + planes(channels, self.dtype) <= self.planes_total
        t = DnTensorX3(self.data.data_ptr(), self.planes_total, plane0, channels, lo_off)
        t._owner = self
        return t

    # conversions for tests / boundaries (torch ops, not on the hot path)
    @staticmethod
    def from_nchw(x: torch.Tensor, dtype: int) -> "PlaneBuffer":
        n, c, h, w = x.shape
        e = epr(dtype)
        p = planes(c, dtype)
        buf = PlaneBuffer(n, p, h, w, dtype, x.device, zero=True)
        xp = torch.zeros((n, p * e, h, w), dtype=torch.float32, device=x.device)
        xp[:, :c] = x
        buf.data.copy_(xp.view(n, p, e, h, w).permute(0, 1, 3, 4, 2))
        return buf

    def to_nchw(self, channels: int, plane0: int = 0) -> torch.Tensor:
        e = epr(self.dtype)
        p = planes(channels, self.dtype)
        d = self.data[:, plane0:plane0 + p].float().permute(0, 1, 4, 2, 3).reshape(self.n, p * e, self.h, self.w)
        return d[:, :channels].contiguous()
