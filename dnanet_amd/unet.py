"""The U-Net forward/backward scheduled on the HIP kernels of ``libdnanet_hip.so``.

Replaces ``UNet.forward`` (reference ``DNAnet/models/segmentation/unet_architecture.py:99-131``):
4 x [DoubleConv -> MaxPool2d((1,2))] -> DoubleConv -> 4 x [ConvTranspose2d -> cat([up, skip]) ->
DoubleConv] -> Conv2d 1x1.  Activations stay in the library's plane-record layout between
kernels; the encoder's second convolution writes its output straight into the skip half of the
decoder's concat buffer and the up-convolution writes into the other half (zero-copy ``cat``).
Only ``(N,1,H,W)`` fp32 tensors cross this class's interface.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import ctypes
import os

import torch

from . import hip
from .hip import BF16, F32, PlaneBuffer, StepArena
from .params import ConvUnit, ParameterTree

BN_EPS = 1e-5       # nn.BatchNorm2d default used by the reference (unet_architecture.py:17,21)
BN_MOMENTUM = 0.1

_DTYPES = {"fp32": F32, "float32": F32, "f32": F32, "bf16": BF16, "bfloat16": BF16}


def parse_dtype(dtype) -> int:
    if isinstance(dtype, int):
        return dtype
    try:
        return _DTYPES[str(dtype).lower().split("x3")[0]]
    except KeyError:
        raise ValueError(f"dtype must be one of {sorted(_DTYPES)}, 'bf16x3' or 'bf16x3@<levels>', got {dtype!r}")


def parse_split_levels(dtype, depth: int) -> int:
    """``bf16x3``: every U-Net level keeps hi + lo bf16 activations and weights (three MFMAs per product: fp32-grade masks
    from the bf16 matrix cores); ``bf16x3@k``: only the k shallowest levels (0..k-1), the deeper ones plain bf16.  0 for
    every other dtype."""
    name = str(dtype).lower()
    if isinstance(dtype, int) or "x3" not in name:
        return 0
    head, _, tail = name.partition("x3")
    if head not in ("bf16", "bfloat16") or (tail and not (tail[0] == "@" and tail[1:].isdigit())):
        raise ValueError(f"dtype {dtype!r}: expected 'bf16x3' or 'bf16x3@<levels>'")
    k = int(tail[1:]) if tail else depth + 1
    if not 1 <= k <= depth + 1:
        raise ValueError(f"dtype {dtype!r}: levels must be in [1, {depth + 1}]")
    return k


class HipUNet(ParameterTree):
    """Parameter tree + HIP execution.  ``forward`` takes/returns fp32 NCHW tensors on the GPU."""

    MAX_CHUNK = 256     # profiles per inference pass (eval-mode BN is batch independent)

    def __init__(self, depth: int, kernel_size: Sequence[int], num_filters: int, device, dtype="fp32"):
        super().__init__(depth, kernel_size, num_filters, device)
        if num_filters % 8:
            raise ValueError("num_filters must be a multiple of 8 for the MI355X kernels")
        self.dtype = parse_dtype(dtype)
        self.split_levels = parse_split_levels(dtype, depth)       # > 0: the bf16x3 inference mode on that many levels
        if self.split_levels and tuple(kernel_size) not in ((3, 5), (1, 3)):
            raise ValueError("dtype bf16x3 is built for kernel_size (3, 5) and (1, 3)")
        self._eval_cache = None
        self._eval_cache_version = None
        self.profile = None     # set to a list to collect (kernel tag, flops, start event, end event) per conv launch
        self._param_epoch = 0   # bumped whenever a HIP kernel rewrites parameters / running statistics
        self._train_cache = None
        self._train_cache_version = None
        self._saved = None
        self._ws = None
        self.wgrad_workspace_bytes = 256 << 20
        # synchronised BatchNorm (data parallel, optional): a (all_reduce(tensor), world_size) pair set by
        # DataParallelTrainer(sync_bn=True); None = per-replica batch statistics
        self.sync_bn = None
        # Backward runs the weight gradients on a second HIP stream: they are off the critical path (nothing in the
        # backward chain reads dW), MFMA-bound, and overlap the HBM-bound BatchNorm / pooling passes of the following
        # units on the main stream.  DN_TUNE_NO_WGRAD_STREAM=1 keeps everything on one stream (A/B switch).
        self.overlap_wgrad = self.flat_params.is_cuda and not os.environ.get("DN_TUNE_NO_WGRAD_STREAM")
        # BatchNorm backward's sums in the epilogue of the data gradient that produces the unit's activation gradient
        self.fuse_bn_sums = not os.environ.get("DN_TUNE_NO_BN_SUMS")
        self.fuse_bn_sums_convt = not os.environ.get("DN_TUNE_NO_BN_SUMS_CONVT")   # ... of the transposed convolution's data gradient
        self._wgrad_stream = None
        self.two_stream_max_pixels = 20_000_000       # N*H*W above which backward stays on one stream (see _side)
        # Device memory of the passes: one StepArena per (kind of pass, batch shape), allocated by the first pass at that
        # shape and reused by every later one (no allocator calls in steady state).  Least recently used arenas are
        # dropped when the total passes ``arena_budget_bytes``.
        self._arenas: Dict[tuple, StepArena] = {}
        self._arena_override: Optional[StepArena] = None     # GraphedTrainStep: the captured step's own arena
        self.arena_budget_bytes = int(os.environ.get("DN_ARENA_BUDGET_GIB", "160")) << 30

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.flat_params.device

    def _require_gpu(self, x: torch.Tensor):
        if not (self.device.type == "cuda" and x.is_cuda):
            raise RuntimeError("dnanet_amd runs on an MI355X only (no CPU fallback): model and input must be on 'cuda'")

    def _arena(self, kind: str, n: int, h: int, w: int) -> StepArena:
        """The arena of a ``kind`` pass at this batch shape, rewound to its first slot."""
        if self._arena_override is not None:
            return self._arena_override.begin()
        key = (kind, n, h, w, self.dtype, self.split_levels, self.profile is not None, self.sync_bn is not None)
        arena = self._arenas.pop(key, None)
        if arena is None:
            arena = StepArena(self.device)
            total = sum(a.nbytes for a in self._arenas.values())
            while self._arenas and total > self.arena_budget_bytes:
                total -= self._arenas.pop(next(iter(self._arenas))).nbytes       # least recently used first
        self._arenas[key] = arena          # most recently used last
        return arena.begin()

    def arena_stats(self) -> dict:
        """{'arenas', 'bytes', 'allocations'}: ``allocations`` stops growing once every batch shape has been seen."""
        arenas = list(self._arenas.values())
        return {"arenas": len(arenas), "bytes": sum(a.nbytes for a in arenas), "allocations": sum(a.allocations for a in arenas)}

    def release_arenas(self):
        self._arenas.clear()

    def _pp(self, c: int) -> int:
        return hip.planes(c, self.dtype)

    def _cpad_planes(self, c: int) -> int:
        """Channel count as stored (whole planes)."""
        return self._pp(c) * hip.epr(self.dtype)

    def _virtual_weight(self, unit: ConvUnit, w: torch.Tensor) -> torch.Tensor:
        """Conv weight over the input channels as they sit in planes.  For a concat input whose
        halves do not fill whole planes, zero columns are inserted so that channel ``i`` of the
        skip half lines up with plane ``P + i // epr``."""
        if unit.split and unit.split % hip.epr(self.dtype):
            half = self._cpad_planes(unit.split)
            wv = torch.zeros((w.shape[0], 2 * half, w.shape[2], w.shape[3]), dtype=w.dtype, device=w.device)
            wv[:, :unit.split] = w[:, :unit.split]
            wv[:, half:half + unit.split] = w[:, unit.split:]
            return wv
        return w

    def _pack(self, w: torch.Tensor, scale: Optional[torch.Tensor], mode: int) -> torch.Tensor:
        if mode in (hip.PACK_FWD, hip.PACK_DGRAD):
            cout, cin, kh, kw = w.shape
        else:
            cin, cout, kh, kw = w.shape[0], w.shape[1], 1, 2
        nbytes = hip.lib.dn_packed_weight_bytes(cout, cin, kh, kw, mode, self.dtype)
        packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        w = w.contiguous()
        hip.check(hip.lib.dn_pack_weight(hip.ptr(w), hip.ptr(scale), hip.ptr(packed), cout, cin, kh, kw, mode,
                                         self.dtype, hip.stream_ptr()))
        return packed

    def _padvec(self, v: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
        n = n or hip.channel_pad(v.numel())
        out = torch.zeros(n, dtype=torch.float32, device=self.device)
        out[:v.numel()] = v.reshape(-1)
        return out

    def _up_bias2(self, b: torch.Tensor) -> torch.Tensor:
        return hip.up_bias2(b, self.dtype)

    # ------------------------------------------------------------------------------------------
    # eval-mode (BatchNorm folded into the packed weights)
    # ------------------------------------------------------------------------------------------
    def _versions(self):
        return (self.flat_params._version, self.flat_stats._version, self._param_epoch, self.dtype, self.split_levels)

    def _unit_level(self, name: str) -> int:
        """U-Net level (0 = full scan resolution) a convolution / up-convolution writes at."""
        part, idx = name.split(".")[0], name.split(".")[1]
        return self.depth if part == "bottleneck" else int(idx) if part == "encoders" else self.depth - 1 - int(idx)

    def _split_terms(self, w: torch.Tensor, dim: int, split_input: bool) -> torch.Tensor:
        """fp32 weight -> the virtual-input-channel image of the bf16x3 kernels: channels along ``dim`` padded to whole planes,
        then [W_hi | W_hi | W_lo] (against the K walk [x_hi | x_lo | x_hi] of a split input) or [W_hi | W_lo] (plain input)."""
        e = hip.epr(BF16)
        pad = (-w.shape[dim]) % e
        if pad:
            shape = list(w.shape)
            shape[dim] = pad
            w = torch.cat([w, torch.zeros(shape, dtype=w.dtype, device=w.device)], dim)
        hi = w.bfloat16().float()
        lo = w - hi
        return torch.cat([hi, hi, lo] if split_input else [hi, lo], dim).contiguous()

    @torch.no_grad()
    def _build_eval_cache(self):
        cache: Dict[str, dict] = {}
        for u in self.conv_units:
            w = self.tensor(u.conv + ".weight")
            b = self.tensor(u.conv + ".bias")
            gamma, beta = self.tensor(u.bn + ".weight"), self.tensor(u.bn + ".bias")
            rm, rv = self.tensor(u.bn + ".running_mean"), self.tensor(u.bn + ".running_var")
            scale = gamma / torch.sqrt(rv + BN_EPS)
            shift = (b - rm) * scale + beta
            entry = {"bias": self._padvec(shift)}
            if u.cin == 1:
                kh, kw = self.kernel_size
                wf = torch.zeros((kh * kw, hip.channel_pad(u.cout)), dtype=torch.float32, device=self.device)
                wf[:, :u.cout] = (w * scale.view(-1, 1, 1, 1)).reshape(u.cout, kh * kw).t()
                entry["wf"] = wf
            elif self._unit_level(u.conv) < self.split_levels:
                wv = self._virtual_weight(u, w * scale.view(-1, 1, 1, 1))
                entry["w"] = self._pack(self._split_terms(wv, 1, True), None, hip.PACK_FWD)
            else:
                entry["w"] = self._pack(self._virtual_weight(u, w), scale.contiguous(), hip.PACK_FWD)
            cache[u.conv] = entry
        for up in self.up_units:
            wu = self.tensor(up.name + ".weight")
            level = self._unit_level(up.name)
            if level < self.split_levels:
                wu = self._split_terms(wu, 0, level + 1 < self.split_levels)
            cache[up.name] = {"w": self._pack(wu, None, hip.PACK_CONVT_FWD),
                              "bias": self._up_bias2(self.tensor(up.name + ".bias"))}
        cache["head_w"] = self._padvec(self.tensor("outputs.weight"))
        cache["head_b"] = self.tensor("outputs.bias").clone()
        self._eval_cache = cache
        self._eval_cache_version = self._versions()

    def conv_kernel_tag(self, cin: int, cout: int, n: int, h: int, w: int) -> str:
        """Name of the instantiation the library dispatches this convolution to (api.hip: choose_variant)."""
        kh, kw = self.kernel_size
        buf = ctypes.create_string_buffer(96)
        hip.check(hip.lib.dn_conv2d_variant(cin, cout, n, h, w, kh, kw, self.dtype, buf, len(buf)))
        return buf.value.decode()

    def _timed(self, tag, flops: float, launch):
        """``launch()``; with ``self.profile`` a list, bracketed by HIP events on the launch stream and recorded as
        (kernel tag, nominal FLOP, start, end) -- what bench.py's roofline figures are computed from.  ``tag`` may be a
        callable (only evaluated when profiling)."""
        if self.profile is None:
            launch()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self.profile.append((tag() if callable(tag) else tag, flops, e0, e1))

    def _conv(self, u: ConvUnit, entry, src, dst, n, h, w, relu=1, pooled=None):
        """``pooled``: also write MaxPool2d((1,2)) of the output there (fused epilogue, ``dn_conv2d_pool``)."""
        if self.profile is not None and u.cin != 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._conv_launch(u, entry, src, dst, n, h, w, relu, pooled)
            e1.record()
            kh, kw = self.kernel_size
            tag = self.conv_kernel_tag(u.cin, u.cout, n, h, w) + ("+pool" if pooled is not None else "")
            self.profile.append((tag, 2.0 * n * h * w * u.cout * u.cin * kh * kw, e0, e1))
        else:
            self._conv_launch(u, entry, src, dst, n, h, w, relu, pooled)

    def _conv_launch(self, u: ConvUnit, entry, src, dst, n, h, w, relu=1, pooled=None):
        kh, kw = self.kernel_size
        if pooled is not None and (u.cin == 1 or (kh, kw) not in ((3, 5), (1, 3))):
            self._conv_launch(u, entry, src, dst, n, h, w, relu)
            hip.check(hip.lib.dn_maxpool_1x2(dst, pooled, n, h, w // 2, self.dtype, hip.stream_ptr()))
        elif pooled is not None:
            hip.check(hip.lib.dn_conv2d_pool(src, hip.ptr(entry["w"]), hip.ptr(entry["bias"]), dst, pooled, n, h, w, kh, kw,
                                             relu, self.dtype, hip.stream_ptr()))
        elif u.cin == 1:
            hip.check(hip.lib.dn_conv_first(hip.ptr(src), hip.ptr(entry["wf"]), hip.ptr(entry["bias"]), dst, n, h, w,
                                            u.cout, kh, kw, relu, self.dtype, hip.stream_ptr()))
        else:
            hip.check(hip.lib.dn_conv2d(src, hip.ptr(entry["w"]), hip.ptr(entry["bias"]), dst, n, h, w, kh, kw, relu,
                                        self.dtype, hip.stream_ptr()))

    @torch.no_grad()
    def _forward_eval_chunk(self, x: torch.Tensor, logits: Optional[torch.Tensor], probs: Optional[torch.Tensor]):
        n, _, h, w = x.shape
        depth, f, dt, dev = self.depth, self.num_filters, self.dtype, self.device
        c = [f * 2 ** l for l in range(depth + 1)]
        ws = [w >> l for l in range(depth + 1)]
        cache = self._eval_cache
        units = {u.conv: u for u in self.conv_units}
        A = self._arena("eval", n, h, w)
        cat = [PlaneBuffer(n, 2 * self._pp(c[l]), h, ws[l], dt, dev, arena=A) for l in range(depth)]
        cur = x
        for l in range(depth):
            pre = f"encoders.{l}.conv.double_conv"
            tmp = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev, arena=A)
            self._conv(units[pre + ".0"], cache[pre + ".0"], cur if l == 0 else cur.ref(c[l - 1]), tmp.ref(c[l]), n, h, ws[l])
            skip = cat[l].ref(c[l], self._pp(c[l]))
            pooled = PlaneBuffer(n, self._pp(c[l]), h, ws[l + 1], dt, dev, arena=A)
            self._conv(units[pre + ".3"], cache[pre + ".3"], tmp.ref(c[l]), skip, n, h, ws[l], pooled=pooled.ref(c[l]))
            # an eval pass runs on one stream: a buffer whose reader has been queued can serve a later layer of the pass
            A.release(tmp.data)
            if l > 0:
                A.release(cur.data)
            cur = pooled
        pre = "bottleneck.double_conv"
        tmp = PlaneBuffer(n, self._pp(c[depth]), h, ws[depth], dt, dev, arena=A)
        self._conv(units[pre + ".0"], cache[pre + ".0"], cur.ref(c[depth - 1]), tmp.ref(c[depth]), n, h, ws[depth])
        A.release(cur.data)
        out = PlaneBuffer(n, self._pp(c[depth]), h, ws[depth], dt, dev, arena=A)
        self._conv(units[pre + ".3"], cache[pre + ".3"], tmp.ref(c[depth]), out.ref(c[depth]), n, h, ws[depth])
        A.release(tmp.data)
        cur = out
        for i in range(depth):
            l = depth - 1 - i
            up = cache[f"decoders.{i}.up"]
            hip.check(hip.lib.dn_conv_transpose_1x2(cur.ref(c[l + 1]), hip.ptr(up["w"]), hip.ptr(up["bias"]),
                                                    cat[l].ref(c[l], 0), n, h, ws[l + 1], dt, hip.stream_ptr()))
            A.release(cur.data)
            pre = f"decoders.{i}.conv.double_conv"
            tmp = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev, arena=A)
            both = cat[l].ref(2 * self._cpad_planes(c[l]), 0)
            self._conv(units[pre + ".0"], cache[pre + ".0"], both, tmp.ref(c[l]), n, h, ws[l])
            A.release(cat[l].data)
            last = units[pre + ".3"]
            kh, kw = self.kernel_size
            if l == 0 and hip.lib.dn_conv2d_head_supported(last.cin, last.cout, n, h, w, kh, kw, dt):
                # the 1x1 output convolution + sigmoid ride in the last convolution's epilogue: its activation is never stored
                entry = cache[pre + ".3"]
                if self.profile is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                hip.check(hip.lib.dn_conv2d_head(tmp.ref(c[0]), hip.ptr(entry["w"]), hip.ptr(entry["bias"]), last.cout,
                                                 hip.ptr(cache["head_w"]), hip.ptr(cache["head_b"]), hip.ptr(logits),
                                                 hip.ptr(probs), n, h, w, kh, kw, 1, dt, hip.stream_ptr()))
                if self.profile is not None:
                    e1.record()
                    self.profile.append((self.conv_kernel_tag(last.cin, last.cout, n, h, w) + "+head",
                                         2.0 * n * h * w * last.cout * (last.cin * kh * kw + 1), e0, e1))
                return
            out = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev, arena=A)
            self._conv(last, cache[pre + ".3"], tmp.ref(c[l]), out.ref(c[l]), n, h, ws[l])
            A.release(tmp.data)
            cur = out
        hip.check(hip.lib.dn_head_1x1(cur.ref(c[0]), hip.ptr(cache["head_w"]), hip.ptr(cache["head_b"]),
                                      hip.ptr(logits), hip.ptr(probs), n, h, w, dt, hip.stream_ptr()))

    @torch.no_grad()
    def _forward_eval_chunk_x3(self, x: torch.Tensor, logits: Optional[torch.Tensor], probs: Optional[torch.Tensor]):
        """The eval forward with hi + lo split tensors on the ``split_levels`` shallowest levels (``dn_*_x3``), plain bf16
        below.  A split tensor of P planes occupies 2P planes of its buffer ([hi | lo]); a split concat buffer is
        [up_hi | skip_hi | up_lo | skip_lo]."""
        n, _, h, w = x.shape
        depth, f, dt, dev = self.depth, self.num_filters, BF16, self.device
        kh, kw = self.kernel_size
        c = [f * 2 ** l for l in range(depth + 1)]
        ws = [w >> l for l in range(depth + 1)]
        P = [self._pp(ch) for ch in c]
        S = lambda l: l < self.split_levels
        cache = self._eval_cache
        units = {u.conv: u for u in self.conv_units}
        stream = hip.stream_ptr()
        A = self._arena("eval", n, h, w)

        def act(l, chan_level, units_wide=1):
            """buffer for `units_wide` tensors of c[chan_level] channels at level l's width and precision"""
            return PlaneBuffer(n, units_wide * P[chan_level] * (2 if S(l) else 1), h, ws[l], dt, dev, arena=A)

        def r3(buf, l, channels, plane0, hi_planes):
            return buf.ref3(channels, plane0, hi_planes if S(l) else 0)

        def conv(name, l, src3, src, dst3, dst, pooled3=None, pooled=None):
            u, entry = units[name], cache[name]
            if self.profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if not S(l):
                self._conv_launch(u, entry, src, dst, n, h, ws[l], 1, pooled)
            elif u.cin == 1:
                hip.check(hip.lib.dn_conv_first_x3(hip.ptr(src), hip.ptr(entry["wf"]), hip.ptr(entry["bias"]), dst3, n, h, ws[l],
                                                   u.cout, kh, kw, 1, stream))
            else:
                hip.check(hip.lib.dn_conv2d_x3(src3, hip.ptr(entry["w"]), hip.ptr(entry["bias"]), dst3,
                                               ctypes.byref(pooled3) if pooled3 is not None else None, n, h, ws[l], kh, kw, 1, stream))
            if self.profile is not None and u.cin != 1:
                e1.record()
                tag = ("x3:" if S(l) else "") + self.conv_kernel_tag(u.cin, u.cout, n, h, ws[l]) + ("+pool" if pooled3 is not None else "")
                self.profile.append((tag, 2.0 * n * h * ws[l] * u.cout * u.cin * kh * kw, e0, e1))

        cat = [act(l, l, 2) for l in range(depth)]
        cur = x
        for l in range(depth):
            pre = f"encoders.{l}.conv.double_conv"
            tmp = act(l, l)
            conv(pre + ".0", l, None if l == 0 else r3(cur, l, c[l - 1], 0, P[l - 1]), cur if l == 0 else cur.ref(c[l - 1]),
                 r3(tmp, l, c[l], 0, P[l]), tmp.ref(c[l]))
            pooled = act(l + 1, l)
            conv(pre + ".3", l, r3(tmp, l, c[l], 0, P[l]), tmp.ref(c[l]), r3(cat[l], l, c[l], P[l], 2 * P[l]), cat[l].ref(c[l], P[l]),
                 r3(pooled, l + 1, c[l], 0, P[l]), pooled.ref(c[l]))
            cur = pooled
        pre = "bottleneck.double_conv"
        tmp, out = act(depth, depth), act(depth, depth)
        conv(pre + ".0", depth, r3(cur, depth, c[depth - 1], 0, P[depth - 1]), cur.ref(c[depth - 1]),
             r3(tmp, depth, c[depth], 0, P[depth]), tmp.ref(c[depth]))
        conv(pre + ".3", depth, r3(tmp, depth, c[depth], 0, P[depth]), tmp.ref(c[depth]),
             r3(out, depth, c[depth], 0, P[depth]), out.ref(c[depth]))
        cur = out
        for i in range(depth):
            l = depth - 1 - i
            up = cache[f"decoders.{i}.up"]
            if S(l):
                hip.check(hip.lib.dn_conv_transpose_1x2_x3(r3(cur, l + 1, c[l + 1], 0, P[l + 1]), hip.ptr(up["w"]), hip.ptr(up["bias"]),
                                                           r3(cat[l], l, c[l], 0, 2 * P[l]), n, h, ws[l + 1], stream))
            else:
                hip.check(hip.lib.dn_conv_transpose_1x2(cur.ref(c[l + 1]), hip.ptr(up["w"]), hip.ptr(up["bias"]),
                                                        cat[l].ref(c[l], 0), n, h, ws[l + 1], dt, stream))
            pre = f"decoders.{i}.conv.double_conv"
            tmp = act(l, l)
            cpad2 = 2 * self._cpad_planes(c[l])
            conv(pre + ".0", l, r3(cat[l], l, cpad2, 0, 2 * P[l]), cat[l].ref(cpad2, 0), r3(tmp, l, c[l], 0, P[l]), tmp.ref(c[l]))
            last = units[pre + ".3"]
            if l == 0 and hip.lib.dn_conv2d_head_supported(last.cin, last.cout, n, h, w, kh, kw, dt):
                entry = cache[pre + ".3"]
                if self.profile is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                hip.check(hip.lib.dn_conv2d_head_x3(r3(tmp, 0, c[0], 0, P[0]), hip.ptr(entry["w"]), hip.ptr(entry["bias"]), last.cout,
                                                    hip.ptr(cache["head_w"]), hip.ptr(cache["head_b"]), hip.ptr(logits),
                                                    hip.ptr(probs), n, h, w, kh, kw, 1, stream))
                if self.profile is not None:
                    e1.record()
                    self.profile.append(("x3:" + self.conv_kernel_tag(last.cin, last.cout, n, h, w) + "+head",
                                         2.0 * n * h * w * last.cout * (last.cin * kh * kw + 1), e0, e1))
                return
            out = act(l, l)
            conv(pre + ".3", l, r3(tmp, l, c[l], 0, P[l]), tmp.ref(c[l]), r3(out, l, c[l], 0, P[l]), out.ref(c[l]))
            cur = out
        # shapes whose last convolution has no single-tile (fused-head) kernel: the head reads the split tensor
        hip.check(hip.lib.dn_head_1x1_x3(r3(cur, 0, c[0], 0, P[0]), hip.ptr(cache["head_w"]), hip.ptr(cache["head_b"]),
                                         hip.ptr(logits), hip.ptr(probs), n, h, w, stream))

    def _check_input(self, x: torch.Tensor):
        self._require_gpu(x)
        if x.dim() != 4 or x.shape[1] != 1 or x.dtype != torch.float32:
            raise ValueError(f"expected an (N,1,H,W) float32 tensor, got {tuple(x.shape)} {x.dtype}")
        if x.shape[3] % (1 << self.depth):
            raise ValueError(f"scan length {x.shape[3]} is not divisible by 2**depth = {1 << self.depth}")

    @torch.no_grad()
    def infer(self, x: torch.Tensor, want_logits: bool = True, want_probs: bool = True):
        """Eval-mode forward: returns (logits, probabilities), each (N,1,H,W) fp32 or None."""
        self._check_input(x)
        x = x.contiguous()
        if self._eval_cache is None or self._eval_cache_version != self._versions():
            self._build_eval_cache()
        n = x.shape[0]
        logits = torch.empty_like(x) if want_logits else None
        probs = torch.empty_like(x) if want_probs else None
        for s in range(0, n, self.MAX_CHUNK):
            e = min(n, s + self.MAX_CHUNK)
            chunk = self._forward_eval_chunk_x3 if self.split_levels else self._forward_eval_chunk
            chunk(x[s:e], logits[s:e] if want_logits else None, probs[s:e] if want_probs else None)
        return logits, probs

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            return self.forward_train(x)
        return self.infer(x, want_logits=True, want_probs=False)[0]

    # ------------------------------------------------------------------------------------------
    # training: forward with batch statistics, backward, both on the HIP kernels
    # ------------------------------------------------------------------------------------------
    def _workspace(self):
        if self._ws is None:
            cmax = self.num_filters * 2 ** self.depth
            kh, kw = self.kernel_size
            # reduction partials: the per-channel reductions need dn_partial_floats(C); the first convolution's weight
            # gradient needs (up to 512 blocks) x taps x channel_pad(filters), which is the larger one for small nets
            partial = max(hip.partial_floats(cmax), hip.first_wgrad_partial_floats(self.num_filters, kh, kw)) + 65536
            self._ws = {
                "partial": torch.empty(partial, dtype=torch.float32, device=self.device),
                "coef": torch.empty(3 * hip.channel_pad(cmax), dtype=torch.float32, device=self.device),
                "wgrad": torch.empty(self.wgrad_workspace_bytes, dtype=torch.uint8, device=self.device),
                "dice_sums": torch.zeros(8, dtype=torch.float32, device=self.device),
            }
        return self._ws

    @torch.no_grad()
    def _build_train_cache(self):
        """Packed GEMM images of the current weights (forward and data-gradient forms), padded biases, the first
        convolution's tap-major weights.  The images live in buffers allocated once; a job table on the device
        (``dn_pack_job``) lets ONE launch refresh all of them -- the weights change every training step."""
        key = (self.dtype, self.flat_params.data_ptr())
        if self._train_cache is None or self._train_cache.get("_key") != key:
            cache: Dict[str, dict] = {"_key": key}
            jobs: List[hip.DnPackJob] = []
            kh, kw = self.kernel_size

            def job(w: torch.Tensor, out: torch.Tensor, cout: int, cin: int, taps: int, mode: int) -> torch.Tensor:
                jobs.append(hip.DnPackJob(hip.ptr(w), None, hip.ptr(out), cout, cin, taps, mode, 0, 0, 0, 0))
                return out

            def image(w: torch.Tensor, mode: int) -> torch.Tensor:
                if mode in (hip.PACK_FWD, hip.PACK_DGRAD):
                    cout, cin, th, tw = w.shape
                else:
                    cin, cout, th, tw = w.shape[0], w.shape[1], 1, 2
                nbytes = hip.lib.dn_packed_weight_bytes(cout, cin, th, tw, mode, self.dtype)
                return job(w, torch.empty(nbytes, dtype=torch.uint8, device=self.device), cout, cin, th * tw, mode)

            def floats(n: int) -> torch.Tensor:
                return torch.empty(n, dtype=torch.float32, device=self.device)

            for u in self.conv_units:
                w, b = self.tensor(u.conv + ".weight"), self.tensor(u.conv + ".bias")
                entry = {"bias": job(b, floats(hip.channel_pad(u.cout)), u.cout, 1, 1, hip.PACK_VEC)}
                if u.cin == 1:
                    entry["wf"] = job(w, floats(kh * kw * hip.channel_pad(u.cout)), u.cout, 1, kh * kw,
                                      hip.PACK_FIRST).view(kh * kw, hip.channel_pad(u.cout))
                elif u.split and u.split % hip.epr(self.dtype):
                    entry["_virtual"] = u           # concat halves that do not fill whole planes: widened copy per build
                else:
                    entry["w"], entry["wd"] = image(w, hip.PACK_FWD), image(w, hip.PACK_DGRAD)
                cache[u.conv] = entry
            for up in self.up_units:
                w, b = self.tensor(up.name + ".weight"), self.tensor(up.name + ".bias")
                cache[up.name] = {"w": image(w, hip.PACK_CONVT_FWD), "wd": image(w, hip.PACK_CONVT_DGRAD),
                                  "bias": job(b, floats(2 * hip.channel_pad(b.numel())), b.numel(), 1, 1, hip.PACK_UP_BIAS)}
            hw = self.tensor("outputs.weight")
            cache["head_w"] = job(hw, floats(hip.channel_pad(hw.numel())), hw.numel(), 1, 1, hip.PACK_VEC)
            table = (hip.DnPackJob * len(jobs))(*jobs)
            blocks = hip.lib.dn_pack_jobs_prepare(table, len(jobs), self.dtype)
            if blocks < 0:
                hip.check(1)
            raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).clone()
            cache["_jobs"] = (raw.to(self.device), len(jobs), blocks)
            self._train_cache = cache
        cache = self._train_cache
        jobs_dev, njobs, blocks = cache["_jobs"]
        hip.check(hip.lib.dn_pack_batched(hip.ptr(jobs_dev), njobs, blocks, self.dtype, hip.stream_ptr()))
        for entry in cache.values():
            if isinstance(entry, dict) and "_virtual" in entry:
                wv = self._virtual_weight(entry["_virtual"], self.tensor(entry["_virtual"].conv + ".weight"))
                entry["w"], entry["wd"] = self._pack(wv, None, hip.PACK_FWD), self._pack(wv, None, hip.PACK_DGRAD)
        self._train_cache_version = self._versions()

    def _unit_forward(self, u: ConvUnit, src, dst, n, h, w, raw_x=None, pooled=None):
        """conv + bias -> y ; batch-stat BatchNorm + ReLU -> dst (and MaxPool2d((1,2)) of it -> pooled).  Saves what
        backward needs."""
        ws, cache, A = self._workspace(), self._train_cache, self._saved["arena"]
        y = PlaneBuffer(n, self._pp(u.cout), h, w, self.dtype, self.device, arena=A)
        cp = hip.channel_pad(u.cout)
        vec = A.take((4 * cp,), torch.float32)
        mean, invstd, scale, shift = (vec[i * cp:(i + 1) * cp] for i in range(4))
        bn_args = (hip.ptr(self.tensor(u.bn + ".weight")), hip.ptr(self.tensor(u.bn + ".bias")),
                   hip.ptr(self.tensor(u.bn + ".running_mean")), hip.ptr(self.tensor(u.bn + ".running_var")), BN_MOMENTUM, BN_EPS,
                   hip.ptr(mean), hip.ptr(invstd), hip.ptr(scale), hip.ptr(shift))
        kh, kw = self.kernel_size
        if (kh, kw) in ((3, 5), (1, 3)):
            # the batch statistics come out of the convolution's epilogue: one pass over y saved
            # tiles are at least 3 rows x 128 columns (the first convolution may tile row by row)
            need = n * (h if u.cin == 1 else (h + 2) // 3) * ((w + 127) // 128) * cp * 2
            if ws.get("bnstat") is None or ws["bnstat"].numel() < need:
                ws["bnstat"] = torch.empty(need, dtype=torch.float32, device=self.device)
            ntiles = ctypes.c_int(0)
            entry = cache[u.conv]
            flops = 2.0 * n * h * w * u.cout * u.cin * kh * kw
            if u.cin == 1:
                self._timed("conv_first+stats", flops, lambda: hip.check(hip.lib.dn_conv_first_stats(
                    hip.ptr(raw_x), hip.ptr(entry["wf"]), hip.ptr(entry["bias"]), y.ref(u.cout), n, h, w, u.cout, kh, kw, self.dtype,
                    hip.ptr(ws["bnstat"]), ws["bnstat"].numel(), ctypes.byref(ntiles), hip.stream_ptr())))
            else:
                self._timed(lambda: self.conv_kernel_tag(u.cin, u.cout, n, h, w) + "+stats", flops,
                            lambda: hip.check(hip.lib.dn_conv2d_stats(
                                src, hip.ptr(entry["w"]), hip.ptr(entry["bias"]), y.ref(u.cout), n, h, w, kh, kw, self.dtype,
                                hip.ptr(ws["bnstat"]), ws["bnstat"].numel(), ctypes.byref(ntiles), hip.stream_ptr())))
            if self.sync_bn is not None:
                # batch statistics of the GLOBAL batch: this rank's per-channel totals (fp64) are summed over the group
                all_reduce, world = self.sync_bn
                totals = A.take((2 * cp,), torch.float64)
                hip.check(hip.lib.dn_bn_partial_totals(hip.ptr(ws["bnstat"]), ntiles.value, u.cout, hip.ptr(totals), hip.stream_ptr()))
                all_reduce(totals)
                hip.check(hip.lib.dn_bn_train_apply_totals(y.ref(u.cout), dst, pooled if pooled is not None else hip.DnTensor(None, 0, 0, 0),
                                                           n, h, w, *bn_args, hip.ptr(totals), float(n * h * w * world), self.dtype,
                                                           hip.stream_ptr()))
            elif pooled is not None:
                hip.check(hip.lib.dn_bn_train_apply_pool(y.ref(u.cout), dst, pooled, n, h, w, *bn_args, hip.ptr(ws["bnstat"]),
                                                         ntiles.value, self.dtype, hip.stream_ptr()))
            else:
                hip.check(hip.lib.dn_bn_train_apply(y.ref(u.cout), dst, n, h, w, *bn_args, hip.ptr(ws["bnstat"]), ntiles.value,
                                                    self.dtype, hip.stream_ptr()))
        else:
            if self.sync_bn is not None:
                raise NotImplementedError("synchronised BatchNorm needs the fused-statistics convolutions: kernel sizes (3,5) / (1,3)")
            self._conv(u, cache[u.conv], raw_x if u.cin == 1 else src, y.ref(u.cout), n, h, w, relu=0)
            hip.check(hip.lib.dn_bn_train_forward(y.ref(u.cout), dst, n, h, w, *bn_args, hip.ptr(ws["partial"]), self.dtype,
                                                  hip.stream_ptr()))
            if pooled is not None:
                hip.check(hip.lib.dn_maxpool_1x2(dst, pooled, n, h, w // 2, self.dtype, hip.stream_ptr()))
        self._saved[u.conv] = {"y": y, "a": dst, "src": src, "mean": mean, "invstd": invstd, "scale": scale, "shift": shift,
                               "vec": vec, "n": n, "h": h, "w": w}

    def forward_train(self, x: torch.Tensor) -> torch.Tensor:
        """Train-mode forward (batch statistics, running-stat update); keeps activations for backward()."""
        if self.split_levels:
            raise RuntimeError("dtype bf16x3 is an inference mode: train with dtype fp32 or bf16 and load the weights")
        self._check_input(x)
        x = x.contiguous()
        if self._train_cache is None or self._train_cache_version != self._versions():
            self._build_train_cache()
        n, _, h, w = x.shape
        depth, f, dt, dev = self.depth, self.num_filters, self.dtype, self.device
        c = [f * 2 ** l for l in range(depth + 1)]
        ws = [w >> l for l in range(depth + 1)]
        units = {u.conv: u for u in self.conv_units}
        A = self._arena("train", n, h, w)
        S = self._saved = {"x": x, "n": n, "h": h, "w": w, "arena": A}
        cat = [PlaneBuffer(n, 2 * self._pp(c[l]), h, ws[l], dt, dev, arena=A) for l in range(depth)]
        S["cat"] = cat
        cur = None
        for l in range(depth):
            pre = f"encoders.{l}.conv.double_conv"
            a1 = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev, arena=A)
            self._unit_forward(units[pre + ".0"], None if l == 0 else cur, a1.ref(c[l]), n, h, ws[l], raw_x=x)
            skip = cat[l].ref(c[l], self._pp(c[l]))
            pooled = PlaneBuffer(n, self._pp(c[l]), h, ws[l + 1], dt, dev, arena=A)
            self._unit_forward(units[pre + ".3"], a1.ref(c[l]), skip, n, h, ws[l], pooled=pooled.ref(c[l]))
            S[f"pooled{l}"] = pooled
            cur = pooled.ref(c[l])
        pre = "bottleneck.double_conv"
        a1 = PlaneBuffer(n, self._pp(c[depth]), h, ws[depth], dt, dev, arena=A)
        self._unit_forward(units[pre + ".0"], cur, a1.ref(c[depth]), n, h, ws[depth])
        a2 = PlaneBuffer(n, self._pp(c[depth]), h, ws[depth], dt, dev, arena=A)
        self._unit_forward(units[pre + ".3"], a1.ref(c[depth]), a2.ref(c[depth]), n, h, ws[depth])
        cur, cur_buf = a2.ref(c[depth]), a2
        for i in range(depth):
            l = depth - 1 - i
            up = self._train_cache[f"decoders.{i}.up"]
            S[f"up{i}_in"] = cur
            hip.check(hip.lib.dn_conv_transpose_1x2(cur, hip.ptr(up["w"]), hip.ptr(up["bias"]), cat[l].ref(c[l], 0), n, h,
                                                    ws[l + 1], dt, hip.stream_ptr()))
            pre = f"decoders.{i}.conv.double_conv"
            a1 = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev, arena=A)
            both = cat[l].ref(2 * self._cpad_planes(c[l]), 0)
            self._unit_forward(units[pre + ".0"], both, a1.ref(c[l]), n, h, ws[l])
            a2 = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev, arena=A)
            self._unit_forward(units[pre + ".3"], a1.ref(c[l]), a2.ref(c[l]), n, h, ws[l])
            cur, cur_buf = a2.ref(c[l]), a2
        S["head_in"], S["keep"] = cur, cur_buf
        logits = torch.empty_like(x)
        hip.check(hip.lib.dn_head_1x1(cur, hip.ptr(self._train_cache["head_w"]), hip.ptr(self.tensor("outputs.bias")),
                                      hip.ptr(logits), None, n, h, w, dt, hip.stream_ptr()))
        self.flat_counts += 1          # num_batches_tracked of every BatchNorm
        self._param_epoch += 1         # running statistics changed
        self._train_cache_version = self._versions()
        S["logits"] = logits
        return logits

    def dice_forward(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """Fused sigmoid + Dice sums + accuracy counts; returns the device vector
        [sum p*t, sum p, sum t, #correct, loss] (loss.py:9-20, trainable_unet.py:411-415)."""
        ws = self._workspace()
        if targets.dtype != torch.int8:
            targets = targets.to(torch.int8)
        targets = targets.contiguous()
        sums = torch.zeros(8, dtype=torch.float32, device=self.device)
        hip.check(hip.lib.dn_dice_forward(hip.ptr(logits), hip.ptr(targets), logits.numel(), hip.ptr(sums),
                                          hip.ptr(ws["partial"]), hip.stream_ptr()))
        if self._saved is not None:
            self._saved["targets"], self._saved["sums"] = targets, sums
        return sums

    def _side(self):
        """The weight-gradient stream of this backward pass, or None (single stream: profiling, switch).  Under HIP-graph
        capture the side stream forks from / joins the capturing stream, so the graph gets the same parallel branch."""
        if not self.overlap_wgrad or self.profile is not None or (torch.cuda.is_current_stream_capturing() and
                                                                   os.environ.get("DN_TUNE_NO_GRAPH_FORK")):
            return None
        # measured (bf16, 6 x 4096): two streams win 3 % at 256 profiles, 5 % at 512, lose nothing at 768 -- and lose 40 % at
        # 1024, where the concurrent kernels' working sets (84 GiB live) evict each other
        S = self._saved
        if S is not None and S["n"] * S["h"] * S["w"] > self.two_stream_max_pixels:
            return None
        # fp32 (round 3, N = 256): the fp32 matrix pipe is issue bound, not power bound, and a second stream's kernels only take
        # issue slots from it -- 363.6 ms on one stream against 368.1 on two (5 dyes: 294.0 / 302.7); at 16 profiles equal
        if S is not None and self.dtype == F32 and S["n"] * S["h"] * S["w"] > 2_000_000:
            return None
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device)
        return self._wgrad_stream

    def _launch_wgrad(self, launch):
        """``launch()`` on the weight-gradient stream, ordered after everything queued on the current stream so far.  What
        it reads are slots of the step's arena: nothing else is handed that memory before the pass ends, and backward()
        joins the side stream before it returns (so the next pass, which reuses the slots, is ordered after it)."""
        side = self._side()
        if side is None:
            launch()
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch()

    def _segment_done(self, segment_hook, name):
        """Tell a data-parallel reducer that the gradients of ``name`` are final: on a stream that is ordered after both the
        main stream's and the weight-gradient stream's work so far."""
        side = self._side()
        if side is None:
            segment_hook(name)
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            segment_hook(name)

    def _bn_sums_buffer(self, channels: int, blocks: int = 512) -> torch.Tensor:
        """Workspace for [blocks][channel_pad][2] BatchNorm backward sums (one producer -> one consumer at a time, in stream
        order); the elementwise producers use at most 512 blocks."""
        ws = self._workspace()
        need = max(blocks, 512) * hip.channel_pad(channels) * 2
        if ws.get("bnsums") is None or ws["bnsums"].numel() < need:
            ws["bnsums"] = torch.empty(need, dtype=torch.float32, device=self.device)
        return ws["bnsums"]

    def _unit_backward(self, u: ConvUnit, g, want_dgrad: bool, feeds: Optional[ConvUnit] = None, sums=None):
        """g: dL/d(output activation) in a buffer (overwritten with dL/dy). Returns dL/d(input) ref.
        ``feeds``: the unit whose activation is this unit's input -- the data gradient then forms that unit's BatchNorm
        backward sums in its epilogue and the buffer returned carries them as ``.bn_sums`` = (sums, ntiles), which the
        caller hands to that unit's call as ``sums`` (its reduction pass over g and y is skipped)."""
        ws, cache, sv, A = self._workspace(), self._train_cache, self._saved[u.conv], self._saved["arena"]
        n, h, w = sv["n"], sv["h"], sv["w"]
        kh, kw = self.kernel_size
        if self.sync_bn is not None:
            all_reduce, world = self.sync_bn
            local = A.take((2 * hip.channel_pad(u.cout),), torch.float64)
            if sums is not None:
                hip.check(hip.lib.dn_bn_backward_sums_totals(hip.ptr(sums[0]), sums[1], u.cout, hip.ptr(local), hip.stream_ptr()))
            else:
                hip.check(hip.lib.dn_bn_train_backward_totals(g, sv["y"].ref(u.cout), hip.ptr(sv["scale"]), hip.ptr(sv["shift"]), n, h, w,
                                                              hip.ptr(sv["mean"]), hip.ptr(sv["invstd"]), hip.ptr(ws["partial"]),
                                                              hip.ptr(local), self.dtype, hip.stream_ptr()))
            total = A.take(tuple(local.shape), torch.float64)
            total.copy_(local)
            all_reduce(total)
            hip.check(hip.lib.dn_bn_train_backward_apply_totals(
                g, sv["y"].ref(u.cout), hip.ptr(sv["scale"]), hip.ptr(sv["shift"]), n, h, w, hip.ptr(self.tensor(u.bn + ".weight")),
                hip.ptr(sv["mean"]), hip.ptr(sv["invstd"]), hip.ptr(local), hip.ptr(total), float(n * h * w * world),
                hip.ptr(self.grad(u.bn + ".weight")), hip.ptr(self.grad(u.bn + ".bias")), hip.ptr(ws["coef"]), self.dtype,
                hip.stream_ptr()))
        elif sums is not None:
            hip.check(hip.lib.dn_bn_train_backward_sums(
                g, sv["y"].ref(u.cout), hip.ptr(sv["scale"]), hip.ptr(sv["shift"]), n, h, w,
                hip.ptr(self.tensor(u.bn + ".weight")), hip.ptr(sv["mean"]),
                hip.ptr(sv["invstd"]), hip.ptr(self.grad(u.bn + ".weight")), hip.ptr(self.grad(u.bn + ".bias")),
                hip.ptr(ws["coef"]), hip.ptr(sums[0]), sums[1], self.dtype, hip.stream_ptr()))
        else:
            hip.check(hip.lib.dn_bn_train_backward(
                g, sv["y"].ref(u.cout), hip.ptr(sv["scale"]), hip.ptr(sv["shift"]), n, h, w,
                hip.ptr(self.tensor(u.bn + ".weight")), hip.ptr(sv["mean"]),
                hip.ptr(sv["invstd"]), hip.ptr(self.grad(u.bn + ".weight")), hip.ptr(self.grad(u.bn + ".bias")),
                hip.ptr(ws["coef"]), hip.ptr(ws["partial"]), self.dtype, hip.stream_ptr()))
        # the conv bias feeds a BatchNorm: its gradient is identically zero, the arena entry stays 0
        if u.cin == 1:
            hip.check(hip.lib.dn_conv_first_wgrad(hip.ptr(self._saved["x"]), g, hip.ptr(self.grad(u.conv + ".weight")),
                                                  hip.ptr(ws["partial"]), ws["partial"].numel(), n, h, w, kh, kw, self.dtype,
                                                  hip.stream_ptr()))
            return None
        flops = 2.0 * n * h * w * u.cout * u.cin * kh * kw
        dt_name = "bf16" if self.dtype == BF16 else "f32"
        # (the event pair of a profiled run brackets the split-K GEMM and its fixed-order reduction: one C entry point)
        self._launch_wgrad(lambda: self._timed(f"conv_wgrad<{dt_name},{kh}x{kw}>", flops, lambda: hip.check(hip.lib.dn_conv2d_wgrad(
            g, sv["src"], hip.ptr(self.grad(u.conv + ".weight")), hip.ptr(ws["wgrad"]), ws["wgrad"].numel(), n, h, w, kh, kw,
            u.cin, u.split, self.dtype, hip.stream_ptr()))))
        if not want_dgrad:
            return None
        cin_stored = sv["src"].channels
        gin = PlaneBuffer(n, self._pp(cin_stored), h, w, self.dtype, self.device, arena=A)
        gin.bn_sums = None
        if feeds is not None and self.fuse_bn_sums and (kh, kw) == (3, 5) and feeds.cout == cin_stored:
            fv = self._saved[feeds.conv]
            self._bn_sums_buffer(feeds.cout, n * ((h + 2) // 3) * ((w + 127) // 128))   # tiles are at least 3 rows x 128 columns
            ntiles = ctypes.c_int(0)
            self._timed(lambda: self.conv_kernel_tag(u.cout, cin_stored, n, h, w) + "+bnsums:dgrad", flops,
                        lambda: hip.check(hip.lib.dn_conv2d_dgrad_bn_sums(
                            g, hip.ptr(cache[u.conv]["wd"]), gin.ref(cin_stored), fv["y"].ref(feeds.cout), hip.ptr(fv["scale"]),
                            hip.ptr(fv["shift"]), hip.ptr(fv["mean"]), hip.ptr(fv["invstd"]), n, h, w, kh, kw, self.dtype,
                            hip.ptr(ws["bnsums"]), ws["bnsums"].numel(), ctypes.byref(ntiles), hip.stream_ptr())))
            gin.bn_sums = (ws["bnsums"], ntiles.value)
            return gin
        self._timed(lambda: self.conv_kernel_tag(u.cout, cin_stored, n, h, w) + ":dgrad", flops, lambda: hip.check(hip.lib.dn_conv2d(
            g, hip.ptr(cache[u.conv]["wd"]), None, gin.ref(cin_stored), n, h, w, kh, kw, 0, self.dtype, hip.stream_ptr())))
        return gin

    def backward(self, dlogits: Optional[torch.Tensor] = None, sums: Optional[torch.Tensor] = None, segment_hook=None):
        """Backward of Dice(logits, targets) (or of a given dL/dlogits) through the whole net into the
        flat gradient arena.  ``sums`` lets a data-parallel caller pass all-reduced Dice sums.
        ``segment_hook(name)`` is called after the gradients of each stage are complete (bucketed all-reduce)."""
        S = self._saved
        if S is None:
            raise RuntimeError("backward() needs a preceding train-mode forward")
        ws, A = self._workspace(), S["arena"]
        n, h, w = S["n"], S["h"], S["w"]
        depth, f, dt, dev = self.depth, self.num_filters, self.dtype, self.device
        c = [f * 2 ** l for l in range(depth + 1)]
        wl = [w >> l for l in range(depth + 1)]
        units = {u.conv: u for u in self.conv_units}
        if dlogits is None:
            sums = S["sums"] if sums is None else sums
            dlogits = A.take(tuple(S["logits"].shape), S["logits"].dtype)
            hip.check(hip.lib.dn_dice_backward(hip.ptr(S["logits"]), hip.ptr(S["targets"]), dlogits.numel(), hip.ptr(sums),
                                               hip.ptr(dlogits), hip.stream_ptr()))
        g = PlaneBuffer(n, self._pp(c[0]), h, w, dt, dev, arena=A)
        # (the producers of an activation gradient form BatchNorm backward's sums of the unit it belongs to: _unit_backward)
        last = units[f"decoders.{depth - 1}.conv.double_conv.3"]
        up_sums = None                      # sums of the unit the next _unit_backward call handles, from its producer
        if self.fuse_bn_sums:
            fv, sums, nblk = S[last.conv], self._bn_sums_buffer(last.cout), ctypes.c_int(0)
            hip.check(hip.lib.dn_head_1x1_backward_bn_sums(
                hip.ptr(dlogits), S["head_in"], hip.ptr(self._train_cache["head_w"]), g.ref(c[0]), hip.ptr(self.grad("outputs.weight")),
                hip.ptr(self.grad("outputs.bias")), hip.ptr(ws["partial"]), fv["y"].ref(last.cout), hip.ptr(fv["scale"]),
                hip.ptr(fv["shift"]), hip.ptr(fv["mean"]), hip.ptr(fv["invstd"]), n, h, w, dt, hip.ptr(sums), sums.numel(),
                ctypes.byref(nblk), hip.stream_ptr()))
            up_sums = (sums, nblk.value)
        else:
            hip.check(hip.lib.dn_head_1x1_backward(hip.ptr(dlogits), S["head_in"], hip.ptr(self._train_cache["head_w"]), g.ref(c[0]),
                                                   hip.ptr(self.grad("outputs.weight")), hip.ptr(self.grad("outputs.bias")),
                                                   hip.ptr(ws["partial"]), n, h, w, dt, hip.stream_ptr()))
        if segment_hook:
            self._segment_done(segment_hook, "head")
        gcat = [None] * depth
        cur = g
        # single-stream backward (profiling, big batches): a gradient buffer whose readers are all queued goes back to the
        # arena for a later layer of this pass; with the weight gradients on a second stream nothing is shared
        done = (lambda buf: A.release(buf.data)) if self._side() is None else (lambda buf: None)
        for i in reversed(range(depth)):
            l = depth - 1 - i
            pre = f"decoders.{i}.conv.double_conv"
            g1 = self._unit_backward(units[pre + ".3"], cur.ref(c[l]), True, feeds=units[pre + ".0"], sums=up_sums)
            done(cur)
            gcat[l] = self._unit_backward(units[pre + ".0"], g1.ref(c[l]), True, sums=g1.bn_sums)
            done(g1)
            d_up = gcat[l].ref(c[l], 0)
            upn = f"decoders.{i}.up"
            xin = S[f"up{i}_in"]
            self._launch_wgrad(lambda: hip.check(hip.lib.dn_conv_transpose_1x2_wgrad(
                d_up, xin, hip.ptr(self.grad(upn + ".weight")), hip.ptr(ws["wgrad"]), ws["wgrad"].numel(), n, h, wl[l + 1], dt,
                hip.stream_ptr())))
            hip.check(hip.lib.dn_channel_sum(d_up, hip.ptr(self.grad(upn + ".bias")), hip.ptr(ws["partial"]), n, h, wl[l], dt,
                                             hip.stream_ptr()))
            gx = PlaneBuffer(n, self._pp(c[l + 1]), h, wl[l + 1], dt, dev, arena=A)
            # gx is the activation gradient of the second unit of the block below (previous decoder, or the bottleneck)
            below = units[(f"decoders.{i - 1}.conv" if i > 0 else "bottleneck") + ".double_conv.3"]
            up_sums = None
            if self.fuse_bn_sums and self.fuse_bn_sums_convt:
                fv, nt = S[below.conv], ctypes.c_int(0)
                sums = self._bn_sums_buffer(below.cout, n * ((h * wl[l + 1] + 127) // 128 + h))
                hip.check(hip.lib.dn_conv_transpose_1x2_dgrad_bn_sums(
                    d_up, hip.ptr(self._train_cache[upn]["wd"]), gx.ref(c[l + 1]), fv["y"].ref(below.cout), hip.ptr(fv["scale"]),
                    hip.ptr(fv["shift"]), hip.ptr(fv["mean"]), hip.ptr(fv["invstd"]), n, h, wl[l + 1], dt, hip.ptr(sums), sums.numel(),
                    ctypes.byref(nt), hip.stream_ptr()))
                up_sums = (sums, nt.value)
            else:
                hip.check(hip.lib.dn_conv_transpose_1x2_dgrad(d_up, hip.ptr(self._train_cache[upn]["wd"]), gx.ref(c[l + 1]), n, h,
                                                              wl[l + 1], dt, hip.stream_ptr()))
            cur = gx
            if segment_hook:
                self._segment_done(segment_hook, f"decoders.{i}")
        pre = "bottleneck.double_conv"
        g1 = self._unit_backward(units[pre + ".3"], cur.ref(c[depth]), True, feeds=units[pre + ".0"], sums=up_sums)
        done(cur)
        gpool = self._unit_backward(units[pre + ".0"], g1.ref(c[depth]), True, sums=g1.bn_sums)
        done(g1)
        if segment_hook:
            self._segment_done(segment_hook, "bottleneck")
        for l in reversed(range(depth)):
            pre = f"encoders.{l}.conv.double_conv"
            skip = S["cat"][l].ref(c[l], self._pp(c[l]))
            g2 = PlaneBuffer(n, self._pp(c[l]), h, wl[l], dt, dev, arena=A)
            pool_sums = None
            if self.fuse_bn_sums:
                u3 = units[pre + ".3"]
                fv, sums, nblk = S[u3.conv], self._bn_sums_buffer(u3.cout), ctypes.c_int(0)
                hip.check(hip.lib.dn_maxpool_1x2_backward_bn_sums(
                    skip, gpool.ref(c[l]), gcat[l].ref(c[l], self._pp(c[l])), g2.ref(c[l]), fv["y"].ref(u3.cout), hip.ptr(fv["scale"]),
                    hip.ptr(fv["shift"]), hip.ptr(fv["mean"]), hip.ptr(fv["invstd"]), n, h, wl[l + 1], dt, hip.ptr(sums), sums.numel(),
                    ctypes.byref(nblk), hip.stream_ptr()))
                pool_sums = (sums, nblk.value)
            else:
                hip.check(hip.lib.dn_maxpool_1x2_backward(skip, gpool.ref(c[l]), gcat[l].ref(c[l], self._pp(c[l])), g2.ref(c[l]), n, h,
                                                          wl[l + 1], dt, hip.stream_ptr()))
            done(gpool)
            done(gcat[l])
            g1 = self._unit_backward(units[pre + ".3"], g2.ref(c[l]), True, feeds=units[pre + ".0"], sums=pool_sums)
            done(g2)
            gpool = self._unit_backward(units[pre + ".0"], g1.ref(c[l]), l > 0, sums=g1.bn_sums)
            done(g1)
            if segment_hook:
                self._segment_done(segment_hook, f"encoders.{l}")
        if self._side() is not None:
            torch.cuda.current_stream().wait_stream(self._wgrad_stream)     # Adam (and the caller) see complete gradients
        self._saved = None


class HipAdam:
    """``torch.optim.Adam(params, lr, weight_decay)`` with coupled L2 (reference
    ``trainable_unet.py:155-157``) as one fused kernel over the model's flat parameter arena;
    ``zero_grad`` (``:344``) is folded into the same pass."""

    def __init__(self, model: HipUNet, lr: float = 1e-3, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model = model
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": betas, "eps": eps}]
        self.exp_avg = torch.zeros_like(model.flat_params)
        self.exp_avg_sq = torch.zeros_like(model.flat_params)
        self.step_count = 0
        self._zeroed = False

    def step(self, grad_scale: float = 1.0, device_state: Optional[Sequence[torch.Tensor]] = None):
        """``device_state = (hyper, step)``: learning rate, bias corrections and step count are read from / kept in
        device memory (``dn_adam_step_graph``), which is what lets ``GraphedTrainStep`` replay this launch."""
        g = self.param_groups[0]
        self.step_count += 1
        m = self.model
        if device_state is None:
            hip.check(hip.lib.dn_adam_step(hip.ptr(m.flat_params), hip.ptr(m.flat_grads), hip.ptr(self.exp_avg),
                                           hip.ptr(self.exp_avg_sq), m.flat_params.numel(), g["lr"], g["betas"][0], g["betas"][1],
                                           g["eps"], g["weight_decay"], self.step_count, grad_scale, 1, hip.stream_ptr()))
        else:
            hyper, step = device_state
            hip.check(hip.lib.dn_adam_step_graph(hip.ptr(m.flat_params), hip.ptr(m.flat_grads), hip.ptr(self.exp_avg),
                                                 hip.ptr(self.exp_avg_sq), m.flat_params.numel(), hip.ptr(hyper), hip.ptr(step),
                                                 g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], grad_scale, 1,
                                                 hip.stream_ptr()))
        m._param_epoch += 1
        self._zeroed = True

    def zero_grad(self):
        if not self._zeroed:
            self.model.flat_grads.zero_()
        self._zeroed = False


class GraphedTrainStep:
    """One whole training step -- weight re-packing, train-mode forward, Dice, backward, Adam, zero_grad: ~330 kernel
    launches for the shipped net -- captured once into a HIP graph and replayed.  At train.py's batch of 16 the step is
    otherwise bound by the host issuing those launches (trainable_unet.py:339-347 is the loop being replaced).

    Batch shape, betas / eps / weight decay are frozen into the graph (``matches`` tells whether a batch fits); the
    learning rate and the Adam step count live in device memory, so ExponentialLR keeps working.  The buffers of the
    step (its own ``StepArena``) stay allocated in the graph's private pool (~110 MB per bf16 profile)."""

    def __init__(self, model: "HipUNet", optimizer: HipAdam, x: torch.Tensor, t: torch.Tensor):
        self.model, self.optimizer = model, optimizer
        g = optimizer.param_groups[0]
        self._frozen = (tuple(x.shape), tuple(g["betas"]), g["eps"], g["weight_decay"])
        self.x = x.detach().clone().contiguous()
        self.t = t.detach().to(torch.int8).contiguous().clone()
        self.hyper = torch.zeros(4, dtype=torch.float32, device=model.device)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=model.device)
        self._lr_dev, self._step_expected = None, None
        model.train()
        model._workspace()
        model._build_train_cache()                         # job table and image buffers exist before the capture ...
        model._train_cache_version = None                  # ... and the re-packing launch itself is part of the graph
        self.graph = torch.cuda.CUDAGraph()
        saved_count, saved_epoch = optimizer.step_count, model._param_epoch
        torch.cuda.synchronize()
        # the captured step owns its arena (allocated from the graph's private pool during the capture): the addresses
        # the graph replays on stay valid whatever the model's own arenas do afterwards
        self.arena = StepArena(model.device)
        model._arena_override = self.arena
        try:
            with torch.cuda.graph(self.graph):
                logits = model.forward_train(self.x)
                self.sums = model.dice_forward(logits, self.t)
                model.backward()
                optimizer.step(device_state=(self.hyper, self.step_dev))
                optimizer.zero_grad()
        finally:
            model._arena_override = None
        # capture executed nothing: undo the host-side bookkeeping it did
        optimizer.step_count, model._param_epoch = saved_count, saved_epoch
        self._cache = model._train_cache                   # the images the graph writes: keep them alive
        model._train_cache_version = None

    def matches(self, x: torch.Tensor) -> bool:
        g = self.optimizer.param_groups[0]
        return self._frozen == (tuple(x.shape), tuple(g["betas"]), g["eps"], g["weight_decay"])

    def __call__(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """Runs one step on (x, t); returns a snapshot of the Dice sums vector (``HipUNet.dice_forward``)."""
        if not self.matches(x):
            raise ValueError("batch shape or optimizer constants differ from the captured step")
        opt, m = self.optimizer, self.model
        lr = float(opt.param_groups[0]["lr"])
        if lr != self._lr_dev:
            self.hyper[0:1].fill_(lr)
            self._lr_dev = lr
        if self._step_expected != opt.step_count:          # someone stepped the optimizer outside the graph
            self.step_dev.fill_(opt.step_count)
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(t, non_blocking=True)
        self.graph.replay()
        opt.step_count += 1
        self._step_expected = opt.step_count
        m._param_epoch += 2                                 # running statistics and parameters were rewritten
        m._saved = None
        return self.sums.clone()


class ExponentialLR:
    """``torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma)`` (``trainable_unet.py:159-162``)."""

    def __init__(self, optimizer: HipAdam, gamma: float = 0.8):
        self.optimizer, self.gamma = optimizer, gamma

    def step(self):
        for g in self.optimizer.param_groups:
            g["lr"] *= self.gamma
