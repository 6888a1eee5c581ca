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

import torch

from . import hip
from .hip import BF16, F32, PlaneBuffer
from .params import ConvUnit, ParameterTree, UpUnit

BN_EPS = 1e-5       # nn.BatchNorm2d default used by the reference (unet_architecture.py:17,21)
BN_MOMENTUM = 0.1

_DTYPES = {"fp32": F32, "float32": F32, "f32": F32, "bf16": BF16, "bfloat16": BF16}


def parse_dtype(dtype) -> int:
    if isinstance(dtype, int):
        return dtype
    try:
        return _DTYPES[str(dtype).lower()]
    except KeyError:
        raise ValueError(f"dtype must be one of {sorted(_DTYPES)}, got {dtype!r}")


class HipUNet(ParameterTree):
    """Parameter tree + HIP execution.  ``forward`` takes/returns fp32 NCHW tensors on the GPU."""

    MAX_CHUNK = 256     # profiles per inference pass (eval-mode BN is batch independent)

    def __init__(self, depth: int, kernel_size: Sequence[int], num_filters: int, device, dtype="fp32"):
        super().__init__(depth, kernel_size, num_filters, device)
        if num_filters % 8:
            raise ValueError("num_filters must be a multiple of 8 for the MI355X kernels")
        self.dtype = parse_dtype(dtype)
        self._eval_cache = None
        self._eval_cache_version = None
        self.profile = None     # set to a list to collect (kernel tag, flops, start event, end event) per conv launch

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.flat_params.device

    def _require_gpu(self, x: torch.Tensor):
        if not (self.device.type == "cuda" and x.is_cuda):
            raise RuntimeError("dnanet_amd runs on an MI355X only (no CPU fallback): model and input must be on 'cuda'")

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
        gs = hip.channel_pad(b.numel())
        out = torch.zeros(2 * gs, dtype=torch.float32, device=self.device)
        out[:b.numel()] = b
        out[gs:gs + b.numel()] = b
        return out

    # ------------------------------------------------------------------------------------------
    # eval-mode (BatchNorm folded into the packed weights)
    # ------------------------------------------------------------------------------------------
    def _versions(self):
        return (self.flat_params._version, self.flat_stats._version, self.dtype)

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
            else:
                entry["w"] = self._pack(self._virtual_weight(u, w), scale.contiguous(), hip.PACK_FWD)
            cache[u.conv] = entry
        for up in self.up_units:
            cache[up.name] = {"w": self._pack(self.tensor(up.name + ".weight"), None, hip.PACK_CONVT_FWD),
                              "bias": self._up_bias2(self.tensor(up.name + ".bias"))}
        cache["head_w"] = self._padvec(self.tensor("outputs.weight"))
        cache["head_b"] = self.tensor("outputs.bias").clone()
        self._eval_cache = cache
        self._eval_cache_version = self._versions()

    def conv_kernel_tag(self, cout: int, h: int) -> str:
        """Name of the conv_igemm instantiation the library dispatches to (api.hip: run_igemm)."""
        kh, kw = self.kernel_size
        cot = 64 if hip.channel_pad(cout) % 64 == 0 else 32
        ht = 6 if ((kh, kw) == (3, 5) and h % 6 == 0 and h % 5 != 0) else 5
        return f"conv_igemm<{'bf16' if self.dtype == BF16 else 'f32'},{kh}x{kw},ht{ht},cot{cot}>"

    def _conv(self, u: ConvUnit, entry, src, dst, n, h, w, relu=1):
        if self.profile is not None and u.cin != 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._conv_launch(u, entry, src, dst, n, h, w, relu)
            e1.record()
            kh, kw = self.kernel_size
            self.profile.append((self.conv_kernel_tag(u.cout, h), 2.0 * n * h * w * u.cout * u.cin * kh * kw, e0, e1))
        else:
            self._conv_launch(u, entry, src, dst, n, h, w, relu)

    def _conv_launch(self, u: ConvUnit, entry, src, dst, n, h, w, relu=1):
        kh, kw = self.kernel_size
        if u.cin == 1:
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
        cat = [PlaneBuffer(n, 2 * self._pp(c[l]), h, ws[l], dt, dev) for l in range(depth)]
        cur = x
        for l in range(depth):
            pre = f"encoders.{l}.conv.double_conv"
            tmp = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev)
            self._conv(units[pre + ".0"], cache[pre + ".0"], cur if l == 0 else cur.ref(c[l - 1]), tmp.ref(c[l]), n, h, ws[l])
            skip = cat[l].ref(c[l], self._pp(c[l]))
            self._conv(units[pre + ".3"], cache[pre + ".3"], tmp.ref(c[l]), skip, n, h, ws[l])
            pooled = PlaneBuffer(n, self._pp(c[l]), h, ws[l + 1], dt, dev)
            hip.check(hip.lib.dn_maxpool_1x2(skip, pooled.ref(c[l]), n, h, ws[l + 1], dt, hip.stream_ptr()))
            cur = pooled
        pre = "bottleneck.double_conv"
        tmp = PlaneBuffer(n, self._pp(c[depth]), h, ws[depth], dt, dev)
        self._conv(units[pre + ".0"], cache[pre + ".0"], cur.ref(c[depth - 1]), tmp.ref(c[depth]), n, h, ws[depth])
        out = PlaneBuffer(n, self._pp(c[depth]), h, ws[depth], dt, dev)
        self._conv(units[pre + ".3"], cache[pre + ".3"], tmp.ref(c[depth]), out.ref(c[depth]), n, h, ws[depth])
        cur = out
        for i in range(depth):
            l = depth - 1 - i
            up = cache[f"decoders.{i}.up"]
            hip.check(hip.lib.dn_conv_transpose_1x2(cur.ref(c[l + 1]), hip.ptr(up["w"]), hip.ptr(up["bias"]),
                                                    cat[l].ref(c[l], 0), n, h, ws[l + 1], dt, hip.stream_ptr()))
            pre = f"decoders.{i}.conv.double_conv"
            tmp = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev)
            both = cat[l].ref(2 * self._cpad_planes(c[l]), 0)
            self._conv(units[pre + ".0"], cache[pre + ".0"], both, tmp.ref(c[l]), n, h, ws[l])
            out = PlaneBuffer(n, self._pp(c[l]), h, ws[l], dt, dev)
            self._conv(units[pre + ".3"], cache[pre + ".3"], tmp.ref(c[l]), out.ref(c[l]), n, h, ws[l])
            cur = out
        hip.check(hip.lib.dn_head_1x1(cur.ref(c[0]), hip.ptr(cache["head_w"]), hip.ptr(cache["head_b"]),
                                      hip.ptr(logits), hip.ptr(probs), n, h, w, dt, hip.stream_ptr()))

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
            self._forward_eval_chunk(x[s:e], logits[s:e] if want_logits else None, probs[s:e] if want_probs else None)
        return logits, probs

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("train-mode forward goes through HipUNet.train_step()/forward_train()")
        return self.infer(x, want_logits=True, want_probs=False)[0]
