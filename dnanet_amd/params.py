"""Parameter tree of the U-Net with the reference's ``state_dict`` wire format.

The reference builds its parameters out of ``torch.nn`` modules
(``DNAnet/models/segmentation/unet_architecture.py:67-97``); ``DNANet_UNet_checkpoint.pt`` is the
``state_dict`` of that tree (``trainable_unet.py:447-470``), so key names, order, shapes and dtypes
are a wire format.  Here the same keys are registered on plain container
modules whose tensors are *views into flat device arenas* -- one fp32 arena for the trainable
parameters (with matching gradient / Adam arenas), one for the BatchNorm running statistics --
which is what the HIP kernels (fused Adam, bucketed all-reduce, weight packing) operate on.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch
from torch import nn


@dataclass
class ConvUnit:
    """One Conv2d + BatchNorm2d + ReLU of a DoubleConv (unet_architecture.py:14-21)."""
    conv: str      # state_dict prefix of the conv, e.g. "encoders.0.conv.double_conv.0"
    bn: str        # prefix of its BatchNorm
    cin: int
    cout: int
    level: int     # resolution level: W / 2**level
    split: int     # >0: the input is the skip-concat [up | skip] of `split` channels each


@dataclass
class UpUnit:
    """ConvTranspose2d(k=(1,2), s=(1,2)) of a DecoderBlock (unet_architecture.py:54-57)."""
    name: str
    cin: int
    cout: int
    level: int     # level of its OUTPUT


def unet_layout(depth: int, kernel_size: Sequence[int], num_filters: int):
    """Ordered parameter spec [(key, shape, kind)] and the unit lists, in module order."""
    kh, kw = int(kernel_size[0]), int(kernel_size[1])
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    convs: List[ConvUnit] = []
    ups: List[UpUnit] = []

    def double_conv(prefix, cin, cout, level, split):
        for ci_idx, bn_idx, c_in, sp in ((0, 1, cin, split), (3, 4, cout, 0)):
            c, b = f"{prefix}.double_conv.{ci_idx}", f"{prefix}.double_conv.{bn_idx}"
            spec.append((c + ".weight", (cout, c_in, kh, kw), "param"))
            spec.append((c + ".bias", (cout,), "param"))
            spec.append((b + ".weight", (cout,), "param"))
            spec.append((b + ".bias", (cout,), "param"))
            spec.append((b + ".running_mean", (cout,), "stat"))
            spec.append((b + ".running_var", (cout,), "stat"))
            spec.append((b + ".num_batches_tracked", (), "count"))
            convs.append(ConvUnit(c, b, c_in, cout, level, sp))

    cin = 1
    for i in range(depth):
        cout = num_filters * 2 ** i
        double_conv(f"encoders.{i}.conv", cin, cout, i, 0)
        cin = cout
    cout = num_filters * 2 ** depth
    double_conv("bottleneck", cin, cout, depth, 0)
    cin = cout
    for i in range(depth):
        cout = num_filters * 2 ** (depth - 1 - i)
        level = depth - 1 - i
        spec.append((f"decoders.{i}.up.weight", (cin, cout, 1, 2), "param"))
        spec.append((f"decoders.{i}.up.bias", (cout,), "param"))
        ups.append(UpUnit(f"decoders.{i}.up", cin, cout, level))
        double_conv(f"decoders.{i}.conv", 2 * cout, cout, level, cout)
        cin = cout
    spec.append(("outputs.weight", (1, cin, 1, 1), "param"))
    spec.append(("outputs.bias", (1,), "param"))
    return spec, convs, ups


class _Node(nn.Module):
    """Name-only container (the arithmetic lives in the HIP kernels)."""


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class ParameterTree(nn.Module):
    """Registers every state_dict entry as a view into a flat arena, on ``device``."""

    def __init__(self, depth: int, kernel_size: Sequence[int], num_filters: int, device):
        super().__init__()
        self.depth, self.kernel_size, self.num_filters = depth, (int(kernel_size[0]), int(kernel_size[1])), num_filters
        spec, self.conv_units, self.up_units = unet_layout(depth, kernel_size, num_filters)
        self.spec = spec
        n_param = sum(_numel(s) for _, s, k in spec if k == "param")
        n_stat = sum(_numel(s) for _, s, k in spec if k == "stat")
        n_count = sum(1 for _, _, k in spec if k == "count")
        dev = torch.device(device)
        # 256-element alignment of every tensor start keeps 16-byte vector access legal
        self.offsets = {}
        off = {"param": 0, "stat": 0, "count": 0}
        for key, shape, kind in spec:
            self.offsets[key] = (kind, off[kind], _numel(shape), shape)
            off[kind] += (_numel(shape) + 3) // 4 * 4 if kind != "count" else 1
        self.flat_params = torch.zeros(off["param"], dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(off["param"], dtype=torch.float32, device=dev)
        self.flat_stats = torch.zeros(off["stat"], dtype=torch.float32, device=dev)
        self.flat_counts = torch.zeros(off["count"], dtype=torch.int64, device=dev)
        self.n_param, self.n_stat, self.n_count = n_param, n_stat, n_count
        self._by_key = {}
        for key, shape, kind in spec:
            *path, leaf = key.split(".")
            node = self
            for part in path:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            _, o, n, _ = self.offsets[key]
            if kind == "param":
                p = nn.Parameter(self.flat_params[o:o + n].view(shape), requires_grad=True)
                p.grad = self.flat_grads[o:o + n].view(shape)
                node.register_parameter(leaf, p)
                self._by_key[key] = p
            elif kind == "stat":
                t = self.flat_stats[o:o + n].view(shape)
                node.register_buffer(leaf, t)
                self._by_key[key] = t
            else:
                t = self.flat_counts[o:o + 1].view(())
                node.register_buffer(leaf, t)
                self._by_key[key] = t
        self.reset_parameters()

    def tensor(self, key: str) -> torch.Tensor:
        return self._by_key[key]

    def grad(self, key: str) -> torch.Tensor:
        _, o, n, shape = self.offsets[key]
        return self.flat_grads[o:o + n].view(shape)

    @torch.no_grad()
    def reset_parameters(self, generator: torch.Generator = None):
        """PyTorch's default initialisation of the reference's modules:
        conv / convT weight and bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BN gamma=1, beta=0,
        running_mean=0, running_var=1."""
        fan_in = 1
        for key, shape, kind in self.spec:
            t = self._by_key[key]
            if kind == "count":
                t.zero_()
            elif key.endswith("running_mean"):
                t.zero_()
            elif key.endswith("running_var"):
                t.fill_(1.0)
            elif len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
                bound = 1.0 / fan_in ** 0.5
                t.copy_((torch.rand(shape, generator=generator) * 2 - 1) * bound)
            elif ".double_conv.1." in key or ".double_conv.4." in key:
                t.fill_(1.0 if key.endswith("weight") else 0.0)
            else:  # conv / up / head bias follows its weight
                bound = 1.0 / fan_in ** 0.5
                t.copy_((torch.rand(shape, generator=generator) * 2 - 1) * bound)

    def _apply(self, fn, recurse=True):
        # The tensors are arena views; moving them one by one would break the sharing.
        probe = fn(torch.empty(0, device=self.flat_params.device))
        if probe.device != self.flat_params.device or probe.dtype != torch.float32:
            raise RuntimeError("dnanet_amd's U-Net lives on the device it was created on in fp32; "
                               "construct it with the target device instead of calling .to()/.half()")
        return self
