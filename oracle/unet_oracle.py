"""CPU oracle for DNANet's U-Net hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain ``torch.nn.functional`` fp32 on the CPU, the algorithm of the
reference's U-Net peak-segmentation path.  It is the checker the HIP path is compared with; it
is never the product path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.

Parity pin: ``tests/golden/*.npz`` were generated in the build container by importing the
reference's own modules (``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks
this restatement against them (logits, loss, every gradient, post-Adam parameters, BN running
statistics).  The reference's own test-suite holds no numeric assertion for the U-Net
(``/root/reference/tests/models/segmentation/test_unet.py:12-19`` is a smoke test), so those
fixtures are the pin.

Reference sites restated here (paths relative to ``/root/reference``):

* ``DNAnet/models/segmentation/unet_architecture.py:5-25``   DoubleConv
* ``DNAnet/models/segmentation/unet_architecture.py:28-42``  EncoderBlock (DoubleConv + MaxPool2d((1,2)))
* ``DNAnet/models/segmentation/unet_architecture.py:45-64``  DecoderBlock (ConvTranspose2d (1,2)/(1,2), cat [up, skip])
* ``DNAnet/models/segmentation/unet_architecture.py:67-131`` UNet (module tree / state_dict names, LIFO skips, 1x1 head)
* ``DNAnet/models/loss.py:9-20``                             DiceLoss (sigmoid, whole-batch sums, smooth=1)
* ``DNAnet/models/segmentation/trainable_unet.py:60-89``     tensorisation of images / targets
* ``DNAnet/models/segmentation/trainable_unet.py:155-162``   Adam(lr, weight_decay) + ExponentialLR(gamma=0.8)
* ``DNAnet/models/segmentation/trainable_unet.py:339-347``   step order: fwd, metric, loss, backward, step, zero_grad
* ``DNAnet/models/segmentation/trainable_unet.py:374-409``   predict_batch: eval, no_grad, sigmoid, movedim(0,-1)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default (unet_architecture.py:17,21)
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default


# ----------------------------------------------------------------------------------------------
# state_dict layout -- unet_architecture.py:67-97
# ----------------------------------------------------------------------------------------------
def _double_conv_spec(prefix: str, cin: int, cout: int, kh: int, kw: int):
    out = []
    for conv_idx, bn_idx, ci in ((0, 1, cin), (3, 4, cout)):
        out.append((f"{prefix}.double_conv.{conv_idx}.weight", (cout, ci, kh, kw), "conv_w"))
        out.append((f"{prefix}.double_conv.{conv_idx}.bias", (cout,), "conv_b"))
        out.append((f"{prefix}.double_conv.{bn_idx}.weight", (cout,), "bn_w"))
        out.append((f"{prefix}.double_conv.{bn_idx}.bias", (cout,), "bn_b"))
        out.append((f"{prefix}.double_conv.{bn_idx}.running_mean", (cout,), "bn_rm"))
        out.append((f"{prefix}.double_conv.{bn_idx}.running_var", (cout,), "bn_rv"))
        out.append((f"{prefix}.double_conv.{bn_idx}.num_batches_tracked", (), "bn_nbt"))
    return out


def state_dict_spec(depth: int, kernel_size: Sequence[int], num_filters: int):
    """Ordered (key, shape, kind) list identical to ``UNet(...).state_dict()`` of the reference."""
    kh, kw = kernel_size
    spec = []
    cin = 1
    for i in range(depth):
        cout = num_filters * 2 ** i
        spec += _double_conv_spec(f"encoders.{i}.conv", cin, cout, kh, kw)
        cin = cout
    cout = num_filters * 2 ** depth
    spec += _double_conv_spec("bottleneck", cin, cout, kh, kw)
    cin = cout
    for i in range(depth):
        cout = num_filters * 2 ** (depth - 1 - i)
        spec.append((f"decoders.{i}.up.weight", (cin, cout, 1, 2), "up_w"))
        spec.append((f"decoders.{i}.up.bias", (cout,), "up_b"))
        spec += _double_conv_spec(f"decoders.{i}.conv", 2 * cout, cout, kh, kw)
        cin = cout
    spec.append(("outputs.weight", (1, cin, 1, 1), "head_w"))
    spec.append(("outputs.bias", (1,), "head_b"))
    return spec


def init_state_dict(depth: int, kernel_size: Sequence[int], num_filters: int, seed: int = 0,
                    randomize_bn_stats: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """PyTorch default initialisation: conv/convT weight and bias
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)), BN gamma=1 beta=0 rm=0 rv=1.  The reference seeds nothing,
    so only the distribution has to match."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    fan_in = 1
    for key, shape, kind in state_dict_spec(depth, kernel_size, num_filters):
        if kind in ("conv_w", "head_w"):
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / fan_in ** 0.5
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "up_w":
            fan_in = shape[1] * shape[2] * shape[3]    # ConvTranspose2d: size(1)*kh*kw
            bound = 1.0 / fan_in ** 0.5
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind in ("conv_b", "up_b", "head_b"):
            bound = 1.0 / fan_in ** 0.5
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "bn_w":
            sd[key] = torch.ones(shape) if not randomize_bn_stats else 0.5 + torch.rand(shape, generator=g)
        elif kind == "bn_b":
            sd[key] = torch.zeros(shape) if not randomize_bn_stats else 0.2 * torch.randn(shape, generator=g)
        elif kind == "bn_rm":
            sd[key] = torch.zeros(shape) if not randomize_bn_stats else 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_rv":
            sd[key] = torch.ones(shape) if not randomize_bn_stats else 0.5 + 1.5 * torch.rand(shape, generator=g)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
        else:  # pragma: no cover
            raise AssertionError(kind)
    return sd


def infer_config(sd: Dict[str, torch.Tensor]) -> Tuple[int, Tuple[int, int], int]:
    depth = len({k.split(".")[1] for k in sd if k.startswith("encoders.")})
    w = sd["encoders.0.conv.double_conv.0.weight"]
    return depth, (int(w.shape[2]), int(w.shape[3])), int(w.shape[0])


def trainable_keys(sd: Dict[str, torch.Tensor]) -> List[str]:
    """Parameter order of ``UNet.parameters()`` (= state_dict order minus BN buffers)."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


# ----------------------------------------------------------------------------------------------
# forward -- unet_architecture.py:5-131
# ----------------------------------------------------------------------------------------------
def _double_conv(sd, prefix, x, train):
    for conv_idx, bn_idx in ((0, 1), (3, 4)):
        w = sd[f"{prefix}.double_conv.{conv_idx}.weight"]
        b = sd[f"{prefix}.double_conv.{conv_idx}.bias"]
        kh, kw = int(w.shape[2]), int(w.shape[3])
        x = F.conv2d(x, w, b, stride=1, padding=(kh // 2, kw // 2))          # :14-16,18-20
        bn = f"{prefix}.double_conv.{bn_idx}"
        x = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"],
                         sd[bn + ".weight"], sd[bn + ".bias"],
                         training=train, momentum=BN_MOMENTUM, eps=BN_EPS)   # :17,21
        if train:
            sd[bn + ".num_batches_tracked"] += 1
        x = F.relu(x)
    return x


def unet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, train: bool = False) -> torch.Tensor:
    """``UNet.forward`` (unet_architecture.py:99-131).  ``x``: (N,1,H,W) fp32 -> logits (N,1,H,W).
    In ``train`` mode the BN running statistics inside ``sd`` are updated in place, like the module."""
    depth, _, _ = infer_config(sd)
    skips = []
    for i in range(depth):
        s = _double_conv(sd, f"encoders.{i}.conv", x, train)       # :40
        skips.append(s)
        x = F.max_pool2d(s, (1, 2))                                 # :37,41
    x = _double_conv(sd, "bottleneck", x, train)                    # :114
    for i in range(depth):
        s = skips.pop()                                             # LIFO :122
        x = F.conv_transpose2d(x, sd[f"decoders.{i}.up.weight"], sd[f"decoders.{i}.up.bias"],
                               stride=(1, 2))                       # :54-57
        x = torch.cat([x, s], dim=1)                                # up first, skip second :62
        x = _double_conv(sd, f"decoders.{i}.conv", x, train)
    return F.conv2d(x, sd["outputs.weight"], sd["outputs.bias"])    # :94-97,130


def dice_loss(logits: torch.Tensor, targets: torch.Tensor, smooth: float = 1.0) -> torch.Tensor:
    """``DiceLoss.forward`` (loss.py:9-20): whole-batch sums, smooth=1."""
    p = torch.sigmoid(logits).view(-1)
    t = targets.view(-1)
    inter = (p * t).sum()
    return 1 - (2.0 * inter + smooth) / (p.sum() + t.sum() + smooth)


def binary_accuracy_counts(logits: torch.Tensor, targets: torch.Tensor) -> Tuple[int, int]:
    """torchmetrics BinaryAccuracy(threshold=0.5) as updated at trainable_unet.py:411-415:
    (correct, total) with prediction = sigmoid(z) >= 0.5."""
    pred = (torch.sigmoid(logits).flatten() >= 0.5)
    t = targets.flatten() != 0
    return int((pred == t).sum()), int(t.numel())


# ----------------------------------------------------------------------------------------------
# tensorisation -- trainable_unet.py:60-89, data/utils.py:56-63
# ----------------------------------------------------------------------------------------------
def get_inputs(images: Iterable) -> torch.Tensor:
    """(H,W,1) arrays -> (N,1,H,W) fp32, raw RFU, no normalisation."""
    return torch.stack([torch.tensor(np.transpose(np.asarray(im.data), (2, 0, 1)), dtype=torch.float32)
                        for im in images])


def get_targets(images: Iterable) -> torch.Tensor:
    return torch.stack([torch.tensor(np.asarray(im.annotation.image)).movedim(2, 0) for im in images])


# ----------------------------------------------------------------------------------------------
# optimiser -- torch.optim.Adam (coupled L2) restated; trainable_unet.py:155-157,343-344
# ----------------------------------------------------------------------------------------------
class AdamState:
    def __init__(self, keys: Sequence[str], sd: Dict[str, torch.Tensor]):
        self.step = 0
        self.m = {k: torch.zeros_like(sd[k]) for k in keys}
        self.v = {k: torch.zeros_like(sd[k]) for k in keys}


def adam_step(sd, grads, state: AdamState, lr: float, weight_decay: float = 5e-4,
              betas=(0.9, 0.999), eps: float = 1e-8):
    """One ``torch.optim.Adam`` step (amsgrad off, coupled weight decay) -- SURVEY appendix A."""
    state.step += 1
    t = state.step
    b1, b2 = betas
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    for k, g in grads.items():
        p = sd[k]
        if weight_decay != 0:
            g = g + weight_decay * p
        m = state.m[k].mul_(b1).add_(g, alpha=1 - b1)
        v = state.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def train_step(sd, x: torch.Tensor, targets: torch.Tensor, state: AdamState, lr: float,
               weight_decay: float = 5e-4):
    """fwd (train-mode BN) + Dice + backward + Adam, as trainable_unet.py:339-347.
    Returns (loss, logits, grads) and updates ``sd``/``state`` in place."""
    keys = trainable_keys(sd)
    leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaf)
    logits = unet_forward(work, x, train=True)
    loss = dice_loss(logits, targets)
    glist = torch.autograd.grad(loss, [leaf[k] for k in keys])
    grads = {k: g for k, g in zip(keys, glist)}
    with torch.no_grad():
        adam_step(sd, grads, state, lr, weight_decay)
    return float(loss.detach()), logits.detach(), grads


def predict_probabilities(sd, x: torch.Tensor) -> torch.Tensor:
    """predict_batch's numeric core (trainable_unet.py:374-382): eval, no_grad, sigmoid."""
    with torch.no_grad():
        return torch.sigmoid(unet_forward(sd, x, train=False))


def exponential_lr(lr0: float, epoch: int, gamma: float = 0.8) -> float:
    """ExponentialLR(gamma=0.8) stepped once per epoch (trainable_unet.py:159-162,273-274)."""
    return lr0 * gamma ** epoch
