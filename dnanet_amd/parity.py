"""Counting how far two runs of the hot path disagree where it matters to the caller: thresholded masks
(``p >= 0.5``, ``DNAnet/allele_callers.py:80``, ``pixel_metrics.py:17-36``) and called alleles
(``NearestBasePairCaller``, ``allele_callers.py:63-125``).  Used by ``bench.py`` (bf16 kernels vs fp32 kernels) and by the
GPU parity tests (kernels vs the fp32 CPU oracle); nothing here computes a reference by itself.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .allele_caller import NearestBasePairCaller
from .data import ArrayImage, Panel
from .prediction import Prediction

DEFAULT_PANEL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "panel.xml")


def called_allele_sets(probs: torch.Tensor, x: torch.Tensor, panel: Panel, scaler: Optional[np.ndarray] = None) -> list:
    """Per profile the set {(dye row, marker, allele)} the caller derives from (N,1,H,W) probabilities on the device."""
    n, _, h, w = probs.shape
    scaler = np.linspace(65, 475, w) if scaler is None else scaler          # the reference's base-pair axis (data/utils.py:11)
    images = [ArrayImage(data=None, _scaler=scaler, _panel=panel) for _ in range(n)]
    preds = [Prediction(image=None, meta={}) for _ in range(n)]
    NearestBasePairCaller().call_alleles_batch(images, preds, probs.contiguous(), x.contiguous())
    return [{(m.dye_row, m.name, a.name) for m in p.meta["called_alleles"] for a in m.alleles} for p in preds]


def disagreement(probs_a: torch.Tensor, probs_b: torch.Tensor, x: torch.Tensor, panel: Optional[Panel] = None) -> Dict[str, float]:
    """Differing mask pixels and differing called alleles between two (N,1,H,W) probability tensors of the same inputs
    ``x`` (raw RFU, used for peak heights only).  ``panel``: a ``Panel`` (default: the kit table of the reference's tests)."""
    dev = probs_a.device
    probs_b = probs_b.to(dev)
    ma, mb = probs_a >= 0.5, probs_b >= 0.5
    out = {
        "profiles": int(probs_a.shape[0]),
        "pixels": int(probs_a.numel()),
        "mask_pixels_differing": int((ma != mb).sum()),
        "mask_positive_pixels": int(mb.sum()),
        "max_abs_prob_diff": float((probs_a - probs_b).abs().max()),
    }
    if panel is None and os.path.isfile(DEFAULT_PANEL):
        panel = Panel(DEFAULT_PANEL)
    if panel is not None and probs_a.shape[2] <= 5:         # the kit table describes 5 dye rows
        sa, sb = called_allele_sets(probs_a, x, panel), called_allele_sets(probs_b, x, panel)
        out["alleles_called"] = int(sum(len(s) for s in sb))
        out["alleles_differing"] = int(sum(len(a ^ b) for a, b in zip(sa, sb)))
        out["profiles_with_allele_difference"] = int(sum(1 for a, b in zip(sa, sb) if a != b))
    return out


def briefly_trained_state(depth: int, kernel_size: Sequence[int], num_filters: int, device, rows: int = 5, width: int = 4096,
                          steps: int = 60, batch: int = 16, lr: float = 2e-3, seed: int = 0):
    """A seeded net of the given shape after a short fp32 training run on synthetic electropherograms (the pretrained
    checkpoint of the reference is a Git-LFS pointer here): its masks are sparse peaks like a real model's, which is the
    regime in which mask / allele disagreement between precisions means something.  Returns a CPU state_dict."""
    from .synthetic import make_batch
    from .unet import HipAdam, HipUNet
    torch.manual_seed(seed)
    m = HipUNet(depth, tuple(kernel_size), num_filters, device, dtype="fp32")
    m.train()
    opt = HipAdam(m, lr=lr, weight_decay=5e-4)
    opt.zero_grad()
    x_np, t_np = make_batch(batch * 4, rows, width, seed=seed + 77, unique=batch * 4)
    x, t = torch.from_numpy(x_np).to(device), torch.from_numpy(t_np).to(device)
    for s in range(steps):
        i = (s % 4) * batch
        m.dice_forward(m(x[i:i + batch]), t[i:i + batch])
        m.backward()
        opt.step()
        opt.zero_grad()
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
