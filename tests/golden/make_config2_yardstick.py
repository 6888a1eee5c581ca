"""Yardstick of tests/test_config2_fit_gpu.py: how far two correct fp32 runs of configs[2] may be apart.

Runs the oracle (oracle/unet_oracle.py, the restatement of the reference's loop that tests/test_oracle_golden.py pins to
the reference's own modules) for one epoch of ``fit()`` at configs[2]'s size -- shipped net, 348 synthetic 5 x 4096 profiles,
batch 16, 21 steps, lr 1e-4, weight decay 5e-4 -- twice: in fp32 (the reference's arithmetic) and in fp64.  Stored:
the fp64 loss trajectory and BatchNorm running statistics, and per tensor the distance of the fp32 run from the fp64 run
(relative l2 error of the 21-step parameter UPDATE; running statistics relative to the tensor's scale).  The GPU test
holds the HIP fp32 path to "as close to torch's fp32 as torch's fp32 is to fp64".  ~8 minutes on 8 cores.

    python tests/golden/make_config2_yardstick.py
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dnanet_amd.data import SyntheticDataset  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

SEED_DATA, SEED_WEIGHTS, BATCH, LR, WD = 3, 11, 16, 1e-4, 5e-4


def loop(sd0, images, dt):
    sd = OrderedDict((k, v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items())
    state = O.AdamState(O.trainable_keys(sd), sd)
    losses = []
    for s in range(len(images) // BATCH):
        batch = images[BATCH * s:BATCH * s + BATCH]
        x = torch.stack([torch.from_numpy(im.data.astype("float32")).permute(2, 0, 1) for im in batch]).to(dt)
        t = torch.stack([torch.from_numpy(im.annotation.image).permute(2, 0, 1) for im in batch])
        loss, _, _ = O.train_step(sd, x, t, state, O.exponential_lr(LR, 0), WD)
        losses.append(float(loss))
        print(dt, s, losses[-1], flush=True)
    return sd, losses


def main():
    images = list(SyntheticDataset(348, 5, 4096, seed=SEED_DATA))
    sd0 = O.init_state_dict(4, (3, 5), 32, seed=SEED_WEIGHTS)
    sd32, loss32 = loop(sd0, images, torch.float32)
    sd64, loss64 = loop(sd0, images, torch.float64)
    out = {"loss64": np.array(loss64), "loss32": np.array(loss32)}
    keys, upd, stat_keys, stat_err = [], [], [], []
    for k, v64 in sd64.items():
        if v64.dtype == torch.int64:
            continue
        if "running" in k:
            out["stats64/" + k] = v64.numpy()
            stat_keys.append(k)
            stat_err.append(float((sd32[k].double() - v64).abs().max() / v64.abs().max()))
        elif not (k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias")):
            ref = (v64 - sd0[k].double()).flatten()
            keys.append(k)
            upd.append(float(((sd32[k].double() - sd0[k].double()).flatten() - ref).norm() / ref.norm()))
            out["update_norm64/" + k] = np.array(float(ref.norm()))
    out["update_keys"], out["update_err_fp32_vs_fp64"] = np.array(keys), np.array(upd)
    out["stats_keys"], out["stats_err_fp32_vs_fp64"] = np.array(stat_keys), np.array(stat_err)
    out["config"] = np.array([SEED_DATA, SEED_WEIGHTS, BATCH, 348, 21])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config2_fit_yardstick.npz"), **out)
    print("worst update error fp32 vs fp64", max(upd), "worst stats error", max(stat_err), "loss dev", max(abs(a - b) for a, b in zip(loss32, loss64)))


if __name__ == "__main__":
    main()
