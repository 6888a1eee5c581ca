"""BASELINE.json configs[2] at its own size: ``fit()`` of the shipped U-Net (depth 4, (3,5), 32 filters) on 348 profiles of
5 x 4096, batch 16, fp32 -- the values of ``config/training/segmentation.yaml`` (lr 1e-4, scheduler on; weight decay is
``fit``'s default 5e-4) -- against the oracle running the reference's loop (``trainable_unet.py:183-203,339-347``):
348 // 16 = 21 steps, the 12-profile tail never trained on, the first step launch by launch and the other 20 replayed
from the HIP graph at the shipped width."""
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fit_348_profiles_batch16_shipped_net_vs_oracle_loop(gpu_device, monkeypatch):
    import dnanet_amd.model as M
    from dnanet_amd.data import SyntheticDataset
    from oracle import unet_oracle as O
    torch.set_num_threads(16)
    ds = SyntheticDataset(348, 5, 4096, seed=3)
    sd0 = O.init_state_dict(4, (3, 5), 32, seed=11)
    model = M.DNANet_UNet(4, (3, 5), 32, device="cuda", apply_allele_caller=False)          # dtype fp32: the parity mode
    model._model.load_state_dict(sd0)

    replays, losses = [], []

    class CountingStep(M.GraphedTrainStep):
        def __call__(self, x, t):
            replays.append(tuple(x.shape))
            return super().__call__(x, t)

    monkeypatch.setattr(M, "GraphedTrainStep", CountingStep)
    eager, graphed = model.step, model._graphed_train_step

    def step(batch, metrics):
        loss = eager(batch, metrics)
        losses.append(loss)
        return loss

    def graphed_step(batch, metrics, optimizer):
        loss = graphed(batch, metrics, optimizer)
        if loss is not None:
            losses.append(loss)
        return loss

    model.step, model._graphed_train_step = step, graphed_step
    epoch_losses, inner = [], model.epoch
    model.epoch = lambda *a, **k: epoch_losses.append(inner(*a, **k)) or epoch_losses[-1]
    model.fit(ds, batch_size=16, num_epochs=1, learning_rate=1e-4, use_scheduler=True)
    assert len(losses) == 21 and len(replays) == 20 and set(replays) == {(16, 1, 5, 4096)}     # 1 eager warm-up + 20 replays
    got_losses = [float(l) for l in losses]

    # The oracle: the reference's loop on the CPU in fp32 (the reference's arithmetic), run here (21 steps, ~1 s each on the
    # box's 16 cores).  Two correct fp32 implementations drift apart through 21 Adam steps -- an element whose gradient is
    # rounding noise moves by +-lr per step in either -- so "equal" needs a yardstick: tests/golden/config2_fit_yardstick.npz
    # (make_config2_yardstick.py) holds the same loop run in fp64 -- its loss trajectory and running statistics -- and, per
    # tensor, how far torch's own fp32 run ends up from it.  The HIP fp32 path must be as close to the fp64 losses /
    # statistics as torch's fp32 is, and as close to torch's fp32 parameters as two independent fp32 runs can be expected
    # to be (each e away from fp64 => up to 2e from each other).
    import os
    import numpy as np
    from conftest import GOLDEN_DIR
    y = np.load(os.path.join(GOLDEN_DIR, "config2_fit_yardstick.npz"), allow_pickle=False)
    assert list(y["config"]) == [3, 11, 16, 348, 21]
    images = list(ds)
    sd = OrderedDict((k, v.clone()) for k, v in sd0.items())
    state = O.AdamState(O.trainable_keys(sd), sd)
    loss32 = []
    for s in range(len(images) // 16):                             # 21 steps; profiles 336..347 are never used
        batch = images[16 * s:16 * s + 16]
        x = torch.stack([torch.from_numpy(im.data.astype("float32")).permute(2, 0, 1) for im in batch])
        t = torch.stack([torch.from_numpy(im.annotation.image).permute(2, 0, 1) for im in batch])
        loss, _, _ = O.train_step(sd, x, t, state, O.exponential_lr(1e-4, 0), 5e-4)
        loss32.append(float(loss))
    loss64 = [float(v) for v in y["loss64"]]
    dev_hip = max(abs(a - b) for a, b in zip(got_losses, loss64))
    dev_cpu = max(max(abs(a - b) for a, b in zip(loss32, loss64)), float(np.abs(y["loss32"] - y["loss64"]).max()))
    print(f"loss trajectory vs fp64: hip {dev_hip:.2e}, torch fp32 {dev_cpu:.2e}; hip vs torch fp32 "
          f"{max(abs(a - b) for a, b in zip(got_losses, loss32)):.2e}; first / last {got_losses[0]:.6f} / {got_losses[-1]:.6f}")
    assert abs(got_losses[0] - loss32[0]) < 2e-6                    # the first step: no drift yet
    assert dev_hip < 2 * dev_cpu + 2e-5, (dev_hip, dev_cpu)
    assert abs(epoch_losses[0] - sum(loss64) / 21) < 2 * dev_cpu + 2e-5

    got = model._model.state_dict()
    for k, v in sd.items():
        if v.dtype == torch.int64:
            assert int(got[k]) == int(v) == 21, k                        # num_batches_tracked: 21 steps, no 22nd for the tail
    worst = {"stats": ("", 0.0, 0.0), "update": ("", 0.0, 0.0)}
    for k, e_cpu in zip(y["stats_keys"], y["stats_err_fp32_vs_fp64"]):
        k, v64 = str(k), torch.from_numpy(y["stats64/" + str(k)])
        e_hip = float((got[k].cpu().double() - v64).abs().max() / v64.abs().max())
        if e_hip > worst["stats"][1]:
            worst["stats"] = (k, e_hip, float(e_cpu))
        assert e_hip < 2 * float(e_cpu) + 0.02, (k, e_hip, float(e_cpu))
    for k, e_cpu in zip(y["update_keys"], y["update_err_fp32_vs_fp64"]):
        k = str(k)
        hip_v, cpu_v = got[k].cpu().double(), sd[k].double()
        assert float((hip_v - cpu_v).abs().max()) <= 2.5 * 1e-4 * 21, k          # no element further than 21 steps can go
        # the UPDATE of 21 steps (parameters after - before), hip against torch fp32, relative to the fp64 update's norm
        e_hip = float((hip_v - cpu_v).flatten().norm() / float(y["update_norm64/" + k]))
        if e_hip > worst["update"][1]:
            worst["update"] = (k, e_hip, float(e_cpu))
        assert e_hip < 2 * float(e_cpu) + 0.02, (k, e_hip, float(e_cpu))
    print("worst (tensor, hip error, torch fp32's distance from fp64):", worst)
    assert model._model.arena_stats()["allocations"] > 0
