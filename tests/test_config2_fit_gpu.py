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

    # the oracle: the reference's loop on the CPU (21 steps, ~1.5 s each on the box's 16 cores)
    sd = OrderedDict((k, v.clone()) for k, v in sd0.items())
    state = O.AdamState(O.trainable_keys(sd), sd)
    images = list(ds)
    want_losses = []
    for s in range(len(images) // 16):
        batch = images[16 * s:16 * s + 16]
        x = torch.stack([torch.from_numpy(im.data.astype("float32")).permute(2, 0, 1) for im in batch])
        t = torch.stack([torch.from_numpy(im.annotation.image).permute(2, 0, 1) for im in batch])
        loss, _, _ = O.train_step(sd, x, t, state, O.exponential_lr(1e-4, 0), 5e-4)
        want_losses.append(float(loss))
    worst = max(abs(a - b) for a, b in zip(got_losses, want_losses))
    print("loss trajectory: worst |delta|", worst, "first/last", got_losses[0], got_losses[-1], want_losses[-1])
    assert worst < 2e-5, (got_losses, want_losses)
    assert abs(epoch_losses[0] - sum(want_losses) / 21) < 2e-5

    got = model._model.state_dict()
    frac_worst = 0.0
    for k, v in sd.items():
        if v.dtype == torch.int64:
            assert int(got[k]) == int(v) == 21, k                        # num_batches_tracked: 21 steps, no 22nd for the tail
        elif "running" in k:
            assert torch.allclose(got[k].cpu(), v, rtol=1e-4, atol=1e-5), k
        else:
            # Adam divides by sqrt(v): an element whose gradient is rounding noise moves by +-lr per step in either
            # implementation -- bound the outliers by the distance 21 steps can cover, require the bulk to agree tightly
            d = (got[k].cpu() - v).abs()
            assert float(d.max()) <= 2.5 * 1e-4 * 21, (k, float(d.max()))
            frac = float((d > 2e-5 + 2e-3 * v.abs()).float().mean())
            frac_worst = max(frac_worst, frac)
            assert frac < 1e-2, (k, frac)
    print("worst share of elements off by more than 2e-5 + 2e-3 |v|:", frac_worst)
    assert model._model.arena_stats()["allocations"] > 0
