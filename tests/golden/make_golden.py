"""Generate the golden fixtures in this directory by running the REFERENCE's own modules.

Run in the build container only (it needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What is imported from the reference (unchanged, via sys.path): ``DNAnet.models.segmentation.
unet_architecture.UNet`` and ``DNAnet.models.loss.DiceLoss``; the optimiser is ``torch.optim.Adam``
exactly as ``trainable_unet.py:155-157`` builds it.  The fixtures hold data only: inputs, the
seeded state_dict, and the reference's outputs (eval logits, train-mode logits, loss, every
gradient of the first step, parameters + BN buffers after the last Adam step).
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from DNAnet.models.loss import DiceLoss  # noqa: E402
from DNAnet.models.segmentation.unet_architecture import UNet  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def synth_profiles(rng, n, h, w):
    """Small synthetic electropherograms with the statistics SURVEY 8d describes."""
    x = np.abs(rng.normal(0, 8, size=(n, 1, h, w)))
    t = np.zeros((n, 1, h, w), dtype=np.int8)
    cols = np.arange(w)
    for i in range(n):
        for r in range(h):
            for _ in range(max(2, w // 128 * 3)):
                c = rng.uniform(4, w - 4)
                s = rng.uniform(1.5, 3.0)
                a = np.exp(rng.uniform(np.log(50), np.log(16000)))
                x[i, 0, r] += a * np.exp(-0.5 * ((cols - c) / s) ** 2)
                lo, hi = int(max(0, c - 2)), int(min(w, c + 3))
                t[i, 0, r, lo:hi] = 1
    return np.round(x).astype(np.int16).astype(np.float32), t


def randomize_bn(model, gen):
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=gen)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=gen)
            m.running_mean.data = 0.1 * torch.randn(m.running_mean.shape, generator=gen)
            m.running_var.data = 0.5 + 1.5 * torch.rand(m.running_var.shape, generator=gen)


def make_case(name, depth, kernel, filters, n, h, w, seed, steps=2, lr=1e-3, wd=5e-4, x=None, t=None,
              full=True):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    model = UNet(depth, kernel, filters, "cpu")
    randomize_bn(model, torch.Generator().manual_seed(seed + 1))
    if x is None:
        x, t = synth_profiles(rng, n, h, w)
    out = {"x": x, "t": t, "depth": depth, "kernel": np.array(kernel), "filters": filters,
           "lr": lr, "wd": wd, "steps": steps}
    for k, v in model.state_dict().items():
        out["sd0/" + k] = v.numpy().copy()
    xt, tt = torch.tensor(x), torch.tensor(t)
    model.eval()
    with torch.no_grad():
        out["logits_eval"] = model(xt).numpy().copy()
    # training steps exactly as trainable_unet.py:155-157,324-347
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    loss_fn = DiceLoss()
    model.train()
    opt.zero_grad()
    for s in range(steps):
        logits = model(xt)
        loss = loss_fn(logits, tt)
        loss.backward()
        out[f"step{s}/logits"] = logits.detach().numpy().copy()
        out[f"step{s}/loss"] = np.float32(float(loss))
        if s == 0 and full:
            for k, p in model.named_parameters():
                out["grad0/" + k] = p.grad.numpy().copy()
        opt.step()
        opt.zero_grad()
        if s == steps - 1 and full:
            for k, v in model.state_dict().items():
                out["sd_final/" + k] = v.numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB",
          "loss0", out["step0/loss"], "params", sum(p.numel() for p in model.parameters()))


def real_profiles():
    d = os.path.join(REF, "tests/resources/profiles/RD")
    xs, ts = [], []
    for stem in ("1A2_A01_01", "1A2_E01_13"):
        a = np.load(os.path.join(d, stem + ".npy"))              # (5,4096,1) int64
        m = np.load(os.path.join(d, stem + "_annotation.npy"))   # (5,4096,1) float64 / int8
        xs.append(np.transpose(a, (2, 0, 1)).astype(np.float32))
        ts.append(np.transpose(m, (2, 0, 1)).astype(np.int8))
    return np.stack(xs), np.stack(ts)


if __name__ == "__main__":
    make_case("g_d2_k35_f8_h5_w64", 2, (3, 5), 8, n=2, h=5, w=64, seed=11)
    make_case("g_d3_k35_f8_h5_w128", 3, (3, 5), 8, n=3, h=5, w=128, seed=12)
    make_case("g_d2_k35_f8_h6_w96", 2, (3, 5), 8, n=2, h=6, w=96, seed=13)
    make_case("g_d4_k13_f8_h5_w256", 4, (1, 3), 8, n=2, h=5, w=256, seed=14)
    x, t = real_profiles()
    # the two real RD profiles of the reference's test resources, full 5 x 4096, reduced width net
    make_case("g_real_d4_k35_f8", 4, (3, 5), 8, n=2, h=5, w=4096, seed=15, steps=1, x=x, t=t, full=False)
