"""Golden fixture for the SHIPPED configuration (depth 4, kernel (3,5), 32 filters: config/models/unet.yaml) made by running
the REFERENCE's own modules.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_shipped.py

The 12.1 M parameters (48 MB) are not stored: the weights are ``oracle.unet_oracle.init_state_dict(4, (3, 5), 32, seed=21,
randomize_bn_stats=True)`` -- a seeded generator, reproducible anywhere -- loaded STRICTLY into the reference's ``UNet``
(``DNAnet.models.segmentation.unet_architecture``), so the fixture pins the 136-key layout at the shipped width too.  Stored:
the inputs (3 synthetic 5 x 512 profiles), the reference's eval logits, train-mode logits, Dice loss
(``DNAnet.models.loss.DiceLoss``), and of every parameter gradient of that step its L2 norm, its sum and its first 256
elements."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from DNAnet.models.loss import DiceLoss  # noqa: E402
from DNAnet.models.segmentation.unet_architecture import UNet  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402
from make_golden import synth_profiles  # noqa: E402

SEED, N, H, W = 21, 3, 5, 512

if __name__ == "__main__":
    torch.set_num_threads(4)
    sd = O.init_state_dict(4, (3, 5), 32, seed=SEED, randomize_bn_stats=True)
    model = UNet(4, (3, 5), 32, "cpu")
    model.load_state_dict(sd, strict=True)
    x, t = synth_profiles(np.random.default_rng(SEED), N, H, W)
    out = {"x": x, "t": t, "seed": SEED, "n_params": sum(p.numel() for p in model.parameters())}
    xt, tt = torch.tensor(x), torch.tensor(t)
    model.eval()
    with torch.no_grad():
        out["logits_eval"] = model(xt).numpy().copy()
    model.train()
    logits = model(xt)
    loss = DiceLoss()(logits, tt)
    loss.backward()
    out["step0/logits"] = logits.detach().numpy().copy()
    out["step0/loss"] = np.float32(float(loss))
    for k, p in model.named_parameters():
        g = p.grad.detach().reshape(-1)
        out["gnorm/" + k] = np.float64(float(g.double().norm()))
        out["gsum/" + k] = np.float64(float(g.double().sum()))
        out["ghead/" + k] = g[:256].numpy().copy()
    for k, v in model.state_dict().items():              # BatchNorm running statistics after the one train-mode forward
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["stats1/" + k] = v.numpy().copy()
    path = os.path.join(HERE, "shipped_d4_k35_f32.npz")
    np.savez_compressed(path, **out)
    print(path, f"{os.path.getsize(path) / 1e6:.2f} MB", "loss", out["step0/loss"], "params", out["n_params"])
