"""The SHIPPED configuration (depth 4, (3,5), 32 filters) on the HIP kernels against outputs of the reference's own UNet /
DiceLoss (tests/golden/shipped_d4_k35_f32.npz, made by tests/golden/make_golden_shipped.py; weights = a seeded generator, so
the 48 MB of parameters are rebuilt here instead of stored)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shipped():
    from oracle import unet_oracle as O
    z = np.load(os.path.join(GOLDEN_DIR, "shipped_d4_k35_f32.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["sd"] = O.init_state_dict(4, (3, 5), 32, seed=int(g["seed"]), randomize_bn_stats=True)
    return g


# eval logits, relative to max |logit|: fp32 = accumulation order only; bf16x3 = 16-bit mantissas; bf16 = 8-bit
EVAL_TOL = {"fp32": 2e-5, "bf16x3": 5e-5, "bf16x3@2": 5e-3, "bf16": 2e-2}


@pytest.mark.parametrize("dtype", list(EVAL_TOL))
def test_eval_logits_and_masks(gpu_device, shipped, dtype):
    from dnanet_amd.unet import HipUNet
    m = HipUNet(4, (3, 5), 32, gpu_device, dtype=dtype)
    m.load_state_dict(shipped["sd"])
    m.eval()
    logits, probs = m.infer(torch.from_numpy(shipped["x"]).to(gpu_device))
    ref = torch.from_numpy(shipped["logits_eval"])
    err = float((logits.cpu() - ref).abs().max() / ref.abs().max())
    assert err <= EVAL_TOL[dtype], err
    flips = int(((probs.cpu() >= 0.5) != (torch.sigmoid(ref) >= 0.5)).sum())
    if dtype in ("fp32", "bf16x3"):
        assert flips == 0
    else:
        # every flipped pixel's reference logit is inside the mode's logit error
        bad = ((probs.cpu() >= 0.5) != (torch.sigmoid(ref) >= 0.5))
        assert flips == 0 or float(ref[bad].abs().max()) <= EVAL_TOL[dtype] * float(ref.abs().max())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_train_step_loss_and_gradients(gpu_device, shipped, dtype):
    from dnanet_amd.unet import HipUNet
    m = HipUNet(4, (3, 5), 32, gpu_device, dtype=dtype)
    m.load_state_dict(shipped["sd"])
    m.train()
    x, t = torch.from_numpy(shipped["x"]).to(gpu_device), torch.from_numpy(shipped["t"]).to(gpu_device)
    m.flat_grads.zero_()
    logits = m(x)
    sums = m.dice_forward(logits, t)
    m.backward()
    assert abs(float(sums[4]) - float(shipped["step0/loss"])) < (2e-5 if dtype == "fp32" else 3e-4)
    ref = torch.from_numpy(shipped["step0/logits"])
    err = float((logits.cpu() - ref).abs().max() / ref.abs().max())
    assert err < (1e-4 if dtype == "fp32" else 3e-2), err
    worst_norm, worst_head, dots = ("", 0.0), ("", 0.0), []
    for k, _, kind in m.spec:
        if "gnorm/" + k not in shipped or k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            continue                        # (a bias in front of BatchNorm: zero gradient, rounding noise in the reference)
        g = m.grad(k).cpu()
        norm = float(shipped["gnorm/" + k])
        e = abs(float(g.double().norm()) - norm) / norm
        if e > worst_norm[1]:
            worst_norm = (k, e)
        head = torch.from_numpy(shipped["ghead/" + k])
        eh = float((g.reshape(-1)[:head.numel()] - head).norm() / (head.norm() + 1e-20))
        if eh > worst_head[1]:
            worst_head = (k, eh)
    print(dtype, "worst gradient-norm error", worst_norm, "worst relative L2 over the stored leading elements", worst_head)
    if dtype == "fp32":
        assert worst_norm[1] < 2e-3 and worst_head[1] < 8e-3, (worst_norm, worst_head)
        # BatchNorm running statistics after the one train-mode forward
        for k, _, kind in m.spec:
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert torch.allclose(m.tensor(k).cpu(), torch.from_numpy(shipped["stats1/" + k]), rtol=1e-4, atol=1e-5), k
    else:
        assert worst_norm[1] < 0.25 and worst_head[1] < 0.6, (worst_norm, worst_head)
