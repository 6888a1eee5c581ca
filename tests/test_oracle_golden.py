"""The oracle (oracle/unet_oracle.py) against fixtures produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import unet_oracle as O


def _sd(g):
    return OrderedDict((k, v.clone()) for k, v in g["sd0"].items())


@pytest.mark.parametrize("name", golden_names())
def test_state_dict_layout(name):
    g = load_golden(name)
    spec = O.state_dict_spec(int(g["depth"]), tuple(int(v) for v in g["kernel"]), int(g["filters"]))
    assert [k for k, _, _ in spec] == list(g["sd0"].keys())
    for k, shape, _ in spec:
        assert tuple(g["sd0"][k].shape) == tuple(shape)


@pytest.mark.parametrize("name", golden_names())
def test_eval_logits(name):
    g = load_golden(name)
    logits = O.unet_forward(_sd(g), torch.from_numpy(g["x"]), train=False)
    ref = torch.from_numpy(g["logits_eval"])
    assert torch.allclose(logits, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("name", golden_names())
def test_train_steps(name):
    g = load_golden(name)
    sd = _sd(g)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    state = O.AdamState(O.trainable_keys(sd), sd)
    for s in range(int(g["steps"])):
        loss, logits, grads = O.train_step(sd, x, t, state, float(g["lr"]), float(g["wd"]))
        assert abs(loss - float(g[f"step{s}/loss"])) < 2e-6
        ref = torch.from_numpy(g[f"step{s}/logits"])
        assert torch.allclose(logits, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
        if s == 0 and g["grad0"]:
            for k, gr in g["grad0"].items():
                scale = float(gr.abs().max()) + 1e-12
                if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
                    # a conv bias in front of BatchNorm has a mathematically zero gradient; what the
                    # reference holds is rounding noise, so compare on the scale of the weight gradient
                    scale = float(g["grad0"][k[:-4] + "weight"].abs().max())
                assert float((grads[k] - gr).abs().max()) <= 2e-4 * scale + 1e-9, k
    if g["sd_final"]:
        for k, v in g["sd_final"].items():
            if v.dtype == torch.int64:
                assert int(sd[k]) == int(v), k
            else:
                assert torch.allclose(sd[k], v, rtol=1e-4, atol=2e-5), k


def test_default_init_distribution():
    sd = O.init_state_dict(2, (3, 5), 8, seed=3)
    w = sd["encoders.1.conv.double_conv.0.weight"]
    bound = 1 / (w.shape[1] * 15) ** 0.5
    assert float(w.abs().max()) <= bound and float(w.abs().max()) > 0.9 * bound
    up = sd["decoders.0.up.weight"]
    assert float(up.abs().max()) <= 1 / (up.shape[1] * 2) ** 0.5
    assert float(sd["encoders.0.conv.double_conv.1.running_var"].min()) == 1.0


def test_adam_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(37)
    sd = {"p": p.clone()}
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=1e-2, weight_decay=5e-4)
    st = O.AdamState(["p"], sd)
    for i in range(5):
        g = torch.randn(37)
        ref.grad = g.clone()
        opt.step()
        O.adam_step(sd, {"p": g}, st, 1e-2, 5e-4)
    assert torch.allclose(sd["p"], ref.data, rtol=1e-6, atol=1e-7)


def _shipped():
    import os
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "shipped_d4_k35_f32.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_shipped_config_vs_reference():
    """The shipped configuration (depth 4, (3,5), 32 filters; 12 123 745 parameters) against outputs of the reference's own
    UNet / DiceLoss on seeded weights (tests/golden/make_golden_shipped.py): eval logits, train-mode logits, loss, and of
    every parameter gradient its norm, sum and first 256 elements."""
    g = _shipped()
    sd = O.init_state_dict(4, (3, 5), 32, seed=int(g["seed"]), randomize_bn_stats=True)
    assert sum(v.numel() for k, v in sd.items() if k in O.trainable_keys(sd)) == int(g["n_params"]) == 12123745
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    ref = torch.from_numpy(g["logits_eval"])
    assert torch.allclose(O.unet_forward(sd, x, train=False), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    state = O.AdamState(O.trainable_keys(sd), sd)
    loss, logits, grads = O.train_step(sd, x, t, state, 1e-4, 5e-4)
    assert abs(loss - float(g["step0/loss"])) < 2e-6
    ref = torch.from_numpy(g["step0/logits"])
    assert torch.allclose(logits, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    for k, gr in grads.items():
        norm = float(g["gnorm/" + k])
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            continue                      # mathematically zero (bias in front of BatchNorm): rounding noise on both sides
        assert abs(float(gr.double().norm()) - norm) <= 2e-4 * norm, k
        head = torch.from_numpy(g["ghead/" + k])
        assert float((gr.reshape(-1)[:256] - head).abs().max()) <= 5e-4 * float(gr.abs().max()), k
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(sd[k], torch.from_numpy(g["stats1/" + k]), rtol=1e-4, atol=1e-5), k
