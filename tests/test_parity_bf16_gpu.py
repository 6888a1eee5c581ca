"""What the precisions cost where the caller looks: thresholded masks and called alleles at the SHIPPED configuration
(depth 4, kernel (3,5), 32 filters, 5 x 4096), fp32 and bf16 kernels against the fp32 CPU oracle, on the two real RD
profiles of the reference's test resources plus 64 synthetic ones.  The pretrained checkpoint is a Git-LFS pointer in
this checkout, so the weights are a seeded net after a short training run (sparse peak masks, like a real model's) --
and, as the worst case, the seeded random-init net whose logits hover around the threshold.

north_star: "logits within a stated fp tolerance, binary segmentation masks and AlleleCaller called alleles identical".
fp32 mode must meet that sentence; bf16 mode is measured and bounded (the counts are printed and also reported by
bench.py's ``parity`` block)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu

N_SYNTH = 64


@pytest.fixture(scope="module")
def shipped_cases(gpu_device):
    """inputs (66,1,5,4096) and, per weight set, the oracle's fp32 probabilities."""
    from dnanet_amd.parity import briefly_trained_state
    from dnanet_amd.synthetic import make_batch
    from oracle import unet_oracle as O
    torch.set_num_threads(16)
    real = load_golden("g_real_d4_k35_f8")["x"].astype(np.float32)            # the two real RD profiles
    x = torch.from_numpy(np.concatenate([real, make_batch(N_SYNTH, 5, 4096, seed=4321, unique=N_SYNTH)[0]]))
    states = {"briefly_trained": briefly_trained_state(4, (3, 5), 32, gpu_device),
              "random_init": O.init_state_dict(4, (3, 5), 32, seed=0, randomize_bn_stats=True)}
    ref = {}
    with torch.no_grad():
        for name, sd in states.items():
            ref[name] = torch.cat([O.unet_forward(sd, x[i:i + 8], train=False) for i in range(0, x.shape[0], 8)])
    return x, states, ref


def _hip_logits(sd, x, dtype, dev):
    from dnanet_amd.unet import HipUNet
    m = HipUNet(4, (3, 5), 32, dev, dtype=dtype)
    m.load_state_dict(sd)
    m.eval()
    return m.infer(x.to(dev))


@pytest.mark.parametrize("weights", ["briefly_trained", "random_init"])
def test_fp32_masks_and_alleles_identical_to_oracle(gpu_device, shipped_cases, weights):
    from dnanet_amd.parity import disagreement
    x, states, ref = shipped_cases
    logits, probs = _hip_logits(states[weights], x, "fp32", gpu_device)
    r = ref[weights]
    err = float((logits.cpu() - r).abs().max() / r.abs().max())
    assert err < 2e-5, err                                             # stated fp32 tolerance on logits
    d = disagreement(probs, torch.sigmoid(r), x.to(gpu_device))
    print(f"\nfp32 kernels vs fp32 oracle, {weights}: {json.dumps(d)}")
    # a pixel may only differ where the reference logit itself is within the logit tolerance of the threshold
    undecided = int((r.abs() <= 2e-5 * float(r.abs().max())).sum())
    assert d["mask_pixels_differing"] <= undecided, (d, undecided)
    if undecided == 0:
        assert d["mask_pixels_differing"] == 0 and d["alleles_differing"] == 0
    if weights == "briefly_trained":
        assert d["mask_pixels_differing"] == 0 and d["alleles_differing"] == 0, d
        assert d["alleles_called"] > 0 and d["mask_positive_pixels"] > 0       # a net that predicts peaks


@pytest.mark.parametrize("weights", ["briefly_trained", "random_init"])
def test_bf16_mask_and_allele_disagreement_is_bounded(gpu_device, shipped_cases, weights):
    from dnanet_amd.parity import disagreement
    x, states, ref = shipped_cases
    logits, probs = _hip_logits(states[weights], x, "bf16", gpu_device)
    r = ref[weights]
    err = float((logits.cpu() - r).abs().max() / r.abs().max())
    d = disagreement(probs, torch.sigmoid(r), x.to(gpu_device))
    d["logits_rel_err"] = err
    print(f"\nbf16 kernels vs fp32 oracle, {weights}: {json.dumps(d)}")
    out = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"parity_bf16_{weights}.json"), "w") as f:
            json.dump(d, f)
    assert err < 4e-2, err                                             # stated bf16 tolerance on logits
    # bf16 is not claimed identical: flips are confined to pixels whose reference logit is within the bf16 logit
    # tolerance of the threshold, and their number is bounded relative to the positives
    near = (r.abs() <= 4e-2 * float(r.abs().max()))
    flipped = (probs.cpu() >= 0.5) != (torch.sigmoid(r) >= 0.5)
    assert not bool((flipped & ~near).any())
    assert d["mask_pixels_differing"] <= BF16_MAX_FLIPPED_SHARE[weights] * max(1, d["mask_positive_pixels"]), d
    if weights == "briefly_trained":
        assert d["alleles_differing"] <= BF16_MAX_ALLELE_SHARE * max(1, d["alleles_called"]), d


# measured on MI355X (round 2, gpurun_out/parity_bf16_*.json) and rounded up; see DESIGN.md section 4
BF16_MAX_FLIPPED_SHARE = {"briefly_trained": 0.05, "random_init": 0.05}
BF16_MAX_ALLELE_SHARE = 0.05
