"""Ingest resampling: the numpy restatement against fixtures produced by the reference's own functions
(tests/golden/make_preprocess_golden.py) and the known answers of the reference's tests/data/test_utils.py; the
batched GPU kernel against the same fixtures."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from dnanet_amd import preprocess as P

NAMES = ["real_ladder", "low_tail", "shifted", "stretched_invalid"]


def _load(name):
    return np.load(os.path.join(GOLDEN_DIR, f"preprocess_{name}.npz"))


@pytest.mark.parametrize("array, result", [([1000, 1001, 1010, 1003, 1002], [2]), ([100, 101, 110, 103, 102], []),
                                           ([1000, 1003, 1003, 999], [1, 2]), ([1000, 1001, 800, 800, 799], [1, 3])])
def test_find_peaks_above_threshold_known_answers(array, result):       # reference tests/data/test_utils.py:16-26
    assert np.array_equal(P.find_peaks_above_threshold(np.array(array), 200), np.array(result))


def test_basepair_interpolator_known_answers():                         # reference tests/data/test_utils.py:82-110
    assert np.array_equal(P.basepair_interpolator([0, 3, 6], [1, 7, 13])(np.arange(6)), [1., 3., 5., 7., 9., 11.])
    ex = P.basepair_interpolator([2, 3], [12, 13], extrapolate=True)
    assert np.array_equal(ex(np.arange(3)), [10., 11., 12.]) and ex(4) == [14.] and ex(5) == [15.]
    assert np.array_equal(P.basepair_interpolator([97.82, 102.13], [97.64, 101.95])(101.22), np.array([101.04]))
    assert np.array_equal(P.basepair_interpolator([97.82, 102.13], [97.64, 101.95], extrapolate=True)(93.47), np.array([93.29]))


@pytest.mark.parametrize("name", NAMES)
def test_restatement_equals_reference_fixture(name):
    g = _load(name)
    profile = g["profile"].astype(np.int64)
    assert np.array_equal(P.extract_ss_peaks(profile[-1]), g["peaks"])
    bps = P.get_interpolated_basepairs(profile[-1])
    assert (bps is not None) == bool(g["valid"])
    if bps is None:
        assert P.rescale_profile(profile) is None
        return
    assert np.array_equal(bps, g["basepairs"])                          # float64, bit for bit
    assert np.array_equal(P.rescale_dye(bps), g["idx"])
    data, scaler = P.rescale_profile(profile)
    assert data.shape == (6, 4096, 1) and np.array_equal(data, g["data"]) and np.array_equal(scaler, g["scaler"])


@pytest.mark.gpu
def test_device_batch_equals_reference_fixtures(gpu_device):
    """dn_rescale_profiles on the whole batch (ragged lengths, two rejected ladders): network-layout fp32 tensor, scalers
    and chosen scans identical to the reference's."""
    import torch
    from dnanet_amd import hip
    fixtures = [_load(n) for n in NAMES]
    x, scalers, valid = P.rescale_profiles_device([g["profile"] for g in fixtures], device=gpu_device)
    assert valid == [i for i, g in enumerate(fixtures) if bool(g["valid"])] and len(valid) == 3
    assert x.shape == (3, 1, 6, 4096) and x.dtype == torch.float32 and x.is_cuda
    for k, i in enumerate(valid):
        g = fixtures[i]
        assert np.array_equal(x[k, 0].cpu().numpy(), g["data"][..., 0].astype(np.float32))
        assert np.array_equal(scalers[k], g["scaler"])
    # with the size standard kept as a seventh row, and the chosen scans reported
    x7, _, _ = P.rescale_profiles_device([fixtures[0]["profile"]], include_size_standard=True, device=gpu_device)
    assert x7.shape == (1, 1, 7, 4096)
    assert np.array_equal(x7[0, 0, 6].cpu().numpy(), fixtures[0]["profile"][-1][fixtures[0]["idx"]].astype(np.float32))
    assert hip.lib.dn_rescale_profiles(None, None, None, None, 1, 1, 2, 1, None, None, None, None) != 0
