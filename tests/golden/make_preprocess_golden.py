"""Known-answer fixtures for the ingest resampling, produced by the REFERENCE's own functions
(``DNAnet/data/utils.py``: ``extract_ss_peaks``, ``get_interpolated_basepairs``, ``rescale_dye``; build container only).

The ``.hid`` files of the reference checkout are Git-LFS pointers, so the raw traces are synthetic: a size standard
with narrow peaks at the 31 scan positions the reference's own test pins for profile 1A2_A01_01
(``tests/data/test_utils.py:70-73``) -- in one variant the last peak is lowered below 180 RFU so that the second,
120-RFU pass of ``extract_ss_peaks`` runs, in another the ladder is stretched so that validation fails -- plus six
dye rows of random peaks.  Stored: the raw profile, the peak indices, the interpolated base pairs, the 4096 chosen
scans, the rescaled data (``profile[:-1][:, idx]``, ``hid_image.py:181-183``) and the scaler (``:111``).
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
from DNAnet.data.utils import extract_ss_peaks, get_interpolated_basepairs, rescale_dye  # noqa: E402

REAL_PEAKS = np.array([2796, 2813, 2901, 2917, 2961, 2989, 3038, 3095, 3144, 3387, 3645, 3702, 3881, 4120, 4348, 4578, 4806,
                       5033, 5257, 5531, 5798, 6059, 6323, 6579, 6827, 7077, 7318, 7560, 7788, 8018, 8238])


def trace(peaks, heights, length, rng):
    t = np.abs(rng.normal(0, 6, size=length))
    x = np.arange(length)
    for p, h in zip(peaks, heights):
        t += h * np.exp(-0.5 * ((x - p) / 2.2) ** 2)
    return np.round(t).astype(np.int64)


def make(name, seed, peaks, low_tail=False, length=9100):
    rng = np.random.default_rng(seed)
    heights = rng.uniform(400, 2500, size=len(peaks))
    if low_tail:
        heights[-1] = 150
    ss = trace(peaks, heights, length, rng)
    dyes = np.stack([trace(rng.uniform(2900, 8200, size=30), np.exp(rng.uniform(np.log(50), np.log(16000), size=30)), length, rng)
                     for _ in range(6)])
    profile = np.concatenate([dyes, ss[None]]).astype(np.int64)
    found = extract_ss_peaks(np.array(profile[-1]))
    bps = get_interpolated_basepairs(np.array(profile[-1]))
    out = {"profile": profile.astype(np.int32), "peaks": found, "valid": np.array(bps is not None)}
    if bps is not None:
        idx = rescale_dye(bps)
        out.update(basepairs=bps, idx=idx, data=profile[:-1][:, idx][..., np.newaxis].astype(np.int32), scaler=bps[idx])
    np.savez_compressed(os.path.join(HERE, f"preprocess_{name}.npz"), **out)
    print(name, "valid" if bps is not None else "rejected", len(found), "peaks")


make("real_ladder", 1, REAL_PEAKS)
make("low_tail", 2, REAL_PEAKS, low_tail=True)
make("shifted", 3, REAL_PEAKS[8:] + np.arange(len(REAL_PEAKS) - 8) * 3 - 400, length=9000)
make("stretched_invalid", 4, (REAL_PEAKS * 1.6).astype(int) - 3000, length=11000)
