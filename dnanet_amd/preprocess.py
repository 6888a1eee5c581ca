"""Raw electropherogram -> network input: size-standard peaks, base-pair interpolation, resampling to 4096 scans.

Restates the array side of the reference's ingest (``DNAnet/data/utils.py:7-279``, driven by
``HIDImage._read`` / ``_rescale_profile``, ``DNAnet/data/data_models/hid_image.py:88-183``); parsing the ``.hid``
container itself stays with the reference.  The host functions are vectorised numpy with the reference's names and
results (pinned by ``tests/golden/preprocess_*.npz``, produced by the reference's own functions, and by the known
answers of ``tests/data/test_utils.py``); ``rescale_profiles_device`` does the per-profile search + gather for a
whole batch in one kernel (``dn_rescale_profiles``) and leaves the result in the ``(N, 1, H, 4096)`` fp32 layout
the U-Net reads.

Difference kept on purpose: the reference shifts arrays with ``scipy.ndimage.shift`` (a cubic-spline resampler);
for the integer RFU arrays it is applied to, that equals the exact shift used here.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

SIZE_STANDARD_BPS = np.array([65, 80, 100, 120, 140, 160, 180, 200, 225, 250, 275, 300, 325, 350, 375, 400, 425, 450, 475])
BASE_PAIR_START, BASE_PAIR_END = 65, 475
RESCALE_SIZE = 4096


def _shift(array: np.ndarray, by: int) -> np.ndarray:
    """array moved ``by`` places to the right (left if negative), zeros coming in."""
    out = np.zeros_like(array)
    if by > 0:
        out[by:] = array[:-by]
    elif by < 0:
        out[:by] = array[-by:]
    else:
        out[...] = array
    return out


def find_peaks_above_threshold(array: np.ndarray, threshold: int) -> np.ndarray:
    """Indices of local maxima >= threshold, including the first and last sample of a flat top (``utils.py:117-127``)."""
    array = np.asarray(array)
    prev, nxt = _shift(array, 1), _shift(array, -1)
    return np.where((((array >= prev) & (array > nxt)) | ((array > prev) & (array >= nxt))) & (array >= threshold))[0]


def extract_ss_peaks(array: np.ndarray) -> np.ndarray:
    """Size-standard peak indices (``utils.py:196-214``): peaks >= 180 RFU, the tail past scan 8200 again at >= 120 when
    nothing was found there, and of two peaks within 15 scans the first is dropped."""
    array = np.asarray(array)
    peaks = find_peaks_above_threshold(array, 180)
    split = 8200
    if len(peaks) > 0 and peaks[-1] <= split:
        peaks = np.union1d(peaks, find_peaks_above_threshold(array[split:], 120) + split)
    return np.delete(peaks, np.where(np.diff(peaks) <= 15)[0])


def validate_ss_peaks(peak_idxs: np.ndarray) -> bool:
    """19 peaks, 7..13 scans per base pair between neighbours (``utils.py:66-88``)."""
    if len(peak_idxs) != 19:
        return False
    relative = np.abs(np.diff(peak_idxs)) / np.abs(np.diff(SIZE_STANDARD_BPS))
    return bool(np.all((relative <= 13) & (relative >= 7)))


class basepair_interpolator:
    """``scipy.interpolate.interp1d(indices, values, bounds_error=False, fill_value=0 | 'extrapolate')`` restated with
    scipy's own arithmetic (slope * (x - x_lo) + y_lo on the bracketing pair), ``utils.py:217-236``."""

    def __init__(self, indices, original_x_values, extrapolate: bool = False):
        self.x = np.asarray(indices, dtype=np.float64)
        self.y = np.asarray(original_x_values, dtype=np.float64)
        self.extrapolate = extrapolate

    def __call__(self, x_new) -> np.ndarray:
        x_new = np.atleast_1d(np.asarray(x_new, dtype=np.float64))
        hi = np.clip(np.searchsorted(self.x, x_new), 1, len(self.x) - 1).astype(int)
        lo = hi - 1
        slope = (self.y[hi] - self.y[lo]) / (self.x[hi] - self.x[lo])
        out = slope * (x_new - self.x[lo]) + self.y[lo]
        if not self.extrapolate:
            out = np.where((x_new < self.x[0]) | (x_new > self.x[-1]), 0.0, out)
        return out


def get_interpolated_basepairs(size_standard: np.ndarray) -> Optional[np.ndarray]:
    """Base pair of every raw scan (0 outside the ladder), or None when the ladder fails validation
    (``utils.py:91-114``)."""
    peaks = extract_ss_peaks(size_standard)[-20:-1]
    if not validate_ss_peaks(peaks):
        return None
    return basepair_interpolator(peaks, SIZE_STANDARD_BPS)(np.arange(len(size_standard)))


def rescale_dye(basepairs: np.ndarray) -> np.ndarray:
    """For each of the 4096 target base pairs ``linspace(65, 475, 4096)`` the raw scan whose base pair is nearest (the
    lower one on a tie), ``utils.py:239-279``."""
    target = np.linspace(BASE_PAIR_START, BASE_PAIR_END, RESCALE_SIZE)
    order = np.argsort(basepairs)
    ordered = basepairs[order]
    ins = np.clip(np.searchsorted(ordered, target, side="left"), 1, len(ordered) - 1)
    left, right = ins - 1, ins
    dl, dr = np.abs(ordered[left] - target), np.abs(ordered[right] - target)
    return np.where(dl <= dr, order[left], order[right])


def rescale_profile(profile: np.ndarray, include_size_standard: bool = False):
    """(data ``(H, 4096, 1)``, scaler ``(4096,)``) of one raw profile ``(dyes + 1, L)`` whose last row is the size
    standard, or None when the ladder fails validation (``hid_image.py:96-111,168-183``)."""
    profile = np.asarray(profile)
    bps = get_interpolated_basepairs(np.array(profile[-1]))
    if bps is None:
        return None
    idx = rescale_dye(bps)
    rows = profile if include_size_standard else profile[:-1]
    return rows[:, idx][..., np.newaxis], bps[idx]


def rescale_profiles_device(profiles: Sequence[np.ndarray], include_size_standard: bool = False, device="cuda"):
    """Batched ingest: the ladder is located on the host (19 peaks per profile), the 4096 nearest-base-pair searches
    and the gather run on the GPU.  Returns ``(x, scalers, valid)``: ``x`` ``(n_valid, 1, H, 4096)`` fp32 on the
    device (the tensor ``DNANet_UNet.get_inputs`` would build), ``scalers`` ``(n_valid, 4096)`` float64 on the host,
    ``valid`` the indices of the profiles whose size standard passed validation (the reference skips the others)."""
    import torch
    from . import hip

    bps, valid = [], []
    for i, profile in enumerate(profiles):
        b = get_interpolated_basepairs(np.asarray(profile)[-1])
        if b is not None:
            bps.append(b)
            valid.append(i)
    if not valid:
        return None, np.zeros((0, RESCALE_SIZE)), []
    length = max(np.asarray(profiles[i]).shape[1] for i in valid)
    rows_all = np.asarray(profiles[valid[0]]).shape[0]
    rows = rows_all if include_size_standard else rows_all - 1
    n = len(valid)
    raw = np.zeros((n, rows, length), dtype=np.float32)
    bp = np.zeros((n, length), dtype=np.float64)
    span = np.zeros((n, 2), dtype=np.int32)
    for k, i in enumerate(valid):
        p = np.asarray(profiles[i])
        if p.shape[0] != rows_all:
            raise ValueError("profiles of one batch must have the same number of dyes")
        raw[k, :, :p.shape[1]] = p[:rows]
        bp[k, :p.shape[1]] = bps[k]
        nz = np.flatnonzero(bps[k])                      # the ladder span: strictly increasing base pairs
        span[k] = (nz[0], nz[-1])
    dev = torch.device(device)
    raw_d, bp_d, span_d = (torch.from_numpy(a).to(dev) for a in (raw, bp, span))
    target = torch.from_numpy(np.linspace(BASE_PAIR_START, BASE_PAIR_END, RESCALE_SIZE)).to(dev)
    x = torch.empty((n, 1, rows, RESCALE_SIZE), dtype=torch.float32, device=dev)
    scalers = torch.empty((n, RESCALE_SIZE), dtype=torch.float64, device=dev)
    hip.check(hip.lib.dn_rescale_profiles(hip.ptr(raw_d), hip.ptr(bp_d), hip.ptr(span_d), hip.ptr(target), n, rows, length,
                                          RESCALE_SIZE, hip.ptr(x), hip.ptr(scalers), None, hip.stream_ptr()))
    return x, scalers.cpu().numpy(), valid
