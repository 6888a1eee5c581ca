"""Synthetic electropherograms of the reference ``HIDImage`` shape: per dye row a
noise floor |N(0,8)| plus 15-40 Gaussian peaks (centre U(0,W), sigma U(3,6) px, height log-uniform
50..16000 RFU), rounded to int16 like ``HIDImage.data``; targets mark +-2 px around every peak."""
from __future__ import annotations

import numpy as np


def make_profiles(n: int, h: int = 5, w: int = 4096, seed: int = 1234, peaks=(15, 40)):
    rng = np.random.default_rng(seed)
    x = np.abs(rng.normal(0.0, 8.0, size=(n, h, w))).astype(np.float32)
    t = np.zeros((n, h, w), dtype=np.int8)
    cols = np.arange(w, dtype=np.float32)
    for i in range(n):
        for r in range(h):
            k = int(rng.integers(peaks[0], peaks[1] + 1))
            centres = rng.uniform(0, w, size=k)
            sigmas = rng.uniform(3.0, 6.0, size=k)
            heights = np.exp(rng.uniform(np.log(50.0), np.log(16000.0), size=k))
            for c, s, a in zip(centres, sigmas, heights):
                lo, hi = max(0, int(c - 6 * s)), min(w, int(c + 6 * s) + 1)
                x[i, r, lo:hi] += a * np.exp(-0.5 * ((cols[lo:hi] - c) / s) ** 2)
                t[i, r, max(0, int(round(c)) - 2):min(w, int(round(c)) + 3)] = 1
    x = np.clip(np.round(x), 0, 32767).astype(np.int16)
    return x[:, :, :, None].transpose(0, 1, 2, 3), t[:, :, :, None]      # (n,h,w,1) like HIDImage.data


def make_batch(n: int, h: int = 5, w: int = 4096, seed: int = 1234, unique: int = 16):
    """(N,1,H,W) fp32 inputs and (N,1,H,W) int8 targets; ``unique`` distinct profiles are tiled to N
    (generation is host-side numpy and not part of any timed region)."""
    u = min(n, unique)
    x, t = make_profiles(u, h, w, seed)
    reps = (n + u - 1) // u
    x = np.tile(x[:, :, :, 0][:, None].astype(np.float32), (reps, 1, 1, 1))[:n]
    t = np.tile(t[:, :, :, 0][:, None], (reps, 1, 1, 1))[:n]
    return x, t
