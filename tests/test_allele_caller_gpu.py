"""Device allele hand-off (dn_call_alleles through NearestBasePairCaller.call_alleles_batch) against the host caller and
the reference's known answers (tests/golden/allele_calls.json, produced by the reference's NearestBasePairCaller):
identical markers, alleles, heights AND order."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from dnanet_amd.allele_caller import NearestBasePairCaller
from dnanet_amd.data import Allele, ArrayImage, Marker, Panel
from dnanet_amd.prediction import Prediction

pytestmark = pytest.mark.gpu


def _flat(markers):
    return [(m.dye_row, m.name, [(a.name, int(a.height)) for a in m.alleles]) for m in markers]


def _device_call(images, probs_np, gpu_device):
    """probs_np: (N,H,W,1) fp32.  Returns the called alleles per image from the device path."""
    probs = torch.from_numpy(np.ascontiguousarray(np.moveaxis(probs_np, -1, 1))).to(gpu_device)        # (N,1,H,W)
    rfu = torch.from_numpy(np.stack([np.moveaxis(np.asarray(im.data), -1, 0) for im in images]).astype(np.float32)).to(gpu_device)
    preds = [Prediction(image=probs_np[i]) for i in range(len(images))]
    NearestBasePairCaller().call_alleles_batch(images, preds, probs, rfu)
    return [p.meta["called_alleles"] for p in preds]


@pytest.fixture(scope="module")
def panel():
    return Panel(os.path.join(GOLDEN_DIR, "panel.xml"))


def test_reference_known_answers(panel, gpu_device):
    expected = json.load(open(os.path.join(GOLDEN_DIR, "allele_calls.json")))
    scaler = np.linspace(65, 475, 4096)
    images, probs = [], []
    for stem in ("1A2_A01_01", "1A2_E01_13"):
        z = np.load(os.path.join(GOLDEN_DIR, f"allele_inputs_{stem}.npz"))
        images.append(ArrayImage(data=z["rfu"], _scaler=scaler, _panel=panel, meta={}))
        probs.append(z["mask"].astype(np.float32))
    called = _device_call(images, np.stack(probs), gpu_device)
    for stem, markers, im, pr in zip(("1A2_A01_01", "1A2_E01_13"), called, images, probs):
        got = [[m.dye_row, m.name, sorted([a.name, int(a.height)] for a in m.alleles)] for m in markers]
        assert got == expected[stem]
        host = NearestBasePairCaller().translate_pixels_to_alleles(scaler[None], pr, im.data, panel)
        assert _flat(markers) == _flat(host)              # and the order of this build's own host caller (the reference keeps a set: no order of its own)
    d18 = [m for m in called[0] if m.name == "D18S51"]
    assert {a.name for a in d18[0].alleles} == {"13.1", "15.1", "16.1"}   # reference tests/test_allele_callers.py:56-61


def test_random_maps_edges_and_per_profile_scalers(panel, gpu_device):
    rng = np.random.default_rng(11)
    n, h, w = 6, 5, 4096
    probs = np.zeros((n, h, w, 1), np.float32)
    rfu = rng.integers(-50, 20000, size=(n, h, w, 1)).astype(np.int16)
    for i in range(n):
        for d in range(h):
            for _ in range(int(rng.integers(0, 60))):
                a = int(rng.integers(0, w)); b = min(w, a + int(rng.integers(1, 14)))
                probs[i, d, a:b, 0] = rng.uniform(0.5, 1.0)
    probs[0, 0, :3, 0] = 0.5            # run touching column 0, exactly at the threshold
    probs[0, 1, -2:, 0] = 0.9           # run touching the last column
    probs[1, 2, :, 0] = 0.7             # a whole row is one run
    probs[2, 3, :, 0] = 0.0             # an empty dye row
    probs[3, 4, 100:104, 0] = np.float32(0.49999997)   # just below the threshold
    images = [ArrayImage(data=rfu[i], _scaler=np.linspace(60 + i, 470 + 2 * i, w), _panel=panel,
                         meta={"called_alleles_manual": []} if i == 4 else {}) for i in range(n)]
    called = _device_call(images, probs, gpu_device)
    host = NearestBasePairCaller()
    for i in range(n):
        expect = host.call_alleles(images[i], Prediction(image=probs[i])).meta["called_alleles"]
        assert _flat(called[i]) == _flat(expect)
    assert all(m.name not in ("AMEL", "DYS391", "DYS576", "DYS570") for m in called[4])


def test_tie_and_mixed_panels(gpu_device):
    p1 = Panel(panel_contents=[Marker(0, "M", [Allele("b", 102.0), Allele("a", 100.0)])])
    p2 = Panel(panel_contents=[Marker(0, "K", [Allele("x", 50.0)]), Marker(1, "L", [Allele("y", 60.0), Allele("z", 61.0)])])
    probs = np.zeros((2, 2, 8, 1), np.float32)
    probs[0, 0, 3, 0] = 1; probs[1, 1, 2:5, 0] = 1; probs[1, 0, 7, 0] = 1
    rfu = np.arange(2 * 2 * 8, dtype=np.int16).reshape(2, 2, 8, 1)
    images = [ArrayImage(data=rfu[0], _scaler=np.full(8, 101.0), _panel=p1, meta={}),
              ArrayImage(data=rfu[1], _scaler=np.linspace(60.0, 61.0, 8), _panel=p2, meta={})]
    called = _device_call(images, probs, gpu_device)
    assert _flat(called[0]) == [(0, "M", [("b", 3)])]          # equidistant: the allele listed first in the panel
    host = NearestBasePairCaller()
    for i in range(2):
        expect = host.translate_pixels_to_alleles(images[i]._scaler[None], probs[i], images[i].data, images[i]._panel)
        assert _flat(called[i]) == _flat(expect)
