"""Known-answer fixture for the allele-caller hand-off, produced by the REFERENCE's
``NearestBasePairCaller.translate_pixels_to_alleles`` (build container only).

``DNAnet.allele_callers`` imports the ``.hid`` parser, which needs the third-party ``construct`` package
(absent here) for three struct-format constants only (``parse_raw_hid.py:215,220``); a three-attribute
in-process stand-in is placed in ``sys.modules`` for this script alone, as SURVEY 8c records.
Inputs: the reference's own test resources -- the analyst's annotation mask of profile 1A2_A01_01 fed
as the "prediction" (the known-answer set-up of ``tests/test_allele_callers.py:7-61``), the profile's
RFU array, ``tests/resources/panel.xml`` and a ``linspace(65, 475, 4096)`` scaler.
The fixture stores the inputs' identity (panel is re-read from the committed copy of the data, not
from the reference at test time) and the called (dye, marker, allele, height) tuples.
"""
import json
import os
import shutil
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
stub = types.ModuleType("construct")
for name, fmt in (("Int32sb", ">l"), ("Int16ub", ">H"), ("Int16sb", ">h")):
    setattr(stub, name, types.SimpleNamespace(fmtstr=fmt))
sys.modules["construct"] = stub
sys.path.insert(0, REF)
from DNAnet.allele_callers import NearestBasePairCaller  # noqa: E402
from DNAnet.data.data_models import Panel  # noqa: E402

res = os.path.join(REF, "tests/resources")
panel = Panel(os.path.join(res, "panel.xml"))
scaler = np.linspace(65, 475, 4096)
out = {}
for stem in ("1A2_A01_01", "1A2_E01_13"):
    rfu = np.load(os.path.join(res, "profiles/RD", stem + ".npy"))
    mask = np.load(os.path.join(res, "profiles/RD", stem + "_annotation.npy")).astype(np.float32)
    markers = NearestBasePairCaller().translate_pixels_to_alleles(scaler[np.newaxis, :], mask, rfu, panel)
    out[stem] = [[m.dye_row, m.name, sorted([a.name, int(a.height)] for a in m.alleles)] for m in markers]
    np.savez_compressed(os.path.join(HERE, f"allele_inputs_{stem}.npz"), rfu=rfu.astype(np.int16), mask=mask.astype(np.int8))
    print(stem, len(markers), "markers", sum(len(m[2]) for m in out[stem]), "alleles")
with open(os.path.join(HERE, "allele_calls.json"), "w") as f:
    json.dump(out, f, indent=1)
shutil.copy(os.path.join(res, "panel.xml"), os.path.join(HERE, "panel.xml"))   # data file the reference's tests hold
