"""Minimal host-side data types on either side of the hot path.

The model is duck-typed on what the reference's ``HIDImage`` exposes to it
(``DNAnet/data/data_models/hid_image.py:65-86,147-165``): ``.data`` (H,W,1) int16, ``.annotation.image``
(H,W,1) {0,1}, ``.path``, ``.meta``, ``._scaler`` (W float64 base pairs), ``._panel``.  These light
classes carry exactly that for tests, benches and callers that have arrays but no ``.hid`` files;
the reference's own ``HIDImage`` / ``HIDDataset`` objects work unchanged.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from itertools import islice
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Allele:
    """``DNAnet/data/data_models/dna_models.py:18-40``"""
    name: str
    base_pair: float = None
    left_bin: float = None
    right_bin: float = None
    height: float = None


@dataclass
class Marker:
    """``DNAnet/data/data_models/dna_models.py:43-54``"""
    dye_row: int
    name: str
    alleles: Sequence[Allele]


class Panel:
    """Marker/allele base-pair table of a kit (``dna_models.py:57-177``): only what the allele caller
    needs -- per dye, the base pair -> (marker, allele) map, in panel order."""

    DYE_MAPPING = {1: 0, 2: 1, 3: 2, 4: 3, 6: 4}     # dna_models.py:13

    def __init__(self, panel_path=None, panel_contents: Optional[Sequence[Marker]] = None):
        if panel_path:
            self._panel = self._parse(panel_path)
        elif panel_contents:
            self._panel = list(panel_contents)
        else:
            raise ValueError("Cannot instantiate Panel object, since panel path nor panel contents are provided.")
        self._mapping = None

    @property
    def dye_bp_to_allele_mapping(self) -> Dict[int, Dict[float, Tuple[str, str]]]:
        if self._mapping is None:
            m: Dict[int, Dict[float, Tuple[str, str]]] = {}
            for marker in self._panel:
                for allele in marker.alleles:
                    m.setdefault(marker.dye_row, {})[allele.base_pair] = (marker.name, allele.name)
            self._mapping = m
        return self._mapping

    def get_allele_name_by_dye_and_bp(self, dye_row: int, base_pair: float) -> Tuple[str, str]:
        try:
            return self.dye_bp_to_allele_mapping[dye_row][base_pair]
        except KeyError:
            raise ValueError(f"Unknown dye row {dye_row} or base_pair {base_pair} found.")

    @classmethod
    def _parse(cls, path) -> List[Marker]:
        panel = []
        for locus in ET.parse(path).getroot().iter("Locus"):
            panel.append(Marker(
                cls.DYE_MAPPING.get(int(locus.find("DyeIndex").text)),
                locus.find("MarkerTitle").text,
                [Allele(a.attrib["Label"], float(a.attrib["Size"]), float(a.attrib["Left_Binning"]),
                        float(a.attrib["Right_Binning"])) for a in locus.findall("Allele")]))
        return panel


class Annotation:
    def __init__(self, image: np.ndarray = None, labels=None, meta: Dict[str, Any] = None):
        self.image, self.labels, self.meta = image, labels, meta or {}


@dataclass
class ArrayImage:
    """An electropherogram given as arrays."""
    data: np.ndarray                        # (H, W, 1) int16 RFU
    annotation: Optional[Annotation] = None  # .image (H, W, 1) int8
    path: Optional[str] = None
    meta: Dict[str, Any] = field(default_factory=dict)
    _scaler: Optional[np.ndarray] = None    # (W,) base pair of every scan column
    _panel: Optional[Panel] = None


def chunks(iterable: Iterable, chunk_size: int, skip_remainder: bool = False) -> Iterator[List]:
    """Consecutive batches of ``chunk_size`` (the last one may be shorter unless ``skip_remainder``);
    same contract as ``DNAnet/utils.py:136-171``."""
    if chunk_size <= 0:
        raise ValueError("chunk_size must be positive")
    it = iter(iterable)
    for first in it:
        batch = [first, *islice(it, chunk_size - 1)]
        if skip_remainder and len(batch) < chunk_size:
            return
        yield batch
