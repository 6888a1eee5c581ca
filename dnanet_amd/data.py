"""Minimal host-side data types on either side of the hot path.

The model is duck-typed on what the reference's ``HIDImage`` exposes to it
(``DNAnet/data/data_models/hid_image.py:65-86,147-165``): ``.data`` (H,W,1) int16, ``.annotation.image``
(H,W,1) {0,1}, ``.path``, ``.meta``, ``._scaler`` (W float64 base pairs), ``._panel``.  These light
classes carry exactly that for tests, benches and callers that have arrays but no ``.hid`` files;
the reference's own ``HIDImage`` / ``HIDDataset`` objects work unchanged.
"""
from __future__ import annotations

import glob
import os
import random
import re
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


def shared_seed(seed):
    """The seed every rank of a torch.distributed job must use for a split / shuffle.  ``random.seed(None)`` seeds from
    OS entropy per process, which would give every rank its own training / validation split (and, in ``fit``, ranks that
    disagree about early stopping and hang in a collective): inside an initialised process group of more than one rank
    a missing seed is drawn on rank 0 and broadcast.  Outside a group, or with an explicit seed, it is returned as is
    (the reference's behaviour, ``DNAnet/data/data_models/base.py:69-111``)."""
    if seed is not None:
        return seed
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return seed
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise RuntimeError("an unseeded split / shuffle inside a multi-rank job needs the process group to be "
                               "initialised first (every rank would draw its own split); pass a seed or call "
                               "torch.distributed.init_process_group before splitting")
        return seed
    box = [random.SystemRandom().randrange(1 << 62) if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


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


class ArrayDataset(Sequence):
    """An in-memory sequence of images with the reference's split contract (``DNAnet/data/data_models/base.py:69-111``,
    ``DNAnet/data/split.py:8-40``): ``split(fraction, seed)`` shuffles with ``random.seed(seed)`` and cuts at
    ``int(len * fraction)``; ``split_k_fold(k, seed)`` deals the shuffled images into k folds whose sizes differ by at
    most one and returns (train, test) pairs, fold i being the test set of pair i."""

    def __init__(self, data: Sequence = (), shuffle: bool = False, group_replicas_in_split: bool = False):
        self._data = list(data)
        self.shuffle = shuffle
        # Shuffled iteration inside a multi-rank job: every rank must walk the epochs in the same order (fit() slices each
        # batch by rank).  The agreement is ONE collective, here at construction (every rank builds the same datasets in
        # the same order); __iter__ itself never communicates, so a rank may iterate on its own (rank 0 scoring in
        # evaluate.py) without anybody joining.  Outside a group: a fresh generator per iteration, as the reference.
        seed = shared_seed(None) if shuffle else None
        self._rng = random.Random(seed) if seed is not None else None
        # the `group_replicas_in_split` key of the reference's data configs (``config/data/dnanet_rd.yaml``,
        # ``hid_dataset.py:344-412``): replicas of one mixture (file names sharing the prefix "1A2") stay on one side of a
        # split and every number of contributors (the prefix's last digit) is divided in the same proportion
        self.group_replicas_in_split = group_replicas_in_split

    def __len__(self) -> int:
        return len(self._data)

    def __getitem__(self, index):
        return self._data[index]

    def __iter__(self):
        if self.shuffle:
            rng = self._rng or random.Random()              # base.py:62 of the reference: random.Random().sample(...)
            return iter(rng.sample(self._data, len(self._data)))
        return iter(self._data)

    def split(self, fraction: float, seed: Optional[float] = None) -> Tuple["ArrayDataset", "ArrayDataset"]:
        if not 0 < fraction < 1:
            raise ValueError(f"Fraction should be between 0 and 1, got {fraction}.")
        if self.group_replicas_in_split:
            return self._split_grouped(fraction, seed)
        random.seed(shared_seed(seed))
        order = random.sample(self._data, len(self._data))
        cut = int(len(order) * fraction)
        return ArrayDataset(order[:cut], self.shuffle), ArrayDataset(order[cut:], self.shuffle)

    def split_k_fold(self, n_folds: int, seed: Optional[float] = None) -> List[Tuple["ArrayDataset", "ArrayDataset"]]:
        if n_folds < 2 or n_folds > len(self._data):
            raise ValueError(f"Cannot make {n_folds} folds out of {len(self._data)} images.")
        seed = shared_seed(seed)
        if self.group_replicas_in_split:
            # per number of contributors: shuffle the mixtures, deal them into k folds, take their replicas along
            random.seed(seed)
            folds = [[] for _ in range(n_folds)]
            for by_prefix in self._replica_groups().values():
                prefixes = list(by_prefix)
                random.shuffle(prefixes)
                for i, part in enumerate(_deal_in_folds(prefixes, n_folds, seed)):
                    folds[i].extend(im for prefix in part for im in by_prefix[prefix])
        else:
            folds = _deal_in_folds(self._data, n_folds, seed)
        for fold in folds:
            random.shuffle(fold)
        pairs = []
        for i in range(n_folds):
            train = [im for j, fold in enumerate(folds) if j != i for im in fold]
            pairs.append((ArrayDataset(train, self.shuffle), ArrayDataset(folds[i], self.shuffle)))
        return pairs


    def _replica_groups(self) -> Dict[str, Dict[str, list]]:
        """number of contributors -> mixture prefix -> its images, both in order of first appearance (``hid_dataset.py:414-426``)."""
        groups: Dict[str, Dict[str, list]] = {}
        for im in self._data:
            prefix = rd_prefix(os.path.splitext(os.path.basename(str(im.path)))[0])
            groups.setdefault(prefix[-1], {}).setdefault(prefix, []).append(im)
        return groups

    def _split_grouped(self, fraction: float, seed) -> Tuple["ArrayDataset", "ArrayDataset"]:
        random.seed(shared_seed(seed))
        first, second = [], []
        for by_prefix in self._replica_groups().values():
            prefixes = list(by_prefix)
            cut = int(len(prefixes) * fraction)
            random.shuffle(prefixes)
            first.extend(im for prefix in prefixes[:cut] for im in by_prefix[prefix])
            second.extend(im for prefix in prefixes[cut:] for im in by_prefix[prefix])
        random.shuffle(first)              # replicas still sit next to each other
        random.shuffle(second)
        return ArrayDataset(first, self.shuffle), ArrayDataset(second, self.shuffle)


def rd_prefix(stem: str) -> str:
    """'1A2_A01_01' -> '1A2' (``DNAnet/utils.py:35-47``): data set, mixture letter, number of contributors."""
    if not re.match(r"\d[ABCDEF]\d", stem[:3]):
        raise ValueError(f"Cannot take prefix from provided file name: {stem}")
    return stem.split("_")[0]


def _deal_in_folds(items: Sequence, n_folds: int, seed) -> List[list]:
    """``split_data_in_k_folds`` (``DNAnet/data/split.py:8-40``): (re)seed, shuffle a copy, fold i gets the i-th run of
    ``len // k`` items, what is left over goes one each to folds 0, 1, ..."""
    random.seed(seed)
    order = random.sample(list(items), len(items))
    base = len(order) // n_folds
    folds = [order[i * base:(i + 1) * base] for i in range(n_folds)]
    for k, item in enumerate(order[n_folds * base:]):
        folds[k % n_folds].append(item)
    return folds


class NpyDirDataset(ArrayDataset):
    """Profiles stored as ``<name>.npy`` (H, W, 1) RFU + ``<name>_annotation.npy`` (H, W, 1) mask, the layout of the
    reference's ``tests/resources/profiles/RD``.  Stands in for ``HIDDataset`` (``.hid`` parsing is outside the path);
    ``panel`` / ``scaler`` are optional and only needed by the allele caller."""

    def __init__(self, root: str, panel: Optional[str] = None, scaler: Optional[Sequence[float]] = None,
                 limit: Optional[int] = None, shuffle: bool = False):
        if not os.path.isdir(root):
            raise FileNotFoundError(f"Dataset directory {root} does not exist.")
        pan = Panel(panel) if panel else None
        images = []
        for path in sorted(glob.glob(os.path.join(root, "*.npy"))):
            if path.endswith("_annotation.npy"):
                continue
            rfu = np.load(path)
            ann_path = path[:-4] + "_annotation.npy"
            ann = Annotation(image=np.load(ann_path).astype(np.int8)) if os.path.isfile(ann_path) else None
            sc = np.linspace(scaler[0], scaler[1], rfu.shape[1]) if scaler is not None else None
            images.append(ArrayImage(data=rfu.astype(np.int16), annotation=ann, path=path, meta={}, _scaler=sc, _panel=pan))
            if limit and len(images) >= limit:
                break
        if not images:
            raise FileNotFoundError(f"No .npy profiles found in {root}.")
        super().__init__(images, shuffle)


class SyntheticDataset(ArrayDataset):
    """``n`` synthetic electropherograms of SURVEY 8d's generator (``dnanet_amd/synthetic.py``)."""

    def __init__(self, n: int, rows: int = 5, width: int = 4096, seed: int = 1234, shuffle: bool = False):
        from .synthetic import make_batch
        x, t = make_batch(n, rows, width, seed=seed, unique=n)
        images = [ArrayImage(data=np.moveaxis(x[i], 0, -1).astype(np.int16),
                             annotation=Annotation(image=np.moveaxis(t[i], 0, -1).astype(np.int8)),
                             path=f"synthetic://{seed}/{i}", meta={}) for i in range(n)]
        super().__init__(images, shuffle)
