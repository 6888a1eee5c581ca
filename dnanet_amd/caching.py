"""Reader / writer of the reference's arrow cache of parsed profiles (``DNAnet/data/caching.py:18-125``).

The reference stores a ``HIDDataset`` as a Hugging Face ``datasets`` directory -- ``state.json`` naming the
``data-*.arrow`` files (arrow IPC stream format) with the columns ``images`` (5 x 4096 x 1 int16), ``annotations``
(5 x 4096 x 1 int8), ``hid_paths``, ``scalers`` (4096 float64), ``panel_contents`` and ``called_alleles`` (JSON of
``Marker`` / ``Allele`` dataclasses).  This module reads that format with ``pyarrow`` alone and hands the hot path what
it wants: ONE contiguous ``(N, H, W, 1)`` int16 block and one int8 block for the whole cache (the images are views into
them), so ``DNANet_UNet.make_resident`` / ``get_inputs`` move the dataset to HBM in a single copy instead of stacking
N small arrays.  ``write_cache`` writes the same directory layout through ``datasets`` (which the reference uses), so a
cache written here is readable by the reference's ``load_from_disk`` and vice versa.
"""
from __future__ import annotations

import json
import re
import os
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .data import Allele, Annotation, ArrayDataset, ArrayImage, Marker, Panel

COLUMNS = ("images", "annotations", "hid_paths", "scalers", "panel_contents", "called_alleles")


def markers_from_json(text: str) -> List[Marker]:
    """``dict_to_marker_list(..., as_json=True)`` (``DNAnet/utils.py:83-96``)."""
    return [Marker(m["dye_row"], m["name"], [Allele(**a) for a in m["alleles"]]) for m in json.loads(text)]


def markers_to_json(markers: Sequence[Marker]) -> str:
    """``marker_list_to_dict(..., as_json=True)`` (``DNAnet/utils.py:73-80``)."""
    return json.dumps([{"dye_row": m.dye_row, "name": m.name,
                        "alleles": [{"name": a.name, "base_pair": a.base_pair, "left_bin": a.left_bin,
                                     "right_bin": a.right_bin, "height": a.height} for a in m.alleles]} for m in markers])


def noc_from_rd_file_name(name: str) -> Optional[str]:
    """``get_noc_from_rd_file_name`` (``DNAnet/utils.py:62-70``): '1A2_A01_01.hid' -> '2p'."""
    stem = Path(name).name
    return f"{stem[2]}p" if re.match(r"\d[ABCDEF]\d", stem[:3]) else None     # the reference's test, the one data.rd_prefix uses


def _fixed_block(column, dtype) -> np.ndarray:
    """An Array3D column (extension type over list<list<list<T>>>, or its bare storage) as one (N, H, W, C) numpy block."""
    import pyarrow as pa
    blocks = []
    for chunk in column.chunks:
        storage = chunk.storage if isinstance(chunk, pa.ExtensionArray) else chunk
        n = len(storage)
        shape = [n]
        level = storage
        while pa.types.is_list(level.type) or pa.types.is_large_list(level.type) or pa.types.is_fixed_size_list(level.type):
            inner = level.flatten()
            shape.append(len(inner) // max(1, int(np.prod(shape))))
            level = inner
        flat = level.to_numpy(zero_copy_only=False)
        blocks.append(np.asarray(flat, dtype=dtype).reshape(shape))
    return np.concatenate(blocks) if len(blocks) != 1 else blocks[0]


def read_cache(cache_directory, limit: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, list]:
    """One cache directory -> (images (N,H,W,1) int16, annotations (N,H,W,1) int8, rows) where ``rows`` holds the other
    columns per profile (``read_from_hf_cache``, ``caching.py:44-58``)."""
    import pyarrow as pa
    cache_directory = str(cache_directory)
    state_path = os.path.join(cache_directory, "state.json")
    if not os.path.isfile(state_path):
        raise FileNotFoundError(f"{cache_directory} is not an arrow cache (no state.json)")
    with open(state_path) as f:
        files = [d["filename"] for d in json.load(f)["_data_files"]]
    tables = []
    for name in files:
        path = os.path.join(cache_directory, name)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"arrow cache {cache_directory} names {name}, which is missing (a Git-LFS pointer not pulled?)")
        with pa.memory_map(path) as source:
            try:
                tables.append(pa.ipc.open_stream(source).read_all())
            except pa.ArrowInvalid:
                tables.append(pa.ipc.open_file(source).read_all())
    table = pa.concat_tables(tables)
    missing = [c for c in COLUMNS if c not in table.column_names]
    if missing:
        raise ValueError(f"arrow cache {cache_directory} lacks the columns {missing}")
    if limit:
        table = table.slice(0, limit)
    images = _fixed_block(table.column("images"), np.int16)
    annotations = _fixed_block(table.column("annotations"), np.int8)
    rows = [dict(zip(COLUMNS[2:], vals)) for vals in zip(*(table.column(c).to_pylist() for c in COLUMNS[2:]))]
    return images, annotations, rows


class ArrowCacheDataset(ArrayDataset):
    """The cached profiles of ``cache_path`` (one cache directory, or a directory of them, ``_load_cached_hf_data``,
    ``caching.py:18-41``) as a dataset for ``fit`` / ``predict_batch``.  ``block_inputs`` / ``block_targets`` are the
    contiguous arrays behind the images."""

    def __init__(self, cache_path, limit: Optional[int] = None, shuffle: bool = False, group_replicas_in_split: bool = False):
        cache_path = str(cache_path)
        if not os.path.isdir(cache_path):
            raise FileNotFoundError(f"Cache directory {cache_path} does not exist.")
        entries = sorted(os.listdir(cache_path))
        dirs = [cache_path] if any(e.endswith(".arrow") for e in entries) else \
            [os.path.join(cache_path, e) for e in entries if os.path.isdir(os.path.join(cache_path, e))]
        xs, ts, rows = [], [], []
        for d in dirs:
            left = None if not limit else limit - len(rows)
            if limit and left <= 0:
                break
            x, t, r = read_cache(d, left)
            xs.append(x); ts.append(t); rows += r
        if not rows:
            raise FileNotFoundError(f"No cached profiles found in {cache_path}.")
        self.block_inputs = np.ascontiguousarray(np.concatenate(xs) if len(xs) > 1 else xs[0])
        self.block_targets = np.ascontiguousarray(np.concatenate(ts) if len(ts) > 1 else ts[0])
        panels = {}
        images = []
        for i, row in enumerate(rows):
            key = row["panel_contents"]
            if key not in panels:                     # profiles of one kit share the (large) panel object
                panels[key] = Panel(panel_contents=markers_from_json(key))
            meta = {"called_alleles": markers_from_json(row["called_alleles"]), "noc": noc_from_rd_file_name(row["hid_paths"])}
            images.append(ArrayImage(data=self.block_inputs[i], annotation=Annotation(image=self.block_targets[i]),
                                     path=row["hid_paths"], meta=meta, _scaler=np.array(row["scalers"]), _panel=panels[key]))
        super().__init__(images, shuffle, group_replicas_in_split)


def hid_dataset(root=None, use_cache: bool = False, cache_path=None, limit: Optional[int] = None, shuffle: bool = False,
                group_replicas_in_split: bool = False, **parsing_options) -> ArrowCacheDataset:
    """The reference's dataset name ``hid`` (``config/data/dnanet_rd.yaml``, ``hid_dataset.py:30-120``) for the case this build
    covers: ``use_cache: true`` with a ``cache_path`` written by the reference (or by ``write_cache``).  The keys that steer the
    ``.hid`` parser (annotations, ladders, thresholds, ...) are accepted and unused -- the cache already holds their result;
    reading raw ``.hid`` files stays with the reference."""
    if not (use_cache and cache_path):
        raise NotImplementedError("dataset `hid`: this build reads the reference's arrow cache (`use_cache: true` + `cache_path`); "
                                  "parsing raw .hid files is done by the reference (run it once with `cache_path` set to write the cache)")
    if parsing_options.get("ground_truth_as_annotations"):
        raise NotImplementedError("dataset `hid`: `ground_truth_as_annotations` replaces the annotated alleles by donor genotypes read from "
                                  "the reference's resource files (hid_dataset.py:130-136); load such a dataset with the reference and "
                                  "hand it to the model API")
    return ArrowCacheDataset(cache_path, limit=limit, shuffle=shuffle, group_replicas_in_split=group_replicas_in_split)


def write_cache(cache_path, images: Sequence) -> None:
    """``write_to_hf_cache`` (``caching.py:82-125``): the same columns and feature types, saved with ``datasets``."""
    import datasets
    from datasets import Array3D, Features, Value
    Path(cache_path).mkdir(parents=True, exist_ok=True)
    first = np.asarray(images[0].data)
    features = Features({"images": Array3D(shape=tuple(first.shape), dtype="int16"),
                         "annotations": Array3D(shape=tuple(first.shape), dtype="int8"),
                         "hid_paths": Value(dtype="string"), "scalers": datasets.Sequence(feature=Value(dtype="float64")),
                         "panel_contents": Value(dtype="string"), "called_alleles": Value(dtype="string")})
    mapping = {c: [] for c in COLUMNS}
    for im in images:
        mapping["images"].append(np.asarray(im.data, dtype=np.int16))
        mapping["annotations"].append(np.asarray(im.annotation.image, dtype=np.int8))
        mapping["hid_paths"].append(str(im.path))
        mapping["scalers"].append(np.asarray(im._scaler, dtype=np.float64).tolist())
        mapping["panel_contents"].append(markers_to_json(im._panel._panel))
        mapping["called_alleles"].append(markers_to_json(im.meta.get("called_alleles", [])))
    datasets.Dataset.from_dict(mapping=mapping, features=features).save_to_disk(str(cache_path))
