"""YAML config -> model, with the reference's registry contract (``config_io.py:28-36,39-79,116-170``).

``load_model('config/models/unet.yaml')`` / ``load_model('unet')`` / ``load_model(<dir with one *.yaml and a
*.pt>)`` behave as in the reference, including the ``FileNotFoundError`` / ``ValueError`` cases of the
directory form.  The reference reads YAML through the third-party ``confidence`` package (absent here);
plain PyYAML gives the same mappings for these files.  Optional extra model key: ``dtype: fp32 | bf16``.
"""
from __future__ import annotations

import glob
import os
from pathlib import Path
from typing import Any, Iterable, Mapping, MutableMapping, Optional

import yaml

from functools import partial

from .caching import ArrowCacheDataset, hid_dataset
from .data import ArrayDataset, NpyDirDataset, SyntheticDataset
from .metrics import METRICS
from .model import DNANet_UNet, HumanAnalysis, Model

MODELS = {"model": {"unet": DNANet_UNet, "human_analysis": HumanAnalysis}}      # config_io.py:29
# The reference registers its .hid reader here (``config_io.py:28``); parsing .hid files is outside the hot path, so
# the datasets of this build are array-backed.  A reference ``HIDDataset`` can be passed to the model API directly.
DATASETS = {"dataset": {"npy": NpyDirDataset, "synthetic": SyntheticDataset, "arrow_cache": ArrowCacheDataset, "hid": hid_dataset}}


def get_config_path(name: str, kind: Optional[str] = None, root="." ) -> Path:
    if not name.endswith(".yaml"):
        name += ".yaml"
    parts = [str(root), "config"] + ([kind] if kind else []) + [name]
    return Path(os.path.join(*parts))


def load_config(path, kind: Optional[str] = None) -> Mapping[str, Any]:
    path = str(path)
    if not os.path.isfile(path):
        path = get_config_path(path, kind)
    with open(path, "r") as f:
        return yaml.safe_load(f) or {}


def parse_config(config: Mapping[str, Any], config_options: Optional[dict] = None) -> MutableMapping[str, Any]:
    """Recursively instantiate mappings that carry a registered ``name`` (``config_io.py:142-170``)."""

    def parse_item(key: str, value: Any) -> Any:
        if isinstance(value, Mapping):
            value = dict(value)
            name = value.pop("name", None)
            value = parse_config(value, config_options)
            if config_options and key in config_options and name in config_options[key]:
                return config_options[key][name](**value)
            return value
        if isinstance(value, Iterable) and not isinstance(value, (str, bytes, bytearray)):
            return [parse_item(key, v) for v in value]
        return value

    return {k: parse_item(k, v) for k, v in config.items()}


def load_training_config(source) -> MutableMapping[str, Any]:
    return parse_config(load_config(source, kind="training"))["training"]


def load_model(source) -> Model:
    source = str(source)
    if os.path.isdir(source):
        yaml_files = glob.glob(os.path.join(source, "*.yaml"))
        if len(yaml_files) == 0:
            raise FileNotFoundError(f"No config file found in {source}.")
        if len(yaml_files) > 1:
            raise ValueError(f"More than one config file found in {source}.")
        if not glob.glob(os.path.join(source, "*.pt")):
            raise FileNotFoundError(f"No checkpoint (.pt) file found in {source}.")
        config = load_config(yaml_files[0])
        model = parse_config({"model": config["model"]}, MODELS)["model"]
        model.load(source)
        return model
    return parse_config(load_config(source, kind="models"), MODELS)["model"]


def load_dataset(source) -> ArrayDataset:
    """``config/data/<name>.yaml`` -> dataset (``config_io.py:107-113``)."""
    return parse_config(load_config(source, kind="data"), DATASETS)["dataset"]


def load_evaluation_config(source) -> MutableMapping[str, Any]:
    """Metric name (+ ``_<kwarg>_<value>`` when it has arguments) -> callable(images=, predictions=)
    (``config_io.py:82-96``)."""
    config = load_config(source, kind="evaluation")
    metrics = {}
    for entry in config["evaluation"]["metrics"]:
        entry = dict(entry)
        name = entry.pop("name")
        if name not in METRICS:
            raise KeyError(f"Unknown metric {name!r}; available: {sorted(METRICS)}")
        label = name + (f"_{next(iter(entry))}_{entry[next(iter(entry))]}" if entry else "")
        metrics[label] = partial(METRICS[name], **entry)
    return metrics


def dump_config(path, data_config, model_config, training_config=None, validation_config=None, dataset=None, model=None):
    """Write the configs a run was started with into its output directory so it can be retraced
    (``config_io.py:173-230``); returns what was written."""
    out = {"data": load_config(data_config, kind="data"), "model": load_config(model_config, kind="models")}
    if training_config:
        out["training"] = load_config(training_config, kind="training")
    if validation_config is not None:
        try:
            out["validation"] = float(validation_config)
        except (TypeError, ValueError):
            out["validation"] = load_config(validation_config, kind="data")
    if dataset is not None:
        out["n_images"] = len(dataset)
    with open(path, "w") as f:
        yaml.safe_dump(out, f, sort_keys=False)
    return out
