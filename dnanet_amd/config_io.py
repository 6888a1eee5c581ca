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
from typing import Any, Iterable, Mapping, MutableMapping, Optional, Union

import yaml

from .model import DNANet_UNet, Model

MODELS = {"model": {"unet": DNANet_UNet}}


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
