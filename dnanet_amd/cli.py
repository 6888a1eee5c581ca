"""Shared plumbing of the command-line entry points (train.py, evaluate.py, cross_validate.py at the repo root).

The three scripts keep the reference's flags and flow (``train.py:16-186``, ``evaluate.py:15-145``,
``cross_validate.py:22-180``) so a command line written for the reference runs here; what they call underneath is this
package.  mlflow autologging of the reference is not reproduced (the package is absent and outside the path).
"""
from __future__ import annotations

import json
import logging
import os
from datetime import datetime
from typing import Mapping, Optional, Sequence

LOGGER = logging.getLogger("dnanet")


def start_run(kind: str, output_dir: Optional[str], log_name: str) -> str:
    """Create the output directory (default ``output/<kind>_<timestamp>``), attach a file log, return the directory."""
    if not output_dir:
        output_dir = os.path.join("output", f"{kind}_{datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}")
    os.makedirs(output_dir, exist_ok=True)
    if not LOGGER.handlers:
        logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s %(message)s")
    if int(os.environ.get("RANK", "0")) != 0:
        # one rank per GPU: rank 0 owns the log file, the others only report problems
        LOGGER.setLevel(logging.WARNING)
        return output_dir
    LOGGER.setLevel(logging.INFO)
    handler = logging.FileHandler(os.path.join(output_dir, log_name))
    handler.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
    LOGGER.addHandler(handler)
    LOGGER.info("Logs will be written to %s", handler.baseFilename)
    return output_dir


def write_metrics(metrics: Mapping, images: Sequence, predictions: Sequence, path: str) -> dict:
    """Evaluate every configured metric, write them as indented JSON to ``path`` and return the values."""
    results = {}
    for name, func in metrics.items():
        label = f"{name} ({func.keywords})" if getattr(func, "keywords", None) else name
        results[label] = float(func(images=images, predictions=predictions))
    with open(path, "w") as f:
        f.write(json.dumps(results, indent=4))
    LOGGER.info("Results:\n%s", json.dumps(results, indent=4))
    LOGGER.info("Results written to %s", path)
    return results


def save_predictions(predictions: Sequence, path: str):
    with open(path, "w") as f:
        json.dump([p.to_dict() for p in predictions], f)


def split_validation(dataset, validation_config):
    """``-v`` is either a fraction of the training set to hold out or the name / path of a data config."""
    from .config_io import load_dataset
    if not validation_config:
        return dataset, None
    try:
        fraction = float(validation_config)
    except ValueError:
        LOGGER.info("Validation config found, loading validation set...")
        return dataset, load_dataset(validation_config)
    LOGGER.info("Validation split found, using %s%% to create a validation set...", fraction * 100)
    return dataset.split(1 - fraction)
