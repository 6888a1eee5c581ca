"""k-fold cross-validation: ``python cross_validate.py -d <data> -m <model> -t <training> -e <evaluation> -k K [-o DIR] [-v F] [-c DIR]``.

Same flags and flow as the reference's ``cross_validate.py`` (folds from ``split_k_fold(k, seed=42)``, a fresh model per
fold, ``fold_<i>/metrics.txt``, means in ``agg_metrics.txt``).  Folds are independent: with ``--gpus N`` fold
i runs on GPU i mod N, one process per fold, no collective.
"""
import argparse
import json
import os
import subprocess
import sys
from collections import defaultdict

from dnanet_amd.cli import LOGGER, start_run, write_metrics
from dnanet_amd.config_io import dump_config, load_dataset, load_evaluation_config, load_model, load_training_config
from dnanet_amd.model import TrainableModel


def run_fold(i, train_set, test_set, model_config, training_config, metrics, output_dir, validation_config=None,
             checkpoint_dir=None, data_config=None):
    fold_dir = os.path.join(output_dir, f"fold_{i}")
    os.makedirs(fold_dir, exist_ok=True)
    validation_set = None
    if validation_config:
        fraction = float(validation_config)
        LOGGER.info("Validation split found, using %s%% to create a validation set...", fraction * 100)
        train_set, validation_set = train_set.split(1 - fraction)
    model = load_model(model_config)
    if not isinstance(model, TrainableModel):
        raise ValueError(f"Model {model} is not trainable.")
    if checkpoint_dir:
        model.load(checkpoint_dir)
    kwargs = load_training_config(training_config)
    kwargs.update({"validation_set": validation_set, "checkpoint_dir": fold_dir})
    LOGGER.info("Fold %d: training on %d images, testing on %d", i, len(train_set), len(test_set))
    try:
        model.fit(train_set, **kwargs)
    except KeyboardInterrupt:
        LOGGER.info("Training interrupted!")
    predictions = model.predict_batch(test_set)
    results = write_metrics(metrics, test_set, predictions, os.path.join(fold_dir, "metrics.txt"))
    dump_config(os.path.join(fold_dir, "config.yaml"), data_config, model_config, training_config, validation_config,
                train_set, model)
    return results


def run(data_config, model_config, training_config, evaluation_config, nr_folds, output_dir=None, validation_config=None,
        checkpoint_dir=None, gpus=1, only_fold=None):
    output_dir = start_run("crossval", output_dir, "log_crossval.txt" if only_fold is None else f"log_fold_{only_fold}.txt")
    dataset = load_dataset(data_config)
    metrics = load_evaluation_config(evaluation_config)
    LOGGER.info("Splitting dataset into %d folds...", nr_folds)
    splits = dataset.split_k_fold(nr_folds, seed=42)

    if only_fold is not None:               # worker process of a multi-GPU run
        train_set, test_set = splits[only_fold - 1]
        run_fold(only_fold, train_set, test_set, model_config, training_config, metrics, output_dir, validation_config,
                 checkpoint_dir, data_config)
        return

    if gpus > 1:
        # one process per fold, fold i on GPU (i-1) mod gpus, at most `gpus` at a time; the split is deterministic
        # (seed 42), so every worker derives the same folds
        # fold -> GPU through the parent's own lease: if HIP_VISIBLE_DEVICES already restricts this process (e.g. "4,5"),
        # the workers get entries of THAT list, never a bare index outside it
        visible = [d for d in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if d.strip()]
        if not visible:
            import torch
            visible = [str(d) for d in range(max(1, torch.cuda.device_count()))]
        pending = list(range(1, nr_folds + 1))
        while pending:
            batch, pending = pending[:gpus], pending[gpus:]
            procs = []
            for slot, i in enumerate(batch):
                env = dict(os.environ, HIP_VISIBLE_DEVICES=visible[slot % len(visible)], HSA_ENABLE_IPC_MODE_LEGACY="0")
                for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):     # a fold worker is a single-device job of its own
                    env.pop(key, None)
                cmd = [sys.executable, os.path.abspath(__file__), "-d", data_config, "-m", model_config, "-t", training_config,
                       "-e", evaluation_config, "-k", str(nr_folds), "-o", output_dir, "--only-fold", str(i)]
                if validation_config:
                    cmd += ["-v", str(validation_config)]
                if checkpoint_dir:
                    cmd += ["-c", checkpoint_dir]
                procs.append(subprocess.Popen(cmd, env=env))
            for pr in procs:
                if pr.wait() != 0:
                    raise RuntimeError(f"fold worker exited with code {pr.returncode}")
    else:
        for i, (train_set, test_set) in enumerate(splits, start=1):
            run_fold(i, train_set, test_set, model_config, training_config, metrics, output_dir, validation_config,
                     checkpoint_dir, data_config)

    agg = defaultdict(list)
    for i in range(1, nr_folds + 1):
        with open(os.path.join(output_dir, f"fold_{i}", "metrics.txt")) as f:
            for name, value in json.load(f).items():
                if isinstance(value, float):
                    agg[name].append(value)
    # every metric's mean goes to the file (the reference writes only the last one, cross_validate.py:121-126)
    with open(os.path.join(output_dir, "agg_metrics.txt"), "w") as f:
        for name, values in agg.items():
            line = f"Mean {name}: {sum(values) / len(values)}"
            LOGGER.info(line)
            f.write(line + "\n")
    return {name: sum(v) / len(v) for name, v in agg.items()}


def get_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("-d", "--data-config", required=True, help="data config: path to a .yaml or a name under config/data/")
    p.add_argument("-m", "--model-config", required=True, help="model config: path or name under config/models/")
    p.add_argument("-t", "--training-config", required=True, help="fit() keyword arguments: path or name under config/training/")
    p.add_argument("-e", "--evaluation-config", required=True, help="metrics: path or name under config/evaluation/")
    p.add_argument("-k", "--nr-folds", type=int, required=True, help="number of folds")
    p.add_argument("-o", "--output-dir", default=None, help="directory the results are written to")
    p.add_argument("-v", "--validation-config", default=None, help="fraction of each training fold held out for validation")
    p.add_argument("-c", "--checkpoint-dir", default=None, help="start every fold from the checkpoint in this directory")
    p.add_argument("--gpus", type=int, default=1, help="run folds in parallel, one process per GPU (not in the reference)")
    p.add_argument("--only-fold", type=int, default=None, help=argparse.SUPPRESS)
    return p


if __name__ == "__main__":
    run(**vars(get_parser().parse_args()))
