"""Apply a model to a dataset and score it: ``python evaluate.py -d <data> -m <model> [-e <evaluation>] [-o DIR] [-c DIR] [-p] [-s F] [-rs N]``.

Same flags and flow as the reference's ``evaluate.py``: ``predict_batch`` over the (optionally split) dataset, the
configured metrics to ``metrics.txt``, optionally the predictions to ``predictions.json``.
"""
import argparse
import os

from dnanet_amd.cli import LOGGER, save_predictions, start_run, write_metrics
from dnanet_amd.config_io import dump_config, load_dataset, load_evaluation_config, load_model


def run(data_config, model_config, evaluation_config=None, output_dir=None, checkpoint_dir=None, save_preds=False,
        split=None, seed=None):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                   # one rank per GPU: pick it before the model is built on "cuda"
        import torch
        import torch.distributed as dist
        if torch.cuda.is_available():
            # (more ranks than GPUs -- a rehearsal on a small box -- share the devices round robin: nothing here needs RCCL)
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        if not dist.is_initialized():
            # before the dataset is split: an unseeded split is agreed on through the group (dnanet_amd.data.shared_seed).
            # gloo: the only exchanges are that seed and the ordered gather of host objects
            dist.init_process_group("gloo")
    output_dir = start_run("evaluate", output_dir, "log_evaluation.txt")
    LOGGER.info("Loading model...")
    model = load_model(model_config)
    if checkpoint_dir:
        model.load(checkpoint_dir)
    LOGGER.info("Loading dataset...")
    dataset = load_dataset(data_config)
    if split:
        # the complement of the training split: same seed, the second part
        LOGGER.info("Splitting dataset, using %s%% for evaluation", split * 100)
        _, dataset = dataset.split(1 - split, seed=seed)
    LOGGER.info("Applying model...")
    # ONE iteration of the dataset, on every rank: a `shuffle: true` dataset draws a new order per iteration, and the
    # metrics below pair images with predictions by position
    images = list(dataset)
    if world > 1:
        # launched with torch.distributed.run: every rank predicts its shard, rank 0 scores
        import torch.distributed as dist
        from dnanet_amd.distributed import predict_batch_sharded
        predictions = predict_batch_sharded(model, images)
        if dist.get_rank() != 0:
            return {}
    else:
        predictions = model.predict_batch(images)
    results = {}
    if evaluation_config:
        LOGGER.info("Computing metrics...")
        results = write_metrics(load_evaluation_config(evaluation_config), images, predictions,
                                os.path.join(output_dir, "metrics.txt"))
    if save_preds:
        LOGGER.info("Saving predictions to JSON...")
        save_predictions(predictions, os.path.join(output_dir, "predictions.json"))
    dump_config(os.path.join(output_dir, "config.yaml"), data_config, model_config, None, None, dataset, model)
    return results


def get_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("-d", "--data-config", required=True, help="data config: path to a .yaml or a name under config/data/")
    p.add_argument("-m", "--model-config", required=True, help="model config: path, name under config/models/, or a model directory")
    p.add_argument("-e", "--evaluation-config", default=None, help="metrics: path or name under config/evaluation/")
    p.add_argument("-o", "--output-dir", default=None, help="directory the results are written to")
    p.add_argument("-c", "--checkpoint-dir", default=None, help="directory holding the .pt checkpoint to load")
    p.add_argument("-p", "--save-preds", action="store_true", help="also write predictions.json")
    p.add_argument("-s", "--split", type=float, default=None, help="fraction in (0, 1) of the dataset to evaluate on")
    p.add_argument("-rs", "--seed", type=int, default=None, help="seed of the split (the one used for training)")
    return p


if __name__ == "__main__":
    run(**vars(get_parser().parse_args()))
