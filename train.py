"""Train a model on a dataset: ``python train.py -d <data> -m <model> -t <training> [-s F] [-o DIR] [-v CFG|F] [-c DIR] [-rs N]``.

Same flags and order of operations as the reference's ``train.py`` (load model, optional checkpoint, load + split
dataset, ``fit``, ``save``, dump the config next to the weights); runs on ``dnanet_amd``.
"""
import argparse
import os

from dnanet_amd.cli import LOGGER, split_validation, start_run
from dnanet_amd.config_io import dump_config, load_dataset, load_model, load_training_config
from dnanet_amd.model import TrainableModel


def run(data_config, model_config, training_config, split=None, output_dir=None, validation_config=None,
        checkpoint_dir=None, seed=None):
    # Under torch.distributed.run (one rank per GPU) fit() trains data parallel: `batch_size` of the training config is the
    # global batch, gradients are all-reduced over RCCL, rank 0 writes the outputs (not in the reference).
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        # DNANET_DIST_BACKEND=gloo: rehearsal of the multi-rank path with ranks sharing a GPU (RCCL wants a device per rank);
        # collectives on device tensors are then staged through the host (dnanet_amd/distributed.py)
        backend = os.environ.get("DNANET_DIST_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local if backend == "nccl" else local % max(1, torch.cuda.device_count()))
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
            else:
                dist.init_process_group(backend)
        if not output_dir:                              # the default name carries a timestamp: rank 0 picks it
            from datetime import datetime
            box = [os.path.join("output", f"train_{datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}")]
            dist.broadcast_object_list(box, src=0)
            output_dir = box[0]
    output_dir = start_run("train", output_dir, "log_training.txt")
    LOGGER.info("Loading model...")
    model = load_model(model_config)
    if not isinstance(model, TrainableModel):
        raise ValueError(f"Model {model} is not trainable.")
    if checkpoint_dir:
        model.load(checkpoint_dir)
        LOGGER.info("Loaded previous model checkpoint from %s", checkpoint_dir)
    else:
        LOGGER.info("Will start training from scratch")

    LOGGER.info("Loading dataset...")
    dataset = load_dataset(data_config)
    if split:
        LOGGER.info("Splitting dataset, using %s%% for training", split * 100)
        dataset, _ = dataset.split(split, seed=seed)
    dataset, validation_set = split_validation(dataset, validation_config)

    kwargs = load_training_config(training_config)
    kwargs.update({"validation_set": validation_set, "checkpoint_dir": output_dir})
    LOGGER.info("Starting training...")
    try:
        model.fit(dataset, **kwargs)
    except KeyboardInterrupt:
        LOGGER.info("Training interrupted!")
    model.save(output_dir)
    LOGGER.info("Saved model to %s", output_dir)
    if rank == 0:
        dump_config(f"{output_dir}/config.yaml", data_config, model_config, training_config, validation_config, dataset, model)
    return output_dir


def get_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("-d", "--data-config", required=True, help="data config: path to a .yaml or a name under config/data/")
    p.add_argument("-m", "--model-config", required=True, help="model config: path or name under config/models/")
    p.add_argument("-t", "--training-config", required=True, help="fit() keyword arguments: path or name under config/training/")
    p.add_argument("-s", "--split", type=float, default=None, help="fraction in (0, 1) of the dataset to train on")
    p.add_argument("-o", "--output-dir", default=None, help="directory the trained model is saved to")
    p.add_argument("-v", "--validation-config", default=None,
                   help="data config of a validation set, or the fraction of the training set to hold out")
    p.add_argument("-c", "--checkpoint-dir", default=None, help="resume from the checkpoint in this directory")
    p.add_argument("-rs", "--seed", type=int, default=None, help="seed of the split (use the same one when evaluating)")
    return p


if __name__ == "__main__":
    run(**vars(get_parser().parse_args()))
