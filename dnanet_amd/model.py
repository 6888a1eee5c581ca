"""``DNANet_UNet`` -- the reference's model API on top of the MI355X kernels.

Mirrors ``DNAnet/models/base.py:14-51`` (``Model`` / ``TrainableModel``) and
``DNAnet/models/segmentation/trainable_unet.py:33-470`` (``DNANet_UNet``): same constructor keywords,
``predict`` / ``predict_batch`` / ``predict_raw`` / ``sigmoid`` / ``get_inputs`` / ``get_targets`` /
``fit`` (same keyword arguments, defaults and ``TypeError``s) / ``epoch`` / ``step`` / ``save`` / ``load``
and the ``_model`` / ``_device`` / ``loss_fn`` / ``allele_caller`` attributes, and the same
``DNANet_UNet_checkpoint.pt`` state-dict wire format.  The arithmetic runs in ``HipUNet``
(``dnanet_amd/unet.py``); nothing here falls back to the CPU.
"""
from __future__ import annotations

import logging
import os
import shutil
from abc import ABC, abstractmethod
from collections import defaultdict
from itertools import islice
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .allele_caller import NearestBasePairCaller
from .data import chunks
from .prediction import Prediction
from .unet import ExponentialLR, GraphedTrainStep, HipAdam, HipUNet

LOGGER = logging.getLogger("dnanet")
TORCH_DEFAULT_DEVICE = "cuda" if torch.cuda.is_available() else "cpu"     # base.py:11

try:  # optional experiment tracking, as in trainable_unet.py:212,239-241 (a no-op when not installed)
    import mlflow as _mlflow
except Exception:  # pragma: no cover
    _mlflow = None

try:
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None


def _log_metric(key, value, step):
    if _mlflow is not None:
        try:
            _mlflow.log_metric(key=key, value=value, step=step)
        except Exception:  # pragma: no cover
            pass


class Model(ABC):
    @abstractmethod
    def predict(self, image) -> Prediction:
        raise NotImplementedError

    def predict_batch(self, batch: Sequence) -> Sequence[Prediction]:
        return list(map(self.predict, batch))

    @abstractmethod
    def save(self, model_dir):
        raise NotImplementedError

    @abstractmethod
    def load(self, model_dir):
        raise NotImplementedError


class HumanAnalysis(Model):
    """The analyst as a "model" (``DNAnet/models/segmentation/human_analysis.py:13-37``, registry name ``human_analysis``):
    the manual annotations stored under ``called_alleles_manual`` in an image's meta come back as the called alleles, without
    the sex / Y markers the donor ground truth does not list, so that the allele metrics score analysts like a network."""

    def predict(self, image) -> Prediction:
        manual = image.meta.get("called_alleles_manual")
        if manual is None:
            raise ValueError("No `called_alleles_manual` found when running HumanAnalysis model, set "
                             "`ground_truth_as_annotations=True` in the data config file to load `called_alleles_manual`.")
        from .allele_caller import NON_AUTOSOMAL_MARKERS
        kept = [marker for marker in manual if marker.name not in NON_AUTOSOMAL_MARKERS]
        return Prediction(image=None, meta={"called_alleles": kept, "original_image_path": image.path})

    def save(self, model_dir):       # nothing to store
        pass

    def load(self, model_dir):
        pass


class TrainableModel(Model, ABC):
    @abstractmethod
    def fit(self, dataset, **kwargs):
        raise NotImplementedError


class BinaryAccuracy:
    """``torchmetrics.classification.BinaryAccuracy`` (threshold 0.5) as the reference uses it
    (``trainable_unet.py:295-303,411-415``); fed by the counts the fused Dice kernel produces."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.correct, self.total = 0.0, 0

    def update_counts(self, correct: float, total: int):
        self.correct += float(correct)
        self.total += int(total)

    def update(self, probabilities: torch.Tensor, target: torch.Tensor):
        self.update_counts(float(((probabilities >= 0.5) == (target != 0)).sum()), target.numel())

    def compute(self) -> float:
        return self.correct / self.total if self.total else 0.0

    def to(self, device):
        return self

    def __repr__(self):
        return "BinaryAccuracy()"


class _Loss:
    """What ``step`` returns: behaves like the scalar loss tensor of the reference's loop
    (``loss.backward()``, ``float(loss)``, ``trainable_unet.py:340-347``)."""

    def __init__(self, model: HipUNet, sums: torch.Tensor, count: int, metrics=None):
        self._model, self.sums, self.count = model, sums, count
        self._metrics = list(metrics or [])
        self._value = None

    def backward(self):
        self._model.backward()

    def item(self) -> float:
        if self._value is None:
            host = self.sums.cpu()                   # the one device->host sync of a step
            self._value = float(host[4])
            for metric in self._metrics:             # accuracy counts came out of the same kernel pass
                metric.update_counts(float(host[3]), self.count)
        return self._value

    __float__ = item

    def detach(self):
        return self


class _ResidentSet:
    """A dataset tensorised once and kept in HBM; batches are contiguous slices (views) when the images come in
    dataset order, otherwise one ``index_select``."""

    def __init__(self, images: Sequence, inputs: torch.Tensor, targets: torch.Tensor):
        self._images = list(images)                          # keeps the ids alive
        self._index = {id(im): i for i, im in enumerate(self._images)}
        self._arrays = {"inputs": inputs, "targets": targets}

    def gather(self, images: Sequence, what: str) -> Optional[torch.Tensor]:
        try:
            idx = [self._index[id(im)] for im in images]
        except KeyError:
            return None
        if not idx:
            return None
        arr = self._arrays[what]
        if idx == list(range(idx[0], idx[0] + len(idx))):
            return arr[idx[0]:idx[0] + len(idx)]
        return arr.index_select(0, torch.tensor(idx, device=arr.device))


class DNANet_UNet(TrainableModel):
    def __init__(self, depth: int, kernel_size: Tuple[int, int], num_filters: int = 64, device: Optional[str] = None,
                 apply_allele_caller: Optional[bool] = True, dtype: str = "fp32"):
        """Same keywords as the reference (``trainable_unet.py:38-58``); ``dtype`` ("fp32" | "bf16" | inference only: "bf16x3", "bf16x3@<levels>") selects
        exact-fp32 MFMA (default, reference numerics) or bf16 MFMA with fp32 accumulation."""
        self._device = device or TORCH_DEFAULT_DEVICE
        # On a 'cpu' device only the parameter tree exists (checkpoint I/O, config plumbing); any forward raises.
        self._model = HipUNet(depth, tuple(kernel_size), num_filters, self._device, dtype=dtype)
        self.loss_fn = self._dice
        self.allele_caller = NearestBasePairCaller() if apply_allele_caller else None
        self._resident: Optional[_ResidentSet] = None
        self._graph = {"enabled": False, "step": None, "warm": False}
        self._dp = None

    @staticmethod
    def _group_size() -> int:
        """Ranks of the default process group, 0 when there is none."""
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 0

    # ---- tensorisation (trainable_unet.py:60-89) ---------------------------------------------------
    @staticmethod
    def get_input(image) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(image.data), (2, 0, 1)), dtype=np.float32))

    def get_inputs(self, images: Sequence) -> torch.Tensor:
        """(N,1,H,W) fp32 on the device.  int16-representable data crosses PCIe as int16 (half the bytes,
        one copy for the whole batch) and is widened on the GPU."""
        if self._resident is not None:
            hit = self._resident.gather(images, "inputs")
            if hit is not None:
                return hit
        arr = np.stack([np.asarray(im.data) for im in images])               # (N,H,W,1)
        arr = np.ascontiguousarray(np.transpose(arr, (0, 3, 1, 2)))
        if arr.dtype.kind in "iu" and arr.size and int(arr.min()) >= -32768 and int(arr.max()) <= 32767:
            src = torch.from_numpy(arr.astype(np.int16, copy=False)).to(self._device, non_blocking=True)
            out = torch.empty(arr.shape, dtype=torch.float32, device=self._device)
            hip.check(hip.lib.dn_i16_to_f32(hip.ptr(src), hip.ptr(out), out.numel(), hip.stream_ptr()))
            return out
        return torch.from_numpy(arr.astype(np.float32, copy=False)).to(self._device)

    def get_targets(self, images: Sequence) -> torch.Tensor:
        if self._resident is not None:
            hit = self._resident.gather(images, "targets")
            if hit is not None:
                return hit
        arr = np.stack([np.asarray(im.annotation.image) for im in images])   # (N,H,W,1)
        arr = np.ascontiguousarray(np.transpose(arr, (0, 3, 1, 2)))
        return torch.from_numpy(arr).to(self._device)

    def make_resident(self, *datasets, limit_bytes: int = 16 << 30) -> bool:
        """Input pipeline: tensorise the given datasets ONCE and keep them in HBM (348 profiles =
        35 MB), so a training step gathers its batch on the device instead of stacking numpy arrays and crossing
        PCIe (``trainable_unet.py:60-89`` does that per step).  Images are matched by identity; a batch with an
        unknown image simply takes the per-step path.  ``fit`` calls this for its training / validation sets and
        drops the copy when it returns; image data must not be modified in between."""
        self.release_resident()
        images = [im for ds in datasets if ds is not None for im in ds]
        if not images or self._model.device.type != "cuda":
            return False
        try:
            first_x, first_t = np.asarray(images[0].data), np.asarray(images[0].annotation.image)
            total = len(images) * (first_x.size * 4 + first_t.size * first_t.dtype.itemsize)
            if total > limit_bytes:
                return False
            step = 1024
            xs = torch.cat([self.get_inputs(images[i:i + step]) for i in range(0, len(images), step)])
            ts = torch.cat([self.get_targets(images[i:i + step]) for i in range(0, len(images), step)])
        except (AttributeError, ValueError, TypeError):   # ragged shapes, images without annotation, ...
            return False
        self._resident = _ResidentSet(images, xs, ts)
        return True

    def release_resident(self):
        self._resident = None

    # ---- loss ------------------------------------------------------------------------------------------
    def _dice(self, logits: torch.Tensor, targets: torch.Tensor) -> _Loss:
        """DiceLoss (``loss.py:9-20``) on the fused kernel."""
        sums = self._model.dice_forward(logits, targets)
        return _Loss(self._model, sums, logits.numel())

    # ---- training (trainable_unet.py:91-369) -----------------------------------------------------------
    def fit(self, dataset, batch_size: int = 4, num_epochs: int = 10, learning_rate: float = 0.005,
            weight_decay: float = 0.0005, tensorboard: bool = False, validation_set=None,
            min_delta: Optional[float] = 0.01, steps_per_epoch: Optional[int] = None, use_evaluation_metric: bool = True,
            checkpoint_dir=None, save_best: bool = False, use_scheduler: bool = False, **kwargs):
        if self._model.split_levels:
            raise RuntimeError("dtype bf16x3 is an inference mode: train with dtype fp32 or bf16 and load the checkpoint")
        if save_best:
            if validation_set is None:
                raise TypeError("`validation_set` is required when `save_best` is True")
            if checkpoint_dir is None:
                raise TypeError("`checkpoint_dir` is required when `save_best` is True")
        optimizer = HipAdam(self._model, lr=learning_rate, weight_decay=weight_decay)
        scheduler = None
        if use_scheduler:
            LOGGER.info(f"Setting up exponential scheduler, starting with learning rate {learning_rate}")
            scheduler = ExponentialLR(optimizer, gamma=0.8)
        metrics = self.set_up_metrics(use_evaluation_metric)
        if isinstance(checkpoint_dir, str):
            checkpoint_dir = Path(checkpoint_dir)
        writer = None
        if tensorboard:
            if checkpoint_dir is None:
                raise TypeError("`checkpoint_dir` is required when `tensorboard` is True")
            tensorboard_dir = checkpoint_dir / "tensorboard"
            os.makedirs(tensorboard_dir, exist_ok=True)
            try:
                from torch.utils.tensorboard import SummaryWriter
                writer = SummaryWriter(log_dir=str(tensorboard_dir))
                LOGGER.info(f"Tensorboard logs are written to {tensorboard_dir}")
            except Exception:
                LOGGER.warning("tensorboard is not installed; no logs are written")
        if not steps_per_epoch:
            steps_per_epoch = max(1, len(dataset) // batch_size)
        steps_per_epoch_val = len(validation_set) // batch_size if validation_set else None
        prev_best = (1., 0)
        best_validation_loss = float("inf")
        self.make_resident(dataset, validation_set)
        # Small batches are bound by the host issuing ~330 launches per step: replay them from a HIP graph
        # (``hip_graph=True/False`` overrides; large batches gain nothing and would pin their activations).
        # Data parallel: when the process runs inside an initialised torch.distributed group of more than
        # one rank (train.py under torch.distributed.run), ``batch_size`` is the GLOBAL batch; every rank takes its
        # slice of each batch, the Dice sums and the gradient arena are all-reduced (``DataParallelTrainer``).
        # ``data_parallel=True`` forces that path in a group of one (tests), ``False`` switches it off.
        dp = kwargs.pop("data_parallel", None)
        group_size = self._group_size()
        if dp is None:
            dp = group_size > 1
        if dp and group_size < 1:
            raise RuntimeError("data_parallel=True needs an initialised torch.distributed process group")
        if dp and batch_size % group_size:
            raise ValueError(f"batch_size {batch_size} is not divisible by the {group_size} ranks of the process group")
        # ``sync_bn=True`` (data parallel only): BatchNorm statistics of the global batch instead of per-replica ones
        sync_bn = bool(kwargs.pop("sync_bn", False))
        if sync_bn and not dp:
            raise ValueError("sync_bn needs data_parallel training (an initialised process group)")
        self._dp = {"trainer": None, "sync_bn": sync_bn} if dp else None
        use_graph = kwargs.pop("hip_graph", None)
        if use_graph is None:
            use_graph = batch_size <= 64 and not dp
        if use_graph and dp:
            raise ValueError("hip_graph and data_parallel exclude each other (the all-reduce is not captured)")
        self._graph = {"enabled": bool(use_graph) and self._model.device.type == "cuda", "step": None, "warm": False,
                       "batch_size": batch_size}
        try:
            self._fit_epochs(dataset, validation_set, num_epochs, steps_per_epoch, steps_per_epoch_val, batch_size, optimizer,
                             scheduler, metrics, writer, save_best, checkpoint_dir, min_delta, prev_best, best_validation_loss)
        finally:
            self.release_resident()
            self._graph = {"enabled": False, "step": None, "warm": False}
            self._dp = None
            self._model.sync_bn = None
        if writer:
            writer.close()
        if save_best:
            self.load(checkpoint_dir / "best_model")
            if self._group_size() > 1:                 # every rank has read it before rank 0 removes it
                import torch.distributed as dist
                dist.barrier()
                if dist.get_rank() == 0:
                    shutil.rmtree(checkpoint_dir / "best_model")
            else:
                shutil.rmtree(checkpoint_dir / "best_model")
            LOGGER.info("Restored previous best model")

    def _fit_epochs(self, dataset, validation_set, num_epochs, steps_per_epoch, steps_per_epoch_val, batch_size, optimizer,
                    scheduler, metrics, writer, save_best, checkpoint_dir, min_delta, prev_best, best_validation_loss):
        """The epoch loop of ``fit`` (``trainable_unet.py:177-287``)."""
        for epoch in range(num_epochs):
            summary: Dict[str, Dict[str, float]] = defaultdict(dict)
            descr = f"Epoch {epoch + 1}/{num_epochs}"
            train_batches = islice(chunks(dataset, batch_size), steps_per_epoch)
            training_loss = self.epoch(batches=train_batches, steps_per_epoch=steps_per_epoch, description=descr,
                                       optimizer=optimizer, train=True, metrics=metrics)
            _log_metric("training_loss", training_loss, epoch)
            summary["Loss"]["training"] = training_loss
            for metric in metrics:
                result = metric.compute()
                summary[metric.__class__.__name__]["training"] = result
                LOGGER.info(f"{descr} - Training {metric.__class__.__name__.lower()}: {result:.6f}")
            LOGGER.info(f"{descr} - Training loss: {training_loss:.6f}")
            if validation_set:
                validation_metric = self.compute_metric(metrics[0], validation_set, batch_size) if metrics else float("nan")
                validation_loss = self.compute_validation_loss(validation_set, batch_size, descr, steps_per_epoch_val, metrics)
                if self._group_size() > 1:
                    # save_best (a barrier inside save()) and early stopping branch on this value: every rank takes
                    # rank 0's, so no rank can leave the loop or enter a collective alone
                    import torch.distributed as dist
                    box = [validation_loss, validation_metric]
                    dist.broadcast_object_list(box, src=0)
                    validation_loss, validation_metric = box
                LOGGER.info(f"{descr} - Validation metric ({metrics[0] if metrics else None}): {validation_metric:.6f} | "
                            f"Validation loss: {validation_loss:.6f}")
                _log_metric("validation_loss", validation_loss, epoch)
                _log_metric("validation_metric", validation_metric, epoch)
                summary["Loss"]["validation"] = validation_loss
                for metric in metrics:
                    result = metric.compute()
                    summary[metric.__class__.__name__]["validation"] = result
                    LOGGER.info(f"{descr} - Validation {metric.__class__.__name__.lower()}: {result:.6f}")
                if save_best and validation_loss < best_validation_loss:
                    best_validation_loss = validation_loss
                    self.save(checkpoint_dir / "best_model")
                    LOGGER.info(f"Saved best model with loss {best_validation_loss:.6f} at epoch {epoch + 1}.")
                if validation_loss < prev_best[0] - min_delta:
                    prev_best = (validation_loss, epoch)
                elif epoch - prev_best[1] > 5:
                    LOGGER.info(f"Early stopping reached at epoch {epoch + 1}.")
                    break
            if scheduler is not None:
                self.update_scheduler(descr, optimizer, scheduler)
            if writer and summary:
                for tag, scalars in summary.items():
                    writer.add_scalars(tag, scalars, global_step=epoch + 1)
                writer.flush()

    def set_up_metrics(self, use_evaluation_metric: bool) -> List[BinaryAccuracy]:
        return [BinaryAccuracy()] if use_evaluation_metric else []

    @staticmethod
    def update_scheduler(descr: str, optimizer: HipAdam, scheduler: ExponentialLR):
        scheduler.step()
        LOGGER.info(f"{descr}: scheduler decreased learning rate to {optimizer.param_groups[0]['lr']}")

    def epoch(self, batches: Iterator[Sequence], steps_per_epoch: int, description: str, optimizer: Optional[HipAdam] = None,
              train: bool = True, metrics: Sequence[BinaryAccuracy] = None) -> float:
        if _tqdm is not None:
            batches = _tqdm(batches, desc=description, total=steps_per_epoch)
        if train:
            self._model.train()
            optimizer.zero_grad()
        else:
            self._model.eval()
        for metric in metrics or []:
            metric.reset()
        epoch_loss = 0.
        pending = None       # the loss of step k is read back after step k+1 is queued: the GPU never waits for the host
        for step, batch in enumerate(batches):
            loss = None
            if train and self._dp is not None:
                loss = self._data_parallel_step(batch, metrics, optimizer)
            elif train:
                loss = self._graphed_train_step(batch, metrics, optimizer)
            if loss is None:
                loss = self.step(batch, metrics)
                if train:
                    loss.backward()
                    optimizer.step()
                    optimizer.zero_grad()
            if pending is not None:
                epoch_loss = (epoch_loss * (step - 1) + float(pending)) / step
                if _tqdm is not None and isinstance(batches, _tqdm):
                    batches.set_postfix({("training" if train else "validation") + "_loss": epoch_loss})
            pending, last = loss, step
        if pending is not None:
            epoch_loss = (epoch_loss * last + float(pending)) / (last + 1)
        if train and self._dp is not None:
            # BatchNorm running statistics are per replica: continue from rank 0's, so that validation losses -- and with
            # them early stopping and best-model decisions -- are identical on every rank
            from .distributed import broadcast
            broadcast(self._model.flat_stats, 0)
            self._model._param_epoch += 1
        return epoch_loss

    def _data_parallel_step(self, batch: Sequence, metrics, optimizer: HipAdam) -> Optional[_Loss]:
        """This rank's slice of a global batch through ``DataParallelTrainer.step`` (forward, all-reduced Dice sums,
        backward with bucketed gradient all-reduce, Adam).  A short last batch (validation-style) is not sliced."""
        import torch.distributed as dist
        from .distributed import DataParallelTrainer
        world, rank = dist.get_world_size(), dist.get_rank()
        if len(batch) % world:
            raise ValueError(f"a batch of {len(batch)} profiles cannot be split over {world} ranks")
        per = len(batch) // world
        local = batch[rank * per:(rank + 1) * per]
        cfg = self._dp
        if cfg["trainer"] is None or cfg["trainer"].optimizer is not optimizer:
            cfg["trainer"] = DataParallelTrainer(self._model, optimizer, sync_bn=cfg.get("sync_bn", False))
        inputs, y_true = self.get_inputs(local), self.get_targets(local)
        sums = cfg["trainer"].step(inputs, y_true)
        return _Loss(self._model, sums, inputs.numel() * world, metrics)

    def _graphed_train_step(self, batch: Sequence, metrics, optimizer: HipAdam) -> Optional[_Loss]:
        """step + backward + optimizer.step + zero_grad as one HIP-graph replay; None when this batch has to take the
        launch-by-launch path (graphs off, first step = warm-up of every lazily built workspace, odd batch size)."""
        cfg = self._graph
        if not cfg["enabled"] or len(batch) != cfg.get("batch_size"):
            return None
        if not cfg["warm"]:
            cfg["warm"] = True
            return None
        inputs, y_true = self.get_inputs(batch), self.get_targets(batch)
        if cfg["step"] is None or not cfg["step"].matches(inputs) or cfg["step"].optimizer is not optimizer:
            count = optimizer.step_count
            try:
                cfg["step"] = GraphedTrainStep(self._model, optimizer, inputs, y_true)
            except Exception as exc:      # capture refused: keep training launch by launch (still on the GPU)
                LOGGER.warning(f"HIP graph capture of the training step failed ({exc}); continuing without it")
                optimizer.step_count = count
                self._model._saved = None
                cfg["enabled"], cfg["step"] = False, None
                return None
        sums = cfg["step"](inputs, y_true)
        return _Loss(self._model, sums, inputs.numel(), metrics)

    def step(self, batch: Sequence, metrics: Optional[Sequence[BinaryAccuracy]]) -> _Loss:
        inputs = self.get_inputs(batch)
        y_true = self.get_targets(batch)
        logits = self._model(inputs)
        loss = self.loss_fn(logits, y_true)
        loss._metrics = list(metrics or [])      # resolved with the loss value (one sync, after the step is queued)
        return loss

    # ---- inference (trainable_unet.py:371-409) ---------------------------------------------------------
    def predict(self, image) -> Prediction:
        return self.predict_batch([image])[0]

    PREDICT_SUB_BATCH = 64     # profiles per pipeline stage of predict_batch (the kernels are at 95 % of their batch-256 rate)

    def predict_batch(self, batch: Sequence) -> List[Prediction]:
        """``trainable_unet.py:371-409`` as a pipeline over sub-batches: stage k's tensorisation, H->D copy, forward, on-device
        pixel counts, allele kernel and D->H copies (into pinned memory, asynchronous) are queued while the host is still
        turning stage k-1's small tables into ``Prediction`` / ``Marker`` objects; at most two stages are in flight."""
        self._model.eval()
        batch = list(batch)
        if self.allele_caller:
            LOGGER.info("Calling alleles from predicted segmentation...")
        predictions: List[Prediction] = []
        in_flight: List[tuple] = []
        for s in range(0, len(batch), self.PREDICT_SUB_BATCH):
            sub = batch[s:s + self.PREDICT_SUB_BATCH]
            if len(in_flight) >= 2:                   # at most two stages in flight: finish the oldest before queuing a third
                predictions += self._predict_finish(*in_flight.pop(0))
            in_flight.append((sub, self._predict_launch(sub)))
        while in_flight:
            predictions += self._predict_finish(*in_flight.pop(0))
        return predictions

    @staticmethod
    def _to_pinned(t: torch.Tensor) -> torch.Tensor:
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        return host

    def _predict_launch(self, sub: Sequence) -> dict:
        """Queue everything one sub-batch needs on the GPU; nothing here waits for the device."""
        inputs = self.get_inputs(sub)
        _, probs = self._model.infer(inputs, want_logits=False, want_probs=True)
        h = {"probs_host": self._to_pinned(probs.movedim(1, -1)), "counts_host": None, "alleles": None}    # (N,H,W,1)
        counts = self._pixel_counts_launch(sub, probs)
        if counts is not None:
            h["counts_host"] = self._to_pinned(counts)
        if self.allele_caller and hasattr(self.allele_caller, "launch_alleles_batch"):
            # device path: runs / nearest base pair / peak heights from the probability maps still in HBM
            h["alleles"] = self.allele_caller.launch_alleles_batch(sub, probs, inputs)
        h["keep"] = (inputs, probs, counts)          # alive until the copies have been waited for
        h["event"] = torch.cuda.Event()
        h["event"].record()
        return h

    def _predict_finish(self, sub: Sequence, h: dict) -> List[Prediction]:
        h["event"].synchronize()
        host = h["probs_host"].numpy().copy()        # own memory: the pinned block goes back to the allocator
        predictions = [Prediction(image=host[i], meta={"original_image_path": getattr(image, "path", None)})
                       for i, image in enumerate(sub)]
        if h["counts_host"] is not None:     # annotated profiles: the pixel metrics' confusion counts from the same pass
            counts = h["counts_host"].numpy()
            for i, pred in enumerate(predictions):
                if counts[i, 6] == 0:                 # a 0/1 mask
                    pred._pixel_counts = (id(sub[i]), 0.5, tuple(int(c) for c in counts[i, :6]))
        if self.allele_caller:
            if h["alleles"] is not None:
                predictions = self.allele_caller.finish_alleles_batch(sub, predictions, h["alleles"])
            else:
                predictions = [self.allele_caller.call_alleles(im, pred) for im, pred in zip(sub, predictions)]
        h["keep"] = None
        return predictions

    def _pixel_counts_launch(self, batch: Sequence, probs: torch.Tensor) -> Optional[torch.Tensor]:
        """On-device metrics: per profile {tp, fp, fn} at ``p >= 0.5`` and at ``p > 0.5`` against the
        annotation, computed from the probability maps while they are in HBM (``dn_pixel_confusion``).  Only for
        annotations that are 0/1 masks of the output's shape; for the others ``dnanet_amd.metrics`` makes its boolean
        passes on the host, like the reference.  Returns the (N,7) int32 device tensor, or None."""
        try:
            truth = np.stack([np.asarray(im.annotation.image) for im in batch])              # (N,H,W,1)
        except (AttributeError, ValueError, TypeError):
            return None
        if truth.size != probs.numel() or truth.dtype.kind not in "iubf":
            return None
        # int8 with everything that is not exactly 0 / 1 mapped to "other" (the kernel counts those per profile)
        if truth.dtype.kind == "f":
            truth = np.where(truth == 1, 1, np.where(truth == 0, 0, 2)).astype(np.int8)
        elif truth.dtype.itemsize > 1:
            truth = np.clip(truth, -1, 2).astype(np.int8)
        else:
            truth = truth.astype(np.int8, copy=False)            # uint8 values >= 128 wrap to negatives: still "other"
        truth = np.ascontiguousarray(np.transpose(truth, (0, 3, 1, 2)))
        truth_dev = torch.from_numpy(truth).to(probs.device)     # pageable temporary: a synchronous copy (it is 20 kB per profile)
        counts = torch.empty((len(batch), 7), dtype=torch.int32, device=probs.device)
        hip.check(hip.lib.dn_pixel_confusion(hip.ptr(probs), hip.ptr(truth_dev), len(batch), probs[0].numel(), 0.5, 0.5,
                                             hip.ptr(counts), hip.stream_ptr()))
        return counts

    def _pixel_counts(self, batch: Sequence, probs: torch.Tensor) -> Optional[np.ndarray]:
        counts = self._pixel_counts_launch(batch, probs)
        return None if counts is None else counts.cpu().numpy()

    def predict_raw(self, images: Sequence) -> torch.Tensor:
        return self._model(self.get_inputs(images))

    @staticmethod
    def sigmoid(logits: torch.Tensor) -> torch.Tensor:
        return torch.sigmoid(logits)

    def update_metric(self, metric: BinaryAccuracy, logits: torch.Tensor, y_true: torch.Tensor):
        metric.update(torch.flatten(self.sigmoid(logits)), torch.flatten(y_true))

    def compute_metric(self, metric: BinaryAccuracy, dataset, batch_size: int) -> float:
        """As the reference (``trainable_unet.py:417-432``) this runs in whatever mode the model is in --
        train mode right after a training epoch, so BatchNorm statistics see these batches too."""
        metric.reset()
        for batch in chunks(dataset, batch_size):
            y_true = self.get_targets(batch)
            logits = self._model(self.get_inputs(batch))
            sums = self._model.dice_forward(logits, y_true)
            metric.update_counts(float(sums[3]), logits.numel())
        self._model._saved = None
        return metric.compute()

    def compute_validation_loss(self, validation_set, batch_size: int, description: str, steps_per_epoch: int,
                                metrics: Sequence[BinaryAccuracy]) -> float:
        return self.epoch(chunks(validation_set, batch_size), steps_per_epoch=steps_per_epoch, description=description,
                          train=False, metrics=metrics)

    # ---- checkpoints (trainable_unet.py:447-470) -------------------------------------------------------
    def save(self, model_dir):
        """Inside a process group only rank 0 writes (the replicas hold the same parameters); everybody waits for it."""
        ranks = self._group_size()
        if ranks > 1:
            import torch.distributed as dist
            if dist.get_rank() == 0:
                self._save(model_dir)
            dist.barrier()
        else:
            self._save(model_dir)

    def _save(self, model_dir):
        os.makedirs(model_dir, exist_ok=True)
        state = {k: v.detach().cpu().clone() for k, v in self._model.state_dict().items()}
        torch.save(state, self._get_weights_path(model_dir))

    def load(self, model_dir):
        state = torch.load(self._get_weights_path(model_dir), map_location="cpu", weights_only=True)
        self._model.load_state_dict(state)

    def _get_weights_path(self, model_dir) -> str:
        return os.path.join(model_dir, f"{self.__class__.__name__}_checkpoint.pt")
