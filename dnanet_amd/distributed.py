"""Multi-GPU execution of the hot path: one process per GPU, ``torch.distributed`` ("nccl" = RCCL over xGMI).

The reference is single-device; the path shards in exactly two ways:

* inference: profiles are independent in eval mode, so a batch is split into contiguous shards, one per
  rank, with NO data-path collective (``shard_range``; results are gathered by the caller if it wants them);
* training: data parallel.  The Dice loss is a ratio of whole-batch sums (``loss.py:16-18``), so the three
  sums (+ the accuracy count) are all-reduced BEFORE the backward pass and every rank back-propagates the
  *global* loss through its local shard; parameter gradients are then summed (not averaged) across ranks.
  The flat gradient arena is reduced in buckets, one per top-level module in the order backward finishes
  them (head, decoders.3 .. decoders.0, bottleneck, encoders.3 .. encoders.0), each launched on a side
  stream as soon as its segment's gradients are complete, overlapping the rest of backward.
  xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 48.5 MB arena costs ~0.6 ms as a ring
  all-reduce against tens of ms of compute per step, so a handful of large buckets is the right grain.
  BatchNorm uses per-replica batch statistics by default (the DDP convention; rank 0's running statistics are saved).
  ``sync_bn=True`` all-reduces the per-channel statistics instead -- 2*C doubles per BatchNorm in the forward pass
  ([sum y, sum y^2]) and in the backward pass ([sum dz, sum dz*xhat]) -- which makes R ranks x N/R profiles compute what
  one device computes at batch N (the only oracle a data-parallel run has): 36 small all-reduces per step.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _host_staged(tensor: torch.Tensor, group) -> bool:
    """A gloo group handed a device tensor (tests: two ranks sharing one GPU; RCCL refuses that): the collective runs on a
    host copy.  gloo's own device-tensor path (pinned bounce buffers on pool streams, a thread pool) was seen to hang
    after a few dozen collectives with two processes on one MI355X; RCCL groups never take this route."""
    return tensor.is_cuda and dist.get_backend(group) != "nccl"


def all_reduce_sum(tensor: torch.Tensor, group=None):
    """In-place sum over the group (synchronous)."""
    if _host_staged(tensor, group):
        host = tensor.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        tensor.copy_(host)
    else:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)


def broadcast(tensor: torch.Tensor, src: int = 0, group=None):
    if _host_staged(tensor, group):
        host = tensor.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        tensor.copy_(host)
    else:
        dist.broadcast(tensor, src=src, group=group)


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, stop) of ``n`` items for ``rank``; sizes differ by at most one."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def predict_batch_sharded(model, images: Sequence, group=None, dst: int = 0, gather_images: bool = True):
    """``model.predict_batch`` over one node's GPUs (inference): every rank predicts its contiguous shard of
    ``images`` -- allele calling included, so the product of the path (called alleles, pixel counts) is a few hundred
    bytes per profile -- and rank ``dst`` receives the complete list in the order of ``images`` (the other ranks get
    None).  No collective on the data path; the ordered gather is one ``gather_object`` (use a gloo group for it: the
    payload is host objects).  A shard's probability maps travel as ONE stacked float32 array (a single buffer, not a
    pickled object per profile); ``gather_images=False`` leaves them on their ranks altogether (``Prediction.image`` is
    then None on ``dst``) for callers that only want the calls and metrics: at 256 profiles per GPU the maps are 20 MB per
    rank, the rest a few hundred kB."""
    import numpy as np
    from .prediction import Prediction
    images = list(images)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    start, stop = shard_range(len(images), rank, world)
    local = list(model.predict_batch(images[start:stop])) if stop > start else []
    maps = None
    if gather_images and local and all(isinstance(p.image, np.ndarray) and p.image.shape == local[0].image.shape for p in local):
        maps = np.stack([p.image for p in local])
    payload = {"maps": maps,
               "items": [(None if maps is not None or not gather_images else p.image, p.meta, getattr(p, "classification", None),
                          tuple(getattr(p, "_pixel_counts", ())[1:]) or None) for p in local]}
    parts = [None] * world if rank == dst else None
    dist.gather_object(payload, parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for part in parts:
        for i, (image, meta, classification, counts) in enumerate(part["items"]):
            if part["maps"] is not None:
                image = part["maps"][i]
            pred = Prediction(image=image, meta=meta)
            if classification is not None:
                pred.classification = classification
            if counts:
                pred._pixel_counts = (id(images[len(out)]),) + tuple(counts)   # device pixel counts are tagged with the image object
            out.append(pred)
    return out


def module_segments(offsets: Dict[str, tuple], depth: int, total: Optional[int] = None) -> List[Tuple[str, int, int]]:
    """Arena ranges [(segment name, start, stop)] of the trainable parameters per top-level module,
    in the order the backward pass completes them.  Ranges tile the arena (alignment padding included)."""
    spans: Dict[str, List[int]] = {}
    for key, (kind, off, numel, _shape) in offsets.items():
        if kind != "param":
            continue
        top = key.split(".")[0]
        name = top if top in ("bottleneck", "outputs") else ".".join(key.split(".")[:2])
        lo, hi = spans.setdefault(name, [off, off + numel])
        spans[name] = [min(lo, off), max(hi, off + numel)]
    starts = sorted(lo for lo, _ in spans.values())
    end_of = {lo: (starts[i + 1] if i + 1 < len(starts) else max(total or 0, max(hi for _, hi in spans.values())))
              for i, lo in enumerate(starts)}
    spans = {n: [lo, end_of[lo]] for n, (lo, _hi) in spans.items()}
    order = ["outputs"] + [f"decoders.{i}" for i in reversed(range(depth))] + ["bottleneck"] + \
            [f"encoders.{i}" for i in reversed(range(depth))]
    return [(("head" if n == "outputs" else n), spans[n][0], spans[n][1]) for n in order]


class ArenaAllReducer:
    """Bucketed sum-all-reduce of a flat gradient arena, segment by segment (device agnostic: the same
    code runs under gloo on CPU tensors in the tests and under RCCL on the GPU)."""

    def __init__(self, flat_grads: torch.Tensor, segments: Sequence[Tuple[str, int, int]], group=None,
                 min_bucket_elems: int = 2 << 20):
        self.flat, self.group = flat_grads, group
        self.segments = list(segments)
        self.min_bucket = min_bucket_elems
        self._pending: List[Tuple[int, int]] = []
        self._deferred: List[Tuple[int, int]] = []
        self._works = []
        # Overlap with backward (side stream, async launches) only under RCCL, whose collectives all run in issue order on
        # the group's own stream; a gloo group given device tensors reduces the buckets one after the other in finish(),
        # through the host (all_reduce_sum).
        self.overlap = flat_grads.is_cuda and dist.get_backend(group) == "nccl"
        self.stream = torch.cuda.Stream(device=flat_grads.device) if self.overlap else None
        self._by_name = {n: (a, b) for n, a, b in self.segments}
        self.launched: List[Tuple[int, int]] = []
        self.last_buckets: List[Tuple[int, int]] = []     # the (start, stop) ranges the latest step reduced

    def _launch(self, lo: int, hi: int):
        view = self.flat[lo:hi]
        if self.flat.is_cuda and not self.overlap:
            self._deferred.append((lo, hi))
            self.launched.append((lo, hi))
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched.append((lo, hi))

    def segment_done(self, name: str):
        """Called by backward when all gradients of ``name`` are final (HipUNet.backward segment_hook)."""
        lo, hi = self._by_name[name]
        self._pending.append((lo, hi))
        lo_all, hi_all = min(a for a, _ in self._pending), max(b for _, b in self._pending)
        if hi_all - lo_all >= self.min_bucket:
            self._launch(lo_all, hi_all)
            self._pending = []

    def finish(self):
        """Flush the last bucket and make the current stream wait for every reduction."""
        if self._pending:
            self._launch(min(a for a, _ in self._pending), max(b for _, b in self._pending))
            self._pending = []
        for lo, hi in self._deferred:
            all_reduce_sum(self.flat[lo:hi], self.group)
        self._deferred = []
        for w in self._works:
            w.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._works, done = [], self.launched
        self.launched = []
        self.last_buckets = done
        return done


def all_reduce_dice_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    """sums = [sum p*t, sum p, sum t, #correct, local loss, ...] -> the same with global sums and the global
    Dice loss 1 - (2I+1)/(P+T+1) in slot 4."""
    out = sums.clone()
    all_reduce_sum(out[:4], group)
    out[4] = 1.0 - (2.0 * out[0] + 1.0) / (out[1] + out[2] + 1.0)
    return out


class DataParallelTrainer:
    """One data-parallel training step of ``HipUNet`` on this rank's shard of the global batch."""

    def __init__(self, model, optimizer, group=None, min_bucket_elems: int = 2 << 20, sync_bn: bool = False):
        self.model, self.optimizer, self.group = model, optimizer, group
        self.sync_bn = bool(sync_bn)
        model.sync_bn = (lambda t: all_reduce_sum(t, group), dist.get_world_size(group)) if sync_bn else None
        self.reducer = ArenaAllReducer(model.flat_grads, module_segments(model.offsets, model.depth, model.flat_grads.numel()), group,
                                       min_bucket_elems)
        # identical initial weights on every rank
        broadcast(model.flat_params, 0, group)
        broadcast(model.flat_stats, 0, group)
        model._param_epoch += 1

    def step(self, x_local: torch.Tensor, t_local: torch.Tensor) -> torch.Tensor:
        m = self.model
        logits = m.forward_train(x_local)
        sums = all_reduce_dice_sums(m.dice_forward(logits, t_local), self.group)
        m.backward(sums=sums, segment_hook=self.reducer.segment_done)
        self.reducer.finish()
        self.optimizer.step()
        self.optimizer.zero_grad()
        return sums
