"""The N>1 logic under gloo, world_size 2, on CPU: shard ranges, segment order / bucketing of the gradient
arena, sum-all-reduce, global Dice sums."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dnanet_amd.distributed import ArenaAllReducer, all_reduce_dice_sums, module_segments, shard_range
from dnanet_amd.params import ParameterTree


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 256, 348):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_segments_cover_arena_in_backward_order():
    tree = ParameterTree(4, (3, 5), 32, "cpu")
    segs = module_segments(tree.offsets, 4, tree.flat_params.numel())
    assert [s[0] for s in segs] == ["head", "decoders.3", "decoders.2", "decoders.1", "decoders.0", "bottleneck",
                                    "encoders.3", "encoders.2", "encoders.1", "encoders.0"]
    spans = sorted((a, b) for _, a, b in segs)
    assert spans[0][0] == 0 and spans[-1][1] == tree.flat_params.numel()
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))     # contiguous, disjoint
    big = {n: b - a for n, a, b in segs}
    assert (big["bottleneck"] + big["decoders.0"]) / tree.n_param > 0.7            # ~75 % of the bytes


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tree = ParameterTree(2, (3, 5), 8, "cpu")
        segs = module_segments(tree.offsets, 2, tree.flat_grads.numel())
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(tree.flat_grads.numel(), generator=g)
        tree.flat_grads.copy_(local)
        red = ArenaAllReducer(tree.flat_grads, segs, min_bucket_elems=20000)
        for name, _, _ in segs:                       # the order backward reports segments
            red.segment_done(name)
        launched = red.finish()
        other = torch.randn(tree.flat_grads.numel(), generator=torch.Generator().manual_seed(100 + (1 - rank)))
        ok_sum = torch.allclose(tree.flat_grads, local + other)
        covered = sorted(launched)
        ok_cover = covered[0][0] == 0 and covered[-1][1] == tree.flat_grads.numel() and \
            all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        sums = torch.tensor([1.0 + rank, 10.0, 3.0, 50.0, 0.0, 0, 0, 0])
        gs = all_reduce_dice_sums(sums)
        ok_dice = torch.allclose(gs[:4], torch.tensor([3.0, 20.0, 6.0, 100.0])) and \
            abs(float(gs[4]) - (1 - (2 * 3 + 1) / (20 + 6 + 1))) < 1e-6
        q.put((rank, ok_sum, ok_cover, len(launched), ok_dice))
    finally:
        dist.destroy_process_group()


def test_bucketed_all_reduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_sum, ok_cover, nb, ok_dice in res:
        assert ok_sum and ok_cover and ok_dice, (rank, ok_sum, ok_cover, ok_dice)
        assert 1 < nb < 6          # a handful of buckets, not one per tensor


def _sharded_predict_worker(rank, world, port, out_path):
    import os
    import numpy as np
    import torch.distributed as dist
    from dnanet_amd.distributed import predict_batch_sharded
    from dnanet_amd.prediction import Prediction
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Image:
        def __init__(self, i):
            self.path, self.value = f"p{i}", i

    class Stub:                                    # stands in for DNANet_UNet.predict_batch (which needs the GPU)
        def predict_batch(self, batch):
            preds = [Prediction(image=np.full((1, 2, 1), im.value, np.float32), meta={"original_image_path": im.path,
                                                                                        "rank": rank}) for im in batch]
            for im, p in zip(batch, preds):
                p._pixel_counts = (id(im), 0.5, (im.value, 0, 0, 0, 0, 0))
            return preds

    images = [Image(i) for i in range(7)]
    got = predict_batch_sharded(Stub(), images)
    if rank == 0:
        assert [p.meta["original_image_path"] for p in got] == [f"p{i}" for i in range(7)]
        assert [p.meta["rank"] for p in got] == [0, 0, 0, 0, 1, 1, 1]
        assert all(float(p.image[0, 0, 0]) == i for i, p in enumerate(got))
        assert all(p._pixel_counts[0] == id(im) and p._pixel_counts[2][0] == i for i, (im, p) in enumerate(zip(images, got)))
        open(out_path, "w").write("ok")
    else:
        assert got is None
    lean = predict_batch_sharded(Stub(), images, gather_images=False)    # calls / counts only: the maps stay on their ranks
    if rank == 0:
        assert all(p.image is None for p in lean) and [p.meta["original_image_path"] for p in lean] == [f"p{i}" for i in range(7)]
        assert all(p._pixel_counts[2][0] == i for i, p in enumerate(lean))
    empty = predict_batch_sharded(Stub(), images[:1])          # fewer images than ranks: rank 1's shard is empty
    assert (len(empty) == 1) if rank == 0 else empty is None
    dist.destroy_process_group()


def test_predict_batch_sharded_gloo_world2(tmp_path):
    """Inference shards by profile: contiguous shards, ordered gather on rank 0, no other exchange."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "ok.txt"
    mp.spawn(_sharded_predict_worker, args=(2, port, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _split_worker(rank, world, port, out_dir):
    import os
    import random
    import torch.distributed as dist
    from dnanet_amd.cli import split_validation
    from dnanet_amd.data import ArrayDataset
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    random.seed(1000 + rank)                       # the ranks' own generators disagree, like processes seeded from entropy
    data = ArrayDataset(list(range(40)))
    train, val = split_validation(data, "0.25")    # train.py's `-v 0.25`: an unseeded split
    part, _ = data.split(0.5)                      # `-s 0.5` without `-rs`
    shuffled = list(ArrayDataset(list(range(10)), shuffle=True))
    folds = data.split_k_fold(4)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(repr((list(train), list(val), list(part), shuffled, [list(t) for _, t in folds])))
    dist.destroy_process_group()


def test_unseeded_splits_agree_across_ranks_gloo_world2(tmp_path):
    """Every rank of a data-parallel job builds its own dataset objects: an unseeded split / shuffle must still be the
    same on all of them (rank 0 draws the seed), or the ranks train on different sets and fit()'s collectives hang."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_split_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (tmp_path / "rank0.txt").read_text(), (tmp_path / "rank1.txt").read_text()
    assert a == b
    train, val, part, shuffled, folds = eval(a)
    assert len(train) == 30 and len(val) == 10 and sorted(train + val) == list(range(40)) and len(part) == 20
    assert sorted(shuffled) == list(range(10))


def test_unseeded_split_in_multirank_env_without_group_is_refused(monkeypatch):
    from dnanet_amd.data import ArrayDataset
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError):
        ArrayDataset(list(range(10))).split(0.5)
    assert len(ArrayDataset(list(range(10))).split(0.5, seed=3)[0]) == 5      # a seeded split needs no group


def _evaluate_shuffled_worker(rank, world, port, out_dir):
    """evaluate.py on a `shuffle: true` dataset inside a 2-rank job: the metrics pass on rank 0 iterates the
    dataset on its own -- no collective may hide in __iter__, and predictions must stay paired with their images."""
    import os
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import evaluate
    from dnanet_amd.prediction import Prediction

    class Oracle:                                  # stands in for DNANet_UNet (which needs the GPU): predicts the annotation
        def predict_batch(self, batch):
            return [Prediction(image=np.asarray(im.annotation.image, np.float32), meta={"original_image_path": im.path})
                    for im in batch]

        def load(self, _):
            pass

    evaluate.load_model = lambda cfg: Oracle()
    cfg = os.path.join(out_dir, "data.yaml")
    if rank == 0:
        with open(cfg, "w") as f:
            f.write("dataset:\n  name: synthetic\n  n: 9\n  rows: 2\n  width: 256\n  seed: 5\n  shuffle: true\n")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    res = evaluate.run(cfg, os.path.join(root, "config/models/unet.yaml"), os.path.join(root, "config/evaluation/segmentation.yaml"),
                       output_dir=os.path.join(out_dir, f"out{rank}"), save_preds=False)
    if rank == 0:
        # paired correctly, a perfect "model" scores 1.0 on every metric; a re-shuffled second iteration would not
        assert res and all(abs(v - 1.0) < 1e-9 for v in res.values()), res
        open(os.path.join(out_dir, "ok.txt"), "w").write("ok")
    else:
        assert res == {}
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_shuffled_dataset_gloo_world2(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_evaluate_shuffled_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok.txt").read_text() == "ok"


def _shuffled_iter_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    from dnanet_amd.data import ArrayDataset
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = ArrayDataset(list(range(12)), shuffle=True)
    epochs = [list(ds) for _ in range(2)]                         # both ranks: two "epochs" in the same order
    extra = [list(ds) for _ in range(3)] if rank == 0 else []     # rank 0 alone goes on iterating: must not wait for anybody
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write(repr(epochs))
    assert all(sorted(e) == list(range(12)) for e in epochs + extra)
    dist.barrier()
    dist.destroy_process_group()


def test_shuffled_iteration_is_rank_local_gloo_world2(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_shuffled_iter_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = eval((tmp_path / "r0.txt").read_text()), eval((tmp_path / "r1.txt").read_text())
    assert a == b and a[0] != a[1]                                # same order on both ranks, a new one per epoch


def _shipped_arena_worker(rank, world, port, out_dir):
    """ArenaAllReducer over the SHIPPED net's gradient arena (12.1 M floats, segments from 33 floats (head) to 5.9 M
    (bottleneck)) with the production bucket floor, four ranks."""
    import json
    import os
    import torch
    import torch.distributed as dist
    from dnanet_amd.distributed import ArenaAllReducer, module_segments
    from dnanet_amd.params import ParameterTree
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tree = ParameterTree(4, (3, 5), 32, "cpu")
    segs = module_segments(tree.offsets, 4, tree.flat_grads.numel())
    tree.flat_grads.fill_(float(rank + 1))
    red = ArenaAllReducer(tree.flat_grads, segs)                 # min_bucket_elems: the default DataParallelTrainer uses
    order = []
    for name, _, _ in segs:                                      # backward's order: head, decoders 3..0, bottleneck, encoders 3..0
        red.segment_done(name)
        order.append((name, list(red.launched)))
    buckets = red.finish()
    ok = bool((tree.flat_grads == float(sum(range(1, world + 1)))).all())
    if rank == 0:
        with open(os.path.join(out_dir, "buckets.json"), "w") as f:
            json.dump({"buckets": buckets, "ok": ok, "segments": [(n, a, b) for n, a, b in segs], "order": order}, f)
    assert ok
    dist.destroy_process_group()


def test_arena_all_reduce_shipped_arena_gloo_world4(tmp_path):
    """3-4 buckets in reverse (backward) order; every element reduced exactly once; uneven segments."""
    import json
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_shipped_arena_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    rec = json.loads((tmp_path / "buckets.json").read_text())
    buckets, segs = rec["buckets"], rec["segments"]
    assert rec["ok"] and 3 <= len(buckets) <= 4, buckets
    total = max(b for _, _, b in segs)
    spans = sorted(map(tuple, buckets))
    assert spans[0][0] == 0 and spans[-1][1] == total and all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    assert all(b - a >= 2 << 20 for a, b in buckets[:-1])        # only the last flush may be smaller than the floor
    # launch order follows backward: the first bucket holds the head and the decoders' end of the arena, the last one encoders.0
    by_name = {n: (a, b) for n, a, b in segs}
    assert buckets[0][0] <= by_name["head"][0] < buckets[0][1]
    assert buckets[-1][0] <= by_name["encoders.0"][0] < buckets[-1][1]
    # a bucket is launched as soon as its segments are complete, not at the end: by the time the bottleneck is reported at
    # least one reduction is already in flight
    launched_at = {name: len(l) for name, l in rec["order"]}
    assert launched_at["bottleneck"] >= 1 and launched_at["encoders.0"] <= len(buckets)
