"""bench.py's N > 1 path on real hardware, as far as one GPU allows: two ranks started by bench.py itself share the box's GPU
(`--backend gloo`: RCCL refuses two ranks on one device; the collectives are staged through the host).  Everything but RCCL
itself runs: the rank spawn, rendezvous, the data-parallel trainer with the two-stream backward and its bucket hooks, the
strong-scaling leg, the max-over-ranks timing and the one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_sharing_the_gpu(gpu_device):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "8",
                           "--steps", "2", "--warmup", "1", "--global-batch", "32"], capture_output=True, text=True, timeout=170,
                          env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size_seen_by_rank0"] == 2 and d["config"]["global_batch"] == 16
    assert d["value"] > 0 and d["roofline"]["launches"] > 0
    tr = d["train_step"]
    assert tr["global_batch"] == 16 and tr["comm"]["world_size_seen_by_rank0"] == 2 and tr["comm"]["gradient_buckets_per_step"] >= 2
    assert tr["comm"]["bytes_all_reduced_per_step"] == 12123748 * 4 or tr["comm"]["bytes_all_reduced_per_step"] >= 12123745 * 4
    assert 0.0 < tr["loss"] < 1.0
    g = d["train_step_global"]
    assert g["per_gpu_batch"] == 16 and g["global_batch"] == 32 and g["scaling"] == "strong" and 0.0 < g["loss"] < 1.0


def test_bench_rccl_code_path_with_one_rank(gpu_device):
    """`--force-distributed`: the process group (backend nccl = RCCL), device barrier, max-reduce of the timing, the data-parallel
    trainer with its bucketed all-reduce on a side stream and the strong-scaling leg, all with one rank: every call of the
    N > 1 run except the other ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-distributed", "--batch", "16",
                           "--steps", "2", "--warmup", "1", "--global-batch", "32"], capture_output=True, text=True, timeout=170,
                          env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    d = json.loads([l for l in proc.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl"
    tr = d["train_step"]
    assert tr["comm"]["backend"] == "nccl" and tr["comm"]["world_size_seen_by_rank0"] == 1 and tr["comm"]["gradient_buckets_per_step"] >= 2
    assert "RCCL" in tr["parallelism"] and 0.0 < tr["loss"] < 1.0
    assert d["train_step_global"]["per_gpu_batch"] == 32 and 0.0 < d["train_step_global"]["loss"] < 1.0
