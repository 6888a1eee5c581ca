"""train.py and evaluate.py under torch.distributed.run with TWO ranks on the box's one GPU (gloo: RCCL wants a device per
rank; device tensors are staged through the host): the data-parallel fit() path end to end -- rank-0 seed for the unseeded
validation split, per-rank batch slices, Dice sums and gradient arena all-reduced, BatchNorm statistics and validation loss
broadcast, rank 0 alone writing -- and the sharded evaluation, which must reproduce the one-process metrics exactly."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(script, args, nproc, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, script)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)


def test_train_and_evaluate_with_two_ranks(gpu_device, tmp_path):
    cfg = tmp_path / "cfg"
    cfg.mkdir()
    (cfg / "data.yaml").write_text(yaml.safe_dump({"dataset": {"name": "synthetic", "n": 24, "rows": 5, "width": 256, "seed": 5}}))
    (cfg / "model.yaml").write_text(yaml.safe_dump({"model": {"name": "unet", "depth": 2, "kernel_size": [3, 5], "num_filters": 8,
                                                              "apply_allele_caller": False, "device": "cuda"}}))
    (cfg / "train.yaml").write_text(yaml.safe_dump({"training": {"batch_size": 4, "num_epochs": 3, "learning_rate": 2e-3,
                                                                 "tensorboard": False, "use_evaluation_metric": True,
                                                                 "use_scheduler": True, "save_best": True}}))
    out = tmp_path / "run"
    # -v 0.25 without -rs: the unseeded validation split that every rank has to agree on
    proc = _launch("train.py", ["-d", str(cfg / "data.yaml"), "-m", str(cfg / "model.yaml"), "-t", str(cfg / "train.yaml"),
                                "-o", str(out), "-v", "0.25"], 2, {"DNANET_DIST_BACKEND": "gloo"})
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert (out / "DNANet_UNet_checkpoint.pt").is_file() and (out / "config.yaml").is_file(), \
        (sorted(os.listdir(out)) if out.is_dir() else "no output dir", proc.stdout[-1500:], proc.stderr[-3000:])
    log = (out / "log_training.txt").read_text()
    assert log.count("Starting training...") == 1                      # rank 0 alone owns the log file
    losses = [float(l.rsplit(" ", 1)[-1]) for l in log.splitlines() if " - Training loss: " in l]
    # sharded evaluation == one-process evaluation on the same checkpoint
    res = {}
    for nproc in (1, 2):
        ev = tmp_path / f"eval{nproc}"
        args = ["-d", str(cfg / "data.yaml"), "-m", str(cfg / "model.yaml"), "-e", os.path.join(ROOT, "config/evaluation/segmentation.yaml"),
                "-o", str(ev), "-c", str(out), "-p"]
        proc = _launch("evaluate.py", args, nproc)
        assert proc.returncode == 0, proc.stderr[-3000:]
        res[nproc] = json.load(open(ev / "metrics.txt"))
        preds = json.load(open(ev / "predictions.json"))
        assert len(preds) == 24
        res[f"p{nproc}"] = np.array([p["image"] for p in preds], dtype=np.float32)
    assert res[1] == res[2]
    assert np.array_equal(res["p1"], res["p2"])
    assert all(np.isfinite(v) for v in res[1].values())
    assert len(losses) == 0 or losses[-1] <= losses[0] + 1e-6
