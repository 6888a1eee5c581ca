"""bench.py's launch contract on the CPU: ``--gpus N`` with no launcher starts N ranks itself, a launcher whose world
size differs from ``--gpus`` is refused, and the N=1 form stays a single process.  ``--dry-run`` replaces the GPU step
by a stub (gloo, no device), so this runs without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _line(proc):
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, (proc.stdout, proc.stderr[-2000:])
    return json.loads(lines[0])


def test_gpus_2_without_launcher_spawns_two_ranks():
    proc = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = _line(proc)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["world_size_seen_by_rank0"] == 2
    assert line["steps"] == 3 and line["warmup"] == 1


def test_single_gpu_form_is_one_process():
    proc = _run(["--steps", "2", "--warmup", "0", "--dry-run"])
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = _line(proc)
    assert line["n_gpus"] == 1 and line["ranks_seen"] == [0]


def test_world_size_mismatch_is_refused():
    # a launcher that started ONE rank for a command that says --gpus 2: the run must not be labelled n_gpus 1 or 2
    proc = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert proc.returncode != 0
    assert "refusing" in proc.stderr and '{"metric"' not in proc.stdout


def test_gpus_4_and_8_without_launcher_spawn_that_many_ranks():
    """The sizes the driver's scaling run uses (N = 1, 2, 4, 8): rendezvous, rank set and the max-over-ranks timing."""
    for n in (4, 8):
        proc = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--dry-run"], timeout=600)
        assert proc.returncode == 0, proc.stderr[-2000:]
        line = _line(proc)
        assert line["n_gpus"] == n and line["ranks_seen"] == list(range(n)) and line["world_size_seen_by_rank0"] == n


def test_launcher_started_ranks_are_accepted_when_world_size_matches():
    """The driver's own form: torch.distributed.run starts the ranks, bench.py reads RANK / WORLD_SIZE from the environment."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                           "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-run"],
                          capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = _line(proc)
    assert line["n_gpus"] == 4 and line["ranks_seen"] == [0, 1, 2, 3]
