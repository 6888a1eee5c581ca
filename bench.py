#!/usr/bin/env python3
"""Headline benchmark of the DNANet U-Net hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one eval-mode pass of the shipped U-Net (depth 4, kernel (3,5), 32 filters,
config/models/unet.yaml) over one batch of 256 synthetic 6-dye x 4096-scan electropherograms already
resident in HBM (BASELINE.json configs[1]).  For N > 1 the driver launches one rank per GPU
(torch.distributed.run); inference shards by profile, so every rank runs its own batch (weak
scaling, no data-path collective) and ``value`` is the whole-job profiles/s.

The JSON line also carries
  * ``roofline``: the dominant kernel (the bf16 3x5 implicit-GEMM convolution) timed live with HIP
    events on the launch stream during the timed steps: algorithmic FLOP / launch duration vs the
    2.5 PFLOP/s dense bf16 MFMA peak (MI355X_MICROARCH.md);
  * ``cpu_baseline``: the CPU oracle (a torch fp32 port of the reference path) timed on this box's
    host cores on a bounded sample (rank 0, N=1 only);
  * ``train_step``: fwd + Dice + bwd + Adam profiles/s once the training path is available.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP_H5 = 58_531_512_320          # SURVEY 8d, per profile, H=5
BF16_PEAK_TFLOPS = 2500.0             # dense bf16 MFMA peak, MI355X_MICROARCH.md
F32_PEAK_TFLOPS = 157.3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="profiles per GPU per step")
    ap.add_argument("--rows", type=int, default=6, help="dye rows (6 = BASELINE.json wording, 5 = reference HIDImage)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    return ap.parse_args()


def cpu_baseline(rows: int, seconds_budget: float = 20.0):
    """The oracle (oracle/unet_oracle.py, a port of the reference path) on the host cores."""
    from oracle import unet_oracle as O
    from dnanet_amd.synthetic import make_batch
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = O.init_state_dict(4, (3, 5), 32, seed=0, randomize_bn_stats=True)
    n = 8
    x = torch.from_numpy(make_batch(n, rows, 4096, seed=1234)[0])
    with torch.no_grad():
        O.unet_forward(sd, x[:2], train=False)                       # warm-up
        t0 = time.perf_counter()
        done = 0
        while True:
            O.unet_forward(sd, x, train=False)
            done += n
            if time.perf_counter() - t0 > seconds_budget / 2 or done >= 64:
                break
        dt = time.perf_counter() - t0
    return {"value": round(done / dt, 3), "unit": "profiles/s", "cores": cores, "kind": "port",
            "sample": f"{done} synthetic {rows}x4096 profiles, eval forward in batches of {n}, torch fp32 oracle"}


TRAIN_FLOP_H5 = 175_574_876_160        # fwd + dgrad + wgrad per profile, H=5


def hbm_traffic_per_launch(tag):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside this process: the figure
    comes from the committed rocprofv3 passes of this same command (profiles/README.md; 2 x FETCH_SIZE + WRITE_SIZE,
    the gfx950 correction of MI355X_MICROARCH.md) and is reported only when it was measured for the kernel that
    dominates this run; otherwise null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    return round(rec["traffic_bytes_per_launch"]) if rec.get("kernel") == tag else None


def train_leg(args, dev, rank, world, distributed, x, t, barrier):
    """fwd (batch-stat BN) + Dice + bwd + Adam on the same synthetic batch; data parallel over RCCL when N > 1
    (global batch = 256 x N, Dice sums and the gradient arena all-reduced; BASELINE.json configs[2]/[3])."""
    from dnanet_amd.unet import HipAdam, HipUNet
    torch.manual_seed(0)
    model = HipUNet(4, (3, 5), 32, dev, dtype=args.dtype)
    model.train()
    opt = HipAdam(model, lr=1e-4, weight_decay=5e-4)
    if distributed:
        from dnanet_amd.distributed import DataParallelTrainer
        trainer = DataParallelTrainer(model, opt)
        step = lambda: trainer.step(x, t)
    else:
        def step():
            logits = model(x)
            sums = model.dice_forward(logits, t)
            model.backward()
            opt.step()
            opt.zero_grad()
            return sums
    steps = max(3, args.steps // 4)
    for _ in range(2):
        sums = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sums = step()
    loss = float(sums[4])              # the per-step host sync of the reference loop (trainable_unet.py:347), once here
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    pps = args.batch * world * steps / elapsed
    peak = BF16_PEAK_TFLOPS if args.dtype == "bf16" else F32_PEAK_TFLOPS
    return {"value": round(pps, 1), "unit": "profiles/s", "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "global_batch": args.batch * world, "loss": round(loss, 5),
            "parallelism": f"dp{world}" + (" (RCCL sum-all-reduce of Dice sums + bucketed gradient arena)" if world > 1 else ""),
            "whole_step_mfma_frac": round(pps / world * TRAIN_FLOP_H5 * args.rows / 5 / 1e12 / peak, 4)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipUNet

    torch.manual_seed(0)
    model = HipUNet(4, (3, 5), 32, dev, dtype=args.dtype)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():   # randomised BN statistics so the folding is exercised
        for key, _, kind in model.spec:
            if key.endswith("running_mean"):
                model.tensor(key).copy_(torch.randn(model.tensor(key).shape, generator=g))
            elif key.endswith("running_var"):
                model.tensor(key).copy_(0.5 + 1.5 * torch.rand(model.tensor(key).shape, generator=g))
    model.eval()
    x_np, t_np = make_batch(args.batch, args.rows, 4096, seed=1234 + rank)
    x = torch.from_numpy(x_np).to(dev)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.infer(x, want_logits=False)
    model.profile = None if os.environ.get("DN_BENCH_NO_EVENTS") else []     # (tuning aid: cost of the per-launch events)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.infer(x, want_logits=False)
    barrier()
    elapsed = time.perf_counter() - t0
    prof, model.profile = model.profile or [], None
    if distributed:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    value = args.batch * world * args.steps / elapsed
    flop_per_profile = FWD_FLOP_H5 * args.rows / 5

    # roofline of the dominant kernel from the per-launch events of the timed region
    by_tag = {}
    for tag, flops, e0, e1 in prof:
        d = by_tag.setdefault(tag, [0.0, 0.0, 0])
        d[0] += flops
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += 1
    roofline = None
    if by_tag:
        tag, (fl, sec, cnt) = max(by_tag.items(), key=lambda kv: kv[1][1])
        peak = BF16_PEAK_TFLOPS if args.dtype == "bf16" else F32_PEAK_TFLOPS
        achieved = fl / sec / 1e12
        roofline = {"bound": "mfma", "kernel": tag, "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": hbm_traffic_per_launch(tag), "launches": cnt,
                    "avg_launch_us": round(sec / cnt * 1e6, 2),
                    "share_of_step_time": round(sec / elapsed, 3),
                    "whole_net_frac": round(value / world * flop_per_profile / 1e12 / peak, 4)}

    out = {
        "metric": "profiles/sec U-Net inference (pretrained-shape U-Net, batch 256 per GPU)",
        "value": round(value, 1), "unit": "profiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"U-Net depth4 k(3,5) f32 eval forward, batch {args.batch}/GPU, {args.rows} dyes x 4096 scans "
                               "(BASELINE.json configs[1])", "global_batch": args.batch * world,
                   "parallelism": f"replicas x{world} (inference shards by profile, no collective)"},
        "roofline": roofline,
    }
    if not args.no_train:
        out["train_step"] = train_leg(args, dev, rank, world, distributed, x, torch.from_numpy(t_np).to(dev), barrier)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.rows)
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
