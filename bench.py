#!/usr/bin/env python3
"""Headline benchmark of the DNANet U-Net hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one eval-mode pass of the shipped U-Net (depth 4, kernel (3,5), 32 filters, config/models/unet.yaml) over one
batch of 256 synthetic 6-dye x 4096-scan electropherograms already resident in HBM (BASELINE.json configs[1]); ``value``
is that leg in **fp32**, the reference's precision and the mode the GPU tests prove mask- and allele-identical to the
reference (bf16 differs in a few dozen pixels / alleles -- it is a leg, not the headline).  With ``--gpus N`` and no
WORLD_SIZE in the environment this script starts the N ranks itself (``python -m torch.distributed.run`` as a child,
before anything touches the GPU); under a launcher it checks that the world size equals ``--gpus`` and refuses to run
otherwise.  Inference shards by profile, so every rank runs its own batch (weak scaling, no data-path collective) and
``value`` is the whole-job profiles/s.

The one JSON line also carries
  * ``roofline``: the dominant kernel of the headline leg, timed live with HIP events on the launch stream.  ``frac`` =
    FLOP of the MFMAs the kernel ISSUES / launch duration / dense MFMA peak of the dtype (MI355X_MICROARCH.md): the
    3x5 kernels leave out, at compile time, the (kernel row, output row) pairs that only meet the zero rows above /
    below the image (2 of 3 H), so the PyTorch-convention FLOP count overstates their work; that nominal figure is
    kept as ``frac_nominal``;
  * ``legs``: the other halves of configs[1] -- bf16 and the hi + lo split bf16 modes on the same batch, and fp32 /
    bf16 / bf16x3 on the reference's HIDImage shape (5 dyes) -- each with its own roofline;
  * ``train_step*``: fwd + Dice + bwd + Adam profiles/s at 256 profiles (fp32, bf16; 6 and 5 dyes), each with the HIP-event
    time beside the wall time, the caching-allocator deltas over the timed region (the step's buffers are a persistent
    arena: 0 device allocations) and the roofline of its dominant kernel, the weight gradient (from a profiled,
    single-stream step after the timed region); data parallel over RCCL when N > 1 (configs[3]);
  * ``train_step_n16_h5*``: configs[2] -- ``fit()``'s own path (resident set + HIP-graph replay) on 348 synthetic
    profiles at train.py's batch of 16: 21 steps per epoch, the 12-profile tail dropped;
  * ``parity``: mask pixels / called alleles that differ between the bf16 modes and the fp32 kernels on 64 profiles;
  * ``cpu_baseline``: the CPU oracle (a torch fp32 port of the reference path) timed on this box's host cores on a
    bounded sample, inference and one training step (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP_H5 = 58_531_512_320          # SURVEY 8d, per profile, H=5
TRAIN_FLOP_H5 = 175_574_876_160       # fwd + dgrad + wgrad per profile, H=5
CONV35_FLOP_H5 = 57_271_910_400       # the eighteen 3x5 convolutions of the forward pass, first one included
FIRST_FLOP_H5 = 19_660_800            # encoders.0 conv1 (1 -> 32): taps are the K dimension, nothing is skipped
BF16_PEAK_TFLOPS = 2500.0             # dense bf16 MFMA peak, MI355X_MICROARCH.md
F32_PEAK_TFLOPS = 157.3
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "r03_traffic.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="profiles per GPU per step")
    ap.add_argument("--rows", type=int, default=6, help="dye rows (6 = BASELINE.json wording, 5 = reference HIDImage)")
    ap.add_argument("--dtype", default="fp32", choices=["bf16", "fp32", "bf16x3", "bf16x3@2"],
                    help="precision of the headline leg; fp32 = the reference's arithmetic and the parity-proven mode "
                         "(bf16x3*: inference-only hi + lo split bf16; the training legs then run in bf16)")
    # (the bf16x3 inference modes are reported as legs: infer_bf16x3_h6, infer_bf16x3@2_h6, infer_bf16x3_h5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline leg only (no fp32 / 5-dye legs, no parity counts)")
    ap.add_argument("--global-batch", type=int, default=2048, help="configs[3]: global batch of the strong-scaling train leg (N > 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1: nccl = RCCL over xGMI (the measured configuration); gloo lets the whole "
                         "N > 1 path be rehearsed with several ranks sharing one GPU (collectives staged through the host)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="initialise the process group even with one rank (exercises the RCCL code path of the N > 1 run on one GPU)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous rehearsal on the CPU (gloo, stub step, no GPU): what the tests use")
    return ap.parse_args(argv)


def peak_of(dtype: str) -> float:
    return BF16_PEAK_TFLOPS if dtype.startswith("bf16") else F32_PEAK_TFLOPS


def spawn_ranks(args) -> int:
    """``--gpus N`` without a launcher: start one rank per GPU as a child job.  Nothing in this process has touched the
    GPU (importing torch does not), and the child is a child -- never an exec of a process that holds the device."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the only place the oracle is used)
# ---------------------------------------------------------------------------------------------------------------------
def host_cpu():
    """(physical cores this process may use, CPU model) from lscpu / the affinity mask."""
    model, cores_per_socket, sockets, threads_per_core = "unknown", None, None, 1
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                model = v
            elif k == "Core(s) per socket":
                cores_per_socket = int(v)
            elif k == "Socket(s)":
                sockets = int(v)
            elif k == "Thread(s) per core":
                threads_per_core = int(v)
    except Exception:
        pass
    physical = cores_per_socket * sockets if cores_per_socket and sockets else (os.cpu_count() or 1) // max(1, threads_per_core)
    return physical, model


def cpu_baseline(rows: int, seconds_budget: float = 24.0):
    """The oracle (oracle/unet_oracle.py, a port of the reference path) on the host cores: eval forward and one
    training step (fwd + Dice + bwd + Adam), N = 16 each."""
    from oracle import unet_oracle as O
    from dnanet_amd.synthetic import make_batch
    physical, model = host_cpu()
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)         # the box's CPU share for one GPU (16)
    try:
        share = min(share or physical, len(os.sched_getaffinity(0)))
    except AttributeError:
        share = share or physical
    cores = max(1, min(physical, share))
    torch.set_num_threads(cores)
    sd = O.init_state_dict(4, (3, 5), 32, seed=0, randomize_bn_stats=True)
    n = 16
    x_np, t_np = make_batch(n, rows, 4096, seed=1234)
    x, t = torch.from_numpy(x_np), torch.from_numpy(t_np)
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
    out = {"value": round(done / dt, 3), "unit": "profiles/s", "cores": cores, "kind": "port", "cpu_model": model,
           "physical_cores": physical,
           "sample": f"{done} synthetic {rows}x4096 profiles, eval forward in batches of {n}, torch fp32 oracle"}
    state = O.AdamState(O.trainable_keys(sd), sd)
    O.train_step(sd, x[:2], t[:2], state, 1e-4, 5e-4)                # warm-up
    t0 = time.perf_counter()
    steps = 0
    while True:
        O.train_step(sd, x, t, state, 1e-4, 5e-4)
        steps += 1
        if time.perf_counter() - t0 > seconds_budget / 2 or steps >= 3:
            break
    dt = time.perf_counter() - t0
    out["train_step"] = {"value": round(steps * n / dt, 3), "unit": "profiles/s", "ms_per_step": round(dt / steps * 1e3, 1),
                         "sample": f"{steps} steps of fwd + Dice + bwd + Adam on {n} synthetic {rows}x4096 profiles"}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# GPU legs
# ---------------------------------------------------------------------------------------------------------------------
# what the kernels behind a bench tag are compiled from (profiles/r03_traffic.json records the hash of these files)
TRAFFIC_SOURCES = {
    "conv_igemm": ["conv_igemm.h", "common.h", "api.hip", "conv_inst_bf16_k35.hip", "conv_inst_f32_k35.hip"],
    "conv_igemm8": ["conv_igemm8.h", "conv_igemm.h", "common.h", "api.hip", "conv8_inst_bf16.hip", "conv8_inst_f32.hip"],
    "conv_igemm8c": ["conv_igemm8.h", "conv_igemm.h", "common.h", "api.hip", "conv8_inst_bf16.hip", "conv8_inst_f32.hip"],
    "conv_wgrad": ["conv_wgrad.h", "common.h", "wgrad_api.hip", "wgrad_inst_bf16_k35.hip", "wgrad_inst_f32_k35.hip"],
}


def kernel_source_digest(names) -> str:
    import hashlib
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(ROOT, "dnanet_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def hbm_traffic(tag, avg_launch_us):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside this process: the figure is
    the one of the committed rocprofv3 passes of this same command (profiles/README.md; 2 x FETCH_SIZE + WRITE_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md).  Bytes per launch are a property of the kernel's code and the workload, not
    of the box's clock, so the figure is reported only for the kernel it was measured on, only while the sources that
    kernel is compiled from are byte-identical to the profiled ones (sha256 in the profile), and only while its live
    duration is within 10 % of the profiled one (the same binary spreads 5-7 % over the boxes of this pool);
    ``traffic_source`` says where it comes from."""
    try:
        with open(TRAFFIC_PROFILE) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, None
    rec = rec.get(tag)                                  # one record per kernel tag
    if not rec:
        return None, None
    rel = os.path.relpath(TRAFFIC_PROFILE, ROOT)
    try:
        if rec.get("kernel_source_sha256") != kernel_source_digest(rec.get("sources", [])):
            return None, f"{rel} (not reported: the kernel's sources changed since the profile)"
    except OSError:
        return None, None
    ref_us = rec.get("bench_avg_launch_us") or rec.get("rocprof_avg_launch_us")
    if not ref_us or abs(avg_launch_us - ref_us) > 0.10 * ref_us:
        return None, f"{rel} (not reported: live launch {avg_launch_us:.0f} us vs profiled {ref_us:.0f} us)"
    return round(rec["traffic_bytes_per_launch"]), f"{rel} (same kernel sources; live launch {avg_launch_us:.0f} us vs profiled {ref_us:.0f} us)"


def executed_share(rows: int) -> float:
    """Share of the nominal 3x5 MFMAs that are issued: the (kernel row, output row) pairs that only see the zero row
    above / below the image are left out at compile time -- 2 of 3*H pairs (conv_igemm.h, conv_igemm8.h, conv_wgrad.h)."""
    return 1.0 - 2.0 / (3.0 * rows)


def executed_flop(tag: str, nominal: float, rows: int) -> float:
    """FLOP of the MFMAs a launch issues, from its PyTorch-convention count: only the 3x5 implicit-GEMM kernels skip."""
    return nominal * executed_share(rows) if "3x5" in tag else nominal


def fwd_flop(rows: int, executed: bool) -> float:
    """Forward FLOP per profile: nominal or as issued (3x5 layers past the first x executed_share)."""
    skip = (CONV35_FLOP_H5 - FIRST_FLOP_H5) * (1.0 - executed_share(rows)) if executed else 0.0
    return (FWD_FLOP_H5 - skip) * rows / 5


def train_flop(rows: int, executed: bool) -> float:
    """fwd + dgrad + wgrad FLOP per profile; forward, data gradient and weight gradient all skip the same pairs."""
    skip = 3 * (CONV35_FLOP_H5 - FIRST_FLOP_H5) * (1.0 - executed_share(rows)) if executed else 0.0
    return (TRAIN_FLOP_H5 - skip) * rows / 5


def roofline_of(prof, dtype, rows, elapsed, pick=None):
    """Roofline record of the kernel with the largest total time (or the tag containing ``pick``) from per-launch
    (tag, nominal FLOP, start event, end event) records; returns (record, {tag: [nominal, executed, seconds, launches]})."""
    by_tag = {}
    for tag, flops, e0, e1 in prof:
        d = by_tag.setdefault(tag, [0.0, 0.0, 0.0, 0])
        d[0] += flops
        d[1] += executed_flop(tag, flops, rows)
        d[2] += e0.elapsed_time(e1) * 1e-3
        d[3] += 1
    if not by_tag:
        return None, by_tag
    cands = {k: v for k, v in by_tag.items() if pick in k} if pick else by_tag
    tag, (nominal, executed, sec, cnt) = max((cands or by_tag).items(), key=lambda kv: kv[1][2])
    peak = peak_of(dtype)
    rec = {"bound": "mfma", "kernel": tag, "achieved": round(executed / sec / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
           "frac": round(executed / sec / 1e12 / peak, 4), "frac_nominal": round(nominal / sec / 1e12 / peak, 4),
           "achieved_nominal": round(nominal / sec / 1e12, 1),
           "flop_basis": "MFMAs issued (3x5 kernels skip the all-zero halo-row pairs: 2 of 3*H); *_nominal = PyTorch-convention FLOP",
           "traffic": None, "traffic_source": None, "launches": cnt, "avg_launch_us": round(sec / cnt * 1e6, 2),
           "share_of_step_time": round(sec / elapsed, 3) if elapsed else None}
    return rec, by_tag


def make_eval_model(dtype, dev):
    from dnanet_amd.unet import HipUNet
    torch.manual_seed(0)
    model = HipUNet(4, (3, 5), 32, dev, dtype=dtype)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():   # randomised BN statistics so the folding is exercised
        for key, _, kind in model.spec:
            if key.endswith("running_mean"):
                model.tensor(key).copy_(torch.randn(model.tensor(key).shape, generator=g))
            elif key.endswith("running_var"):
                model.tensor(key).copy_(0.5 + 1.5 * torch.rand(model.tensor(key).shape, generator=g))
    model.eval()
    return model


def infer_leg(dtype, rows, batch, steps, warmup, dev, rank, world, barrier, reduce_max):
    """``steps`` timed eval passes over one resident batch; returns (profiles/s whole job, ms/step, roofline dict)."""
    from dnanet_amd.synthetic import make_batch
    model = make_eval_model(dtype, dev)
    x = torch.from_numpy(make_batch(batch, rows, 4096, seed=1234 + rank)[0]).to(dev)
    for _ in range(warmup):
        model.infer(x, want_logits=False)
    model.profile = None if os.environ.get("DN_BENCH_NO_EVENTS") else []     # (tuning aid: cost of the per-launch events)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.infer(x, want_logits=False)
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0)
    prof, model.profile = model.profile or [], None
    value = batch * world * steps / elapsed
    roofline, by_tag = roofline_of(prof, dtype, rows, elapsed)
    if roofline:
        peak = peak_of(dtype)
        if rows == 6 and batch == 256 and dtype in ("fp32", "bf16"):
            roofline["traffic"], roofline["traffic_source"] = hbm_traffic(roofline["kernel"], roofline["avg_launch_us"])
        conv_sec = sum(v[2] for v in by_tag.values())
        roofline.update({
            "all_conv_layers_frac": round(sum(v[1] for v in by_tag.values()) / conv_sec / 1e12 / peak, 4),
            "all_conv_layers_frac_nominal": round(sum(v[0] for v in by_tag.values()) / conv_sec / 1e12 / peak, 4),
            "whole_net_frac": round(value / world * fwd_flop(rows, True) / 1e12 / peak, 4),
            "whole_net_frac_nominal": round(value / world * fwd_flop(rows, False) / 1e12 / peak, 4)})
    stats = model.arena_stats()
    del model, x
    torch.cuda.empty_cache()
    if roofline:
        roofline["arena"] = stats
    return value, elapsed / steps * 1e3, roofline


ALLOC_KEYS = {"hipMalloc_calls": "num_device_alloc", "hipFree_calls": "num_device_free", "alloc_retries": "num_alloc_retries",
              "allocator_requests": "allocation.all.allocated"}


def alloc_snapshot(dev):
    st = torch.cuda.memory_stats(dev)
    return {k: int(st.get(v, 0)) for k, v in ALLOC_KEYS.items()}


def train_leg(dtype, rows, batch, steps, dev, rank, world, distributed, barrier, reduce_max, label=""):
    """fwd (batch-stat BN) + Dice + bwd + Adam on one synthetic batch per rank; data parallel over RCCL when N > 1
    (global batch = batch x N, Dice sums and the gradient arena all-reduced; BASELINE.json configs[2]/[3])."""
    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipAdam, HipUNet
    x_np, t_np = make_batch(batch, rows, 4096, seed=1234 + rank)
    x, t = torch.from_numpy(x_np).to(dev), torch.from_numpy(t_np).to(dev)
    torch.manual_seed(0)
    model = HipUNet(4, (3, 5), 32, dev, dtype=dtype)
    model.train()
    opt = HipAdam(model, lr=1e-4, weight_decay=5e-4)
    comm = None
    if distributed:
        import torch.distributed as dist
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
    for _ in range(2):
        sums = step()
    barrier()
    torch.cuda.reset_peak_memory_stats(dev)
    before = alloc_snapshot(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        sums = step()
    ev1.record()
    loss = float(sums[4])              # the per-step host sync of the reference loop (trainable_unet.py:347), once here
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0)
    event_ms = ev0.elapsed_time(ev1)
    after = alloc_snapshot(dev)
    allocator = {k: after[k] - before[k] for k in after}
    allocator["reserved_peak_GiB"] = round(torch.cuda.memory_stats(dev).get("reserved_bytes.all.peak", 0) / 2**30, 2)
    allocator["arena"] = model.arena_stats()
    # roofline of the step's dominant kernel: two more steps with per-launch HIP events (profiling keeps the weight
    # gradients on the main stream, so durations are not stretched by overlap); outside the timed region
    roofline = None
    if not distributed:
        model.profile = []
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        prof, model.profile = model.profile, None
        roofline, by_tag = roofline_of(prof, dtype, rows, None, pick="conv_wgrad")
        if roofline:
            if rows == 6 and batch == 256:
                roofline["traffic"], roofline["traffic_source"] = hbm_traffic(roofline["kernel"], roofline["avg_launch_us"])
            roofline["share_of_step_time"] = round(by_tag[roofline["kernel"]][2] / 2 / (elapsed / steps), 3)
            roofline["note"] = ("event pairs bracket dn_conv2d_wgrad = the split-K GEMM + its fixed-order slab reduction; "
                                "share_of_step_time is single-stream kernel time over the two-stream step time")
            peak = peak_of(dtype)
            groups = {}
            for tag, v in by_tag.items():
                g = "weight_gradient" if "wgrad" in tag else "data_gradient" if tag.endswith(":dgrad") else "forward_with_statistics"
                a = groups.setdefault(g, [0.0, 0.0, 0.0])
                a[0] += v[1]; a[1] += v[0]; a[2] += v[2]
            roofline["groups"] = {g: {"ms_per_step": round(a[2] / 2 * 1e3, 3), "frac": round(a[0] / a[2] / 1e12 / peak, 4),
                                      "frac_nominal": round(a[1] / a[2] / 1e12 / peak, 4)} for g, a in groups.items()}
    if distributed:
        buckets = trainer.reducer.last_buckets
        comm = {"backend": dist.get_backend(), "world_size_seen_by_rank0": dist.get_world_size(),
                "gradient_buckets_per_step": len(buckets), "bytes_all_reduced_per_step": int(sum(b - a for a, b in buckets) * 4),
                "dice_sums_all_reduced": 4, "batchnorm": "sync" if getattr(trainer, "sync_bn", False) else "per replica"}
    pps = batch * world * steps / elapsed
    out = {"value": round(pps, 1), "unit": "profiles/s", "dtype": "bf16" if dtype == "bf16" else "f32", "steps": steps,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "per_gpu_batch": batch, "global_batch": batch * world,
           "rows": rows, "loss": round(loss, 5),
           "parallelism": f"dp{world}" + (f" ({'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()} sum-all-reduce of Dice sums "
                                          "+ bucketed gradient arena)" if distributed else ""),
           "event_ms_per_step": round(event_ms / steps, 3),
           "event_vs_wall": round(event_ms / steps / (elapsed / steps * 1e3), 4),
           "allocator_in_timed_region": allocator,
           "whole_step_mfma_frac": round(pps / world * train_flop(rows, True) / 1e12 / peak_of(dtype), 4),
           "whole_step_mfma_frac_nominal": round(pps / world * train_flop(rows, False) / 1e12 / peak_of(dtype), 4),
           "roofline": roofline}
    if comm:
        out["comm"] = comm
    del model, opt, x, t
    torch.cuda.empty_cache()
    return out


def fit_leg(dtype, dev, epochs=4, profiles=348, batch=16, rows=5):
    """BASELINE.json configs[2] on ``fit()``'s own path (reference: train.py -> trainable_unet.py:183-203,339-347 with
    config/training/segmentation.yaml: batch 16, lr 1e-4, scheduler on): 348 synthetic profiles tensorised once and kept in
    HBM, 348 // 16 = 21 steps per epoch (the 12-profile tail is never trained on), the step replayed from a HIP graph.
    Epoch 1 carries the eager warm-up step and the capture; the figure is the median of the later epochs, timed around
    ``DNANet_UNet.epoch`` (per-step loss read-back and accuracy bookkeeping included)."""
    from dnanet_amd.data import SyntheticDataset
    from dnanet_amd.model import DNANet_UNet
    ds = SyntheticDataset(profiles, rows, 4096, seed=3)
    torch.manual_seed(0)
    m = DNANet_UNet(4, (3, 5), 32, device=str(dev), apply_allele_caller=False, dtype=dtype)
    times, losses = [], []
    inner = m.epoch

    def timed_epoch(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = inner(*a, **k)
        torch.cuda.synchronize()
        if k.get("train", True):
            times.append(time.perf_counter() - t0)
            losses.append(loss)
        return loss

    m.epoch = timed_epoch
    before = alloc_snapshot(dev)
    m.fit(ds, batch_size=batch, num_epochs=epochs, learning_rate=1e-4, use_scheduler=True, use_evaluation_metric=True)
    after = alloc_snapshot(dev)
    steps = profiles // batch
    steady = sorted(times[1:])
    t = steady[len(steady) // 2]
    out = {"value": round(steps * batch / t, 1), "unit": "profiles/s", "dtype": "bf16" if dtype == "bf16" else "f32",
           "ms_per_step": round(t / steps * 1e3, 3), "ms_per_epoch": round(t * 1e3, 2), "first_epoch_ms": round(times[0] * 1e3, 1),
           "epochs_timed": len(steady), "steps_per_epoch": steps, "batch": batch, "profiles": profiles,
           "profiles_never_trained_on": profiles - steps * batch, "rows": rows,
           "path": "DNANet_UNet.fit(): resident set + HIP-graph replay of the whole step (trainable_unet.py:183-203,339-347)",
           "epoch_losses": [round(float(l), 5) for l in losses],
           "whole_step_mfma_frac": round(steps * batch / t * train_flop(rows, True) / 1e12 / peak_of(dtype), 4),
           "allocator_whole_fit": {k: after[k] - before[k] for k in after}}
    del m, ds
    torch.cuda.empty_cache()
    return out


def parity_leg(dev, n=64):
    """bf16 kernels vs fp32 kernels at the shipped configuration, on a briefly trained net (sparse masks, like a real
    model) and on the seeded random-init net of the throughput legs: differing mask pixels and called alleles.  The
    fp32 kernels themselves are pinned to the reference by the GPU tests (tests/test_parity_bf16_gpu.py, ...)."""
    from dnanet_amd.parity import briefly_trained_state, disagreement
    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipUNet
    x = torch.from_numpy(make_batch(n, 5, 4096, seed=4321, unique=n)[0]).to(dev)
    out = {"what": f"bf16 / bf16x3 (hi + lo split) / bf16x3@2 (split on the two shallowest levels) kernels vs fp32 kernels, shipped "
                   f"config, {n} synthetic 5x4096 profiles; masks p >= 0.5, alleles by the NearestBasePairCaller hand-off"}
    states = {"briefly_trained": briefly_trained_state(4, (3, 5), 32, dev)}
    ref = make_eval_model("fp32", dev)
    states["random_init"] = {k: v.detach().cpu().clone() for k, v in ref.state_dict().items()}
    del ref
    for name, sd in states.items():
        probs = {}
        for dtype in ("fp32", "bf16", "bf16x3", "bf16x3@2"):
            m = HipUNet(4, (3, 5), 32, dev, dtype=dtype)
            m.load_state_dict(sd)
            m.eval()
            probs[dtype] = m.infer(x, want_logits=False)[1]
            del m
        out[name] = disagreement(probs["bf16"], probs["fp32"], x)
        for dtype in ("bf16x3", "bf16x3@2"):
            out[name][dtype] = {k: v for k, v in disagreement(probs[dtype], probs["fp32"], x).items()
                                if k in ("mask_pixels_differing", "alleles_differing", "profiles_with_allele_difference", "max_abs_prob_diff")}
    torch.cuda.empty_cache()
    return out


def dry_run(args, rank, world):
    """Rendezvous rehearsal without a GPU: gloo group, a stub step with the same barrier / max-over-ranks timing."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    ranks = [None] * world
    if world > 1:
        dist.all_gather_object(ranks, rank)
    else:
        ranks = [0]
    t0 = time.perf_counter()
    acc = torch.zeros(4)
    for _ in range(args.steps):
        acc += 1
        if world > 1:
            dist.all_reduce(acc)
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "profiles/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ranks_seen": sorted(ranks),
                          "world_size_seen_by_rank0": dist.get_world_size() if world > 1 else 1, "dry_run": True,
                          "ms_per_step": round(float(elapsed) / max(1, args.steps) * 1e3, 3)}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))                     # parent: argument parsing only, no GPU call before this point
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); refusing to mislabel the run",
                  file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, rank, world)
    distributed = world > 1 or args.force_distributed
    if distributed and "MASTER_ADDR" not in os.environ:       # one rank without a launcher
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        print(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible); RCCL needs one device per rank", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank % max(ndev, 1))        # gloo rehearsal: ranks may share a device
    torch.cuda.set_device(dev)
    if distributed:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(seconds):
        if not distributed:
            return seconds
        tmax = torch.tensor([seconds], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax)

    common = (dev, rank, world, barrier, reduce_max)
    value, ms, roofline = infer_leg(args.dtype, args.rows, args.batch, args.steps, args.warmup, *common)
    dt = lambda d: d if d.startswith("bf16") else "f32"
    out = {
        "metric": "profiles/sec U-Net inference (pretrained-shape U-Net, batch 256 per GPU)",
        "value": round(value, 1), "unit": "profiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dt(args.dtype), "data": "synthetic",
        "config": {"workload": f"U-Net depth4 k(3,5) f32 eval forward, batch {args.batch}/GPU, {args.rows} dyes x 4096 scans "
                               f"(BASELINE.json configs[1], {args.dtype} half)", "global_batch": args.batch * world,
                   "parallelism": f"replicas x{world} (inference shards by profile, no collective)",
                   "world_size_seen_by_rank0": world, **({"backend": args.backend, "devices": ndev} if distributed else {}),
                   "precision": "fp32 = the reference's arithmetic; masks and called alleles identical to the reference "
                                "(tests/test_parity_bf16_gpu.py, tests/test_shipped_golden_gpu.py); bf16 / bf16x3 are under legs"
                                if args.dtype == "fp32" else f"{args.dtype} (not the parity mode: see parity)"},
        "roofline": roofline,
    }
    if not args.no_legs and not distributed:
        # the other halves of configs[1]: the reference's precision on the same batch, and the reference's HIDImage shape
        legs = {}
        # bf16x3: hi + lo split bf16 tensors, three MFMAs per product -- fp32-grade masks from the bf16 matrix cores (every
        # level, or only the two shallowest: "bf16x3@2"); their roofline counts a product once (algorithmic FLOP)
        todo = {f"infer_{dt(d)}_h{args.rows}": (d, args.rows) for d in ("fp32", "bf16", "bf16x3", "bf16x3@2") if d != args.dtype}
        todo.update({f"infer_{dt(d)}_h5": (d, 5) for d in ("bf16", "fp32", "bf16x3") if (d, 5) != (args.dtype, args.rows)})
        for name, (d, rows) in todo.items():
            steps = max(3, args.steps // 2) if d == "fp32" else max(5, args.steps)
            v, m, r = infer_leg(d, rows, args.batch, steps, min(2, args.warmup), *common)
            legs[name] = {"value": round(v, 1), "unit": "profiles/s", "dtype": dt(d), "rows": rows, "batch": args.batch,
                          "steps": steps, "ms_per_step": round(m, 3), "roofline": r}
        out["legs"] = legs
        out["parity"] = parity_leg(dev)
    if not args.no_train:
        tsteps = max(3, args.steps // 4)
        if args.dtype.startswith("bf16x3"):
            args.dtype = "bf16"            # inference-only precision: the training legs are the bf16 ones
        out["train_step"] = train_leg(args.dtype, args.rows, args.batch, tsteps, dev, rank, world, distributed, barrier, reduce_max)
        if distributed and args.global_batch % world == 0:
            # configs[3]: the same global batch on 2 / 4 / 8 GPUs (strong scaling of the training step)
            per = args.global_batch // world
            try:
                out["train_step_global"] = train_leg(args.dtype, args.rows, per, 3, dev, rank, world, True, barrier, reduce_max)
                out["train_step_global"]["scaling"] = "strong"
            except Exception as exc:      # e.g. out of memory at 1024 profiles per GPU: report, keep the line
                out["train_step_global"] = {"error": f"{type(exc).__name__}: {exc}"[:300], "per_gpu_batch": per}
        if distributed and args.dtype != "bf16" and not args.no_legs:
            # the throughput precision on the same data-parallel path (256 per GPU, and configs[3]'s global batch)
            out["train_step_bf16"] = train_leg("bf16", args.rows, args.batch, max(5, args.steps // 2), dev, rank, world, True, barrier, reduce_max)
            if args.global_batch % world == 0:
                try:
                    out["train_step_global_bf16"] = train_leg("bf16", args.rows, args.global_batch // world, 3, dev, rank, world, True,
                                                              barrier, reduce_max)
                    out["train_step_global_bf16"]["scaling"] = "strong"
                except Exception as exc:
                    out["train_step_global_bf16"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if not distributed and not args.no_legs:
            other = "fp32" if args.dtype == "bf16" else "bf16"
            osteps = tsteps if other == "fp32" else max(5, args.steps // 2)
            out[f"train_step_{dt(other)}"] = train_leg(other, args.rows, args.batch, osteps, dev, rank, world, False, barrier, reduce_max)
            out["train_step_bf16_h5"] = train_leg("bf16", 5, args.batch, max(5, args.steps // 2), dev, rank, world, False, barrier, reduce_max)
            out["train_step_f32_h5"] = train_leg("fp32", 5, args.batch, tsteps, dev, rank, world, False, barrier, reduce_max)
            # configs[2]: fit() at train.py's batch of 16 over 348 profiles
            out["train_step_n16_h5"] = fit_leg("fp32", dev)
            out["train_step_n16_h5_bf16"] = fit_leg("bf16", dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.rows)
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
