"""Synchronised BatchNorm (optional, data parallel): "R ranks x N/R profiles == 1 rank x N profiles".

* a process group of ONE rank: the totals path (partials -> fp64 totals -> all-reduce -> finalize) must reproduce the plain
  step bit for bit;
* TWO ranks sharing this box's one GPU (gloo carries the all-reduces; RCCL refuses two ranks on one device): each trains
  on half of a batch with sync_bn=True; parameters, BatchNorm running statistics and the loss after two steps must equal
  a single-process run on the whole batch (fp32; not bitwise -- the split-K order of the weight gradient depends on the
  batch size -- but to round-off), while per-replica BatchNorm visibly does not."""
import os
import socket

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _steps(m, opt, x, t, trainer=None, n=2):
    for _ in range(n):
        if trainer is not None:
            sums = trainer.step(x, t)
        else:
            logits = m(x); sums = m.dice_forward(logits, t); m.backward(); opt.step(); opt.zero_grad()
    return float(sums[4])


def test_sync_bn_world1_is_bitwise_the_plain_step(gpu_device):
    import torch.distributed as dist
    from dnanet_amd.distributed import DataParallelTrainer
    from dnanet_amd.unet import HipAdam, HipUNet
    g = load_golden("g_d3_k35_f8_h5_w128")
    x, t = torch.from_numpy(g["x"]).to(gpu_device), torch.from_numpy(g["t"]).to(gpu_device)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu_device)
    try:
        out = []
        for dtype in ("fp32", "bf16"):
            for sync in (False, True):
                m = HipUNet(3, (3, 5), 8, gpu_device, dtype=dtype)
                m.load_state_dict(g["sd0"])
                m.train()
                opt = HipAdam(m, lr=1e-3, weight_decay=5e-4)
                tr = DataParallelTrainer(m, opt, min_bucket_elems=10000, sync_bn=sync)
                loss = _steps(m, opt, x, t, tr)
                out.append((m.flat_params.clone(), m.flat_stats.clone(), loss))
            assert torch.equal(out[-2][0], out[-1][0]) and torch.equal(out[-2][1], out[-1][1]) and out[-2][2] == out[-1][2], dtype
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_on_one_stream_equals_two_streams(gpu_device):
    """Above ``two_stream_max_pixels`` (1024 six-dye profiles per GPU: configs[3] on 2 GPUs) backward keeps the weight gradients
    on the main stream and the bucket hooks are called in stream order right behind them.  Forced here on a small net, through
    the RCCL group of one rank: the step must be bitwise the two-stream step and the plain step, with the same buckets."""
    import torch.distributed as dist
    from dnanet_amd.distributed import DataParallelTrainer
    from dnanet_amd.unet import HipAdam, HipUNet
    g = load_golden("g_d3_k35_f8_h5_w128")
    x, t = torch.from_numpy(g["x"]).to(gpu_device), torch.from_numpy(g["t"]).to(gpu_device)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu_device)
    try:
        out = []
        for limit, parallel in ((20_000_000, True), (0, True), (20_000_000, False)):
            m = HipUNet(3, (3, 5), 8, gpu_device, dtype="fp32")
            m.load_state_dict(g["sd0"])
            m.train()
            m.two_stream_max_pixels = limit
            opt = HipAdam(m, lr=1e-3, weight_decay=5e-4)
            tr = DataParallelTrainer(m, opt, min_bucket_elems=10000) if parallel else None
            calls = []
            if tr is not None:
                inner = tr.reducer.segment_done
                tr.reducer.segment_done = lambda name: (calls.append((name, torch.cuda.current_stream() == m._wgrad_stream)),
                                                        inner(name))[1]
            loss = _steps(m, opt, x, t, tr)
            out.append((m.flat_params.clone(), m.flat_stats.clone(), loss, tr.reducer.last_buckets if tr else None, calls))
        two, one, plain = out
        assert [c[0] for c in one[4][:8]] == ["head", "decoders.2", "decoders.1", "decoders.0", "bottleneck", "encoders.2",
                                               "encoders.1", "encoders.0"]
        assert all(on_side for _, on_side in two[4]) and not any(on_side for _, on_side in one[4])
        assert one[3] == two[3] and len(one[3]) >= 2
        for a, b in ((one, two), (one, plain)):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # (the trainer forms the loss from the all-reduced sums with torch arithmetic, the plain step inside the Dice kernel)
        assert one[2] == two[2] and abs(one[2] - plain[2]) < 1e-6
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, sync_bn, out_path):
    import torch.distributed as dist
    from conftest import load_golden as lg
    from dnanet_amd.distributed import DataParallelTrainer
    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipAdam, HipUNet
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)      # a rank stuck in a collective ends (and says where) before pytest's limit
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = lg("g_d3_k35_f8_h5_w128")
        xn, tn = make_batch(8, 5, 256, seed=31, unique=8)
        per = 8 // world
        x = torch.from_numpy(xn[rank * per:(rank + 1) * per]).to(dev)
        t = torch.from_numpy(tn[rank * per:(rank + 1) * per]).to(dev)
        m = HipUNet(3, (3, 5), 8, dev, dtype="fp32")
        m.load_state_dict(g["sd0"])
        m.train()
        opt = HipAdam(m, lr=1e-3, weight_decay=5e-4)
        tr = DataParallelTrainer(m, opt, min_bucket_elems=10000, sync_bn=sync_bn)
        # the first step piece by piece (what DataParallelTrainer.step does), to look at the all-reduced gradients
        from dnanet_amd.distributed import all_reduce_dice_sums
        sums = all_reduce_dice_sums(m.dice_forward(m.forward_train(x), t))
        m.backward(sums=sums, segment_hook=tr.reducer.segment_done)
        tr.reducer.finish()
        grads = m.flat_grads.cpu()
        opt.step(); opt.zero_grad()
        loss = _steps(m, opt, x, t, tr, n=1)
        if rank == 0:
            torch.save({"params": m.flat_params.cpu(), "stats": m.flat_stats.cpu(), "loss": loss, "grads": grads}, out_path)
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_sync_bn_equal_one_rank_on_the_whole_batch(gpu_device, tmp_path):
    import torch.multiprocessing as mp
    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipAdam, HipUNet
    g = load_golden("g_d3_k35_f8_h5_w128")
    xn, tn = make_batch(8, 5, 256, seed=31, unique=8)
    m = HipUNet(3, (3, 5), 8, gpu_device, dtype="fp32")
    m.load_state_dict(g["sd0"])
    m.train()
    opt = HipAdam(m, lr=1e-3, weight_decay=5e-4)
    xd, td = torch.from_numpy(xn).to(gpu_device), torch.from_numpy(tn).to(gpu_device)
    m.dice_forward(m(xd), td); m.backward()
    ref_g = m.flat_grads.cpu()
    opt.step(); opt.zero_grad()
    loss1 = _steps(m, opt, xd, td, n=1)
    ref_p, ref_s = m.flat_params.cpu(), m.flat_stats.cpu()
    results = {}
    for sync in (True, False):
        out = tmp_path / f"sync{int(sync)}.pt"
        mp.spawn(_two_rank_worker, args=(2, _free_port(), sync, str(out)), nprocs=2, join=True)
        results[sync] = torch.load(out, weights_only=True)
    r = results[True]
    assert abs(r["loss"] - loss1) < 1e-6, (r["loss"], loss1)
    # the all-reduced gradient of step 1 is the single-device gradient up to fp32 summation order (the split-K order of the
    # weight gradient depends on the batch size); per-replica statistics give a different gradient altogether
    rel = lambda gr: float((gr - ref_g).norm() / ref_g.norm())
    print(f"\ngradient rel-L2 vs one rank on the whole batch: sync BN {rel(r['grads']):.2e}, per-replica BN {rel(results[False]['grads']):.2e}")
    assert rel(r["grads"]) < 1e-5, rel(r["grads"])          # measured 4.9e-07
    assert rel(results[False]["grads"]) > 30 * rel(r["grads"])
    assert torch.allclose(r["stats"], ref_s, rtol=1e-5, atol=1e-6)
    assert float((results[False]["stats"] - ref_s).abs().max()) > 1e-3
    # Adam divides by sqrt(v): an element whose gradient is rounding noise moves by up to +-lr per step in either run
    assert float((r["params"] - ref_p).abs().max()) <= 2.5 * 1e-3 * 2
