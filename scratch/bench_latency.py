"""Eval-forward latency at small batches (kernel side, inputs resident): python scratch/bench_latency.py [dtype]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd.unet import HipUNet
from dnanet_amd.synthetic import make_batch
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = HipUNet(4, (3, 5), 32, dev, dtype=dtype); m.eval()
for n in (1, 2, 4, 8, 16, 32, 64):
    x = torch.from_numpy(make_batch(n, 5, 4096, seed=1)[0]).to(dev)
    for _ in range(5): m.infer(x, want_logits=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.infer(x, want_logits=False)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50 * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print(f"{dtype} N={n:3d}: {wall:7.3f} ms per call  {n / wall * 1e3:9.0f} profiles/s")
