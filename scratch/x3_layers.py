"""Per-launch times of the eval forward in a given dtype (bf16 / bf16x3 / bf16x3@2), N = 256: python scratch/x3_layers.py [dtype] [rows]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd.unet import HipUNet
from dnanet_amd.synthetic import make_batch
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16x3@2"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = HipUNet(4, (3, 5), 32, dev, dtype=dtype); m.eval()
x = torch.from_numpy(make_batch(256, rows, 4096, seed=1)[0]).to(dev)
for _ in range(3): m.infer(x, want_logits=False)
m.profile = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): m.infer(x, want_logits=False)
e1.record(); torch.cuda.synchronize()
prof, m.profile = m.profile, None
n = len(prof) // 5
print(f"{dtype} rows {rows}: {e0.elapsed_time(e1) / 5:.3f} ms per batch; conv launches per pass {n}")
tot = 0
for i in range(n):
    us = sum(prof[k * n + i][2].elapsed_time(prof[k * n + i][3]) for k in range(5)) / 5 * 1e3
    tot += us
    print(f"{i:2d} {prof[i][0]:48s} {us:8.1f} us  {prof[i][1] / us / 1e6:8.1f} TFLOP/s nominal")
print(f"sum of timed convolutions {tot / 1e3:.3f} ms")
