"""Train-step timing: python scratch/bench_train.py [rows=6] ["bf16:256,bf16:16,fp32:32"]"""
import sys, time; sys.path.insert(0, '.')
import torch
from dnanet_amd.unet import HipUNet, HipAdam
from dnanet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
TRAIN = 175_574_876_160
h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cases = [(c.split(':')[0], int(c.split(':')[1])) for c in (sys.argv[2] if len(sys.argv) > 2 else "bf16:256,bf16:16").split(',')]
for dtype, n in cases:
    m = HipUNet(4, (3, 5), 32, dev, dtype=dtype); m.train()
    opt = HipAdam(m, lr=1e-4, weight_decay=5e-4)
    xn, tn = make_batch(n, h, 4096)
    x, t = torch.from_numpy(xn).to(dev), torch.from_numpy(tn).to(dev)
    def step():
        logits = m(x); sums = m.dice_forward(logits, t); m.backward(); opt.step(); opt.zero_grad(); return sums
    for _ in range(2): s = step()
    torch.cuda.synchronize()
    reps = 5; t0 = time.perf_counter()
    for _ in range(reps): s = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f'{dtype} N={n} H={h}: {dt*1e3:.2f} ms/step  {n/dt:.0f} profiles/s  {n/dt*TRAIN*h/5/1e12:.1f} TFLOP/s  loss {float(s[4]):.4f} mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB', flush=True)
    del m, opt, x, t; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
