import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd.unet import HipUNet, HipAdam
from dnanet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = HipUNet(4, (3, 5), 32, dev, dtype='bf16'); m.train()
opt = HipAdam(m, lr=1e-4, weight_decay=5e-4)
xn, tn = make_batch(n, 5, 4096)
x, t = torch.from_numpy(xn).to(dev), torch.from_numpy(tn).to(dev)
def step():
    logits = m(x); sums = m.dice_forward(logits, t); m.backward(); opt.step(); opt.zero_grad(); return sums
for _ in range(3): s = step()
torch.cuda.synchronize()
reps = 20; t0 = time.perf_counter()
for _ in range(reps): s = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f'{os.environ.get("TAG","default")}: bf16 N={n}: {dt*1e3:.2f} ms/step  {n/dt:.0f} profiles/s')
