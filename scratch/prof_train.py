import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd.unet import HipUNet, HipAdam
from dnanet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
n, h = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 5
m = HipUNet(4, (3, 5), 32, dev, dtype='bf16'); m.train()
opt = HipAdam(m, lr=1e-4, weight_decay=5e-4)
xn, tn = make_batch(n, h, 4096)
x, t = torch.from_numpy(xn).to(dev), torch.from_numpy(tn).to(dev)
for _ in range(4):
    logits = m(x); sums = m.dice_forward(logits, t); m.backward(); opt.step(); opt.zero_grad()
torch.cuda.synchronize()
