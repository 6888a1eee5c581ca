"""Is the fp32 gradient deviation round-off?  HIP fp32 and the fp32 oracle are both compared with the oracle in fp64."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from collections import OrderedDict
import torch
from dnanet_amd.unet import HipUNet
from oracle import unet_oracle as O
dev = torch.device('cuda:0')
for (depth, k, f, n, h, w, case) in [(4, (3, 5), 24, 3, 7, 256, 13), (4, (3, 5), 32, 5, 7, 512, 31), (4, (3, 5), 32, 2, 5, 4096, 3)]:
    torch.manual_seed(case)
    sd = O.init_state_dict(depth, k, f, seed=case, randomize_bn_stats=True)
    x = (torch.rand(n, 1, h, w) * 2000).round(); t = (torch.rand(n, 1, h, w) < 0.1).to(torch.int8)
    def run(dt):
        s = OrderedDict((kk, v.clone().to(dt) if v.is_floating_point() else v.clone()) for kk, v in sd.items())
        st = O.AdamState(O.trainable_keys(s), s)
        return O.train_step(s, x.to(dt), t, st, 1e-3, 5e-4)
    l32, _, g32 = run(torch.float32)
    l64, _, g64 = run(torch.float64)
    m = HipUNet(depth, k, f, dev, dtype='fp32'); m.load_state_dict(sd); m.train()
    logits = m(x.to(dev)); sums = m.dice_forward(logits, t.to(dev)); m.backward()
    worst = []
    for kk, g in g64.items():
        if kk.endswith("double_conv.0.bias") or kk.endswith("double_conv.3.bias"): continue
        nrm = float(g.norm()) + 1e-30
        worst.append((float((m.grad(kk).cpu().double() - g).norm()) / nrm, float((g32[kk].double() - g).norm()) / nrm, kk))
    worst.sort(reverse=True)
    print(dict(depth=depth, f=f, n=n, h=h, w=w), 'loss hip/32/64', float(sums[4]), float(l32), float(l64))
    for a, b, kk in worst[:4]: print(f'   vs fp64: hip {a:.2e}  torch-fp32 {b:.2e}  {kk}')
