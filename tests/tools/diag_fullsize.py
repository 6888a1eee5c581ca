import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from collections import OrderedDict
import torch
from dnanet_amd.synthetic import make_batch
from dnanet_amd.unet import HipAdam, HipUNet
from oracle import unet_oracle as O
torch.set_num_threads(16)
dev = torch.device('cuda:0')
for seed in (3, 4):
    sd = O.init_state_dict(4, (3, 5), 32, seed=seed)
    xn, tn = make_batch(2, 5, 4096, seed=7, unique=2)
    x, t = torch.from_numpy(xn), torch.from_numpy(tn)
    ref_sd = OrderedDict((k, v.clone()) for k, v in sd.items())
    state = O.AdamState(O.trainable_keys(ref_sd), ref_sd)
    ref_loss, ref_logits, ref_grads = O.train_step(ref_sd, x, t, state, 1e-4, 5e-4)
    m = HipUNet(4, (3, 5), 32, dev, dtype='fp32'); m.load_state_dict(sd); m.train()
    logits = m(x.to(dev)); sums = m.dice_forward(logits, t.to(dev)); m.backward()
    print('seed', seed, 'loss', float(sums[4]), ref_loss, 'logits err', float((logits.cpu() - ref_logits).abs().max() / ref_logits.abs().max()))
    rows = []
    for k, g in ref_grads.items():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"): continue
        d = (m.grad(k).cpu() - g)
        rows.append((float(d.abs().max() / (g.abs().max() + 1e-20)), float(d.norm() / (g.norm() + 1e-20)), k))
    for e, l2, k in sorted(rows, reverse=True)[:6]:
        print(f'   max {e:.4f}  l2 {l2:.5f}  {k}')
