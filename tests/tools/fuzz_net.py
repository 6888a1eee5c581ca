"""One-off: whole-net inference + one training step against the oracle on random small configurations (fp32 and bf16 modes; inference also in the bf16x3 modes)."""
import os, sys, random; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from collections import OrderedDict
import torch
from dnanet_amd.unet import HipAdam, HipUNet
from oracle import unet_oracle as O
dev = torch.device('cuda:0')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(ncase):
    depth = rng.choice([1, 2, 3, 4]); k = rng.choice([(3, 5), (3, 5), (1, 3)]); f = rng.choice([8, 16, 24, 32])
    n = rng.choice([1, 2, 3, 5]); h = rng.choice([1, 3, 4, 5, 6, 7]); w = (1 << depth) * rng.choice([4, 8, 9, 16, 20, 32])
    torch.manual_seed(case)
    sd = O.init_state_dict(depth, k, f, seed=case, randomize_bn_stats=True)
    x = (torch.rand(n, 1, h, w) * 2000).round()
    t = (torch.rand(n, 1, h, w) < 0.1).to(torch.int8)
    ref_eval = O.unet_forward(sd, x, train=False)
    ref_sd = OrderedDict((kk, v.clone()) for kk, v in sd.items())
    state = O.AdamState(O.trainable_keys(ref_sd), ref_sd)
    ref_loss, ref_logits, ref_grads = O.train_step(ref_sd, x, t, state, 1e-3, 5e-4)
    msgs = []
    for dtype in ('fp32', 'bf16'):
        m = HipUNet(depth, k, f, dev, dtype=dtype); m.load_state_dict(sd)
        m.eval(); lg, pr = m.infer(x.to(dev))
        tol = 2e-5 if dtype == 'fp32' else 6e-2
        e = float((lg.cpu() - ref_eval).abs().max() / (ref_eval.abs().max() + 1e-20))
        if not e < tol: msgs.append(f'{dtype} eval logits {e:.2e}')
        m.train(); opt = HipAdam(m, lr=1e-3, weight_decay=5e-4)
        logits = m(x.to(dev)); sums = m.dice_forward(logits, t.to(dev)); m.backward()
        if abs(float(sums[4]) - ref_loss) > (1e-5 if dtype == 'fp32' else 3e-2): msgs.append(f'{dtype} loss {float(sums[4])} vs {ref_loss}')
        if dtype == 'fp32':
            for kk, g in ref_grads.items():
                if kk.endswith("double_conv.0.bias") or kk.endswith("double_conv.3.bias"): continue
                l2 = float((m.grad(kk).cpu() - g).norm() / (g.norm() + 1e-20))
                if not l2 < 8e-3: msgs.append(f'grad {kk} l2 {l2:.2e}')
            opt.step()
            got = m.state_dict()
            for kk in ref_sd:
                if 'running' in kk and not torch.allclose(got[kk].cpu(), ref_sd[kk], rtol=1e-4, atol=1e-4): msgs.append(f'{kk} running stat')
    for dtype in ('bf16x3', f'bf16x3@{rng.randint(1, depth + 1)}'):       # inference-only split modes
        m = HipUNet(depth, k, f, dev, dtype=dtype); m.load_state_dict(sd); m.eval()
        lg, pr = m.infer(x.to(dev))
        e = float((lg.cpu() - ref_eval).abs().max() / (ref_eval.abs().max() + 1e-20))
        if not e < (1e-4 if m.split_levels == depth + 1 else 6e-2): msgs.append(f'{dtype} eval logits {e:.2e}')
    if msgs:
        bad += 1
        print('CASE', dict(depth=depth, k=k, f=f, n=n, h=h, w=w), msgs[:4])
print(f'{ncase} configurations, {bad} with deviations')
