"""Follow-up to diag_bf16_where.py: at U-Net levels 0-1, is it the rounding of the WEIGHTS or of the stored ACTIVATIONS that
flips mask pixels, and is a two-term bf16 split (hi + lo, ~16 mantissa bits) of either enough?  CPU emulation, briefly trained
net.     python tests/tools/diag_bf16_split.py [steps] [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.nn.functional as F
from oracle import unet_oracle as O
from dnanet_amd.synthetic import make_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_eval = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
sd = O.init_state_dict(4, (3, 5), 32, seed=0)
xn, tn = make_batch(32, 5, 4096, seed=77, unique=32)
x, t = torch.from_numpy(xn), torch.from_numpy(tn)
state = O.AdamState(O.trainable_keys(sd), sd)
for s in range(steps):
    i = (s % 4) * 8
    O.train_step(sd, x[i:i + 8], t[i:i + 8], state, 2e-3, 5e-4)

def q(v, mode):
    """mode: 'x' exact, 'b' bf16, 's' hi+lo bf16 split"""
    if mode == 'x':
        return v
    hi = v.bfloat16().float()
    return hi if mode == 'b' else hi + (v - hi).bfloat16().float()

def dconv(prefix, x, mw, ma):
    for conv_idx, bn_idx in ((0, 1), (3, 4)):
        w, b = sd[f"{prefix}.double_conv.{conv_idx}.weight"], sd[f"{prefix}.double_conv.{conv_idx}.bias"]
        bn = f"{prefix}.double_conv.{bn_idx}"
        scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
        shift = (b - sd[bn + ".running_mean"]) * scale + sd[bn + ".bias"]
        wf = w * scale.view(-1, 1, 1, 1)
        x = F.relu(F.conv2d(x, wf if w.shape[1] == 1 else q(wf, mw), shift, padding=(1, 2)))
        x = q(x, ma)
    return x

def forward(x, mw01, ma01):
    """weights / activations mode at levels 0-1; levels 2-4 plain bf16"""
    mw = lambda l: mw01 if l < 2 else 'b'
    ma = lambda l: ma01 if l < 2 else 'b'
    skips = []
    for i in range(4):
        s = dconv(f"encoders.{i}.conv", x, mw(i), ma(i))
        skips.append(s)
        x = q(F.max_pool2d(s, (1, 2)), ma(i + 1))
    x = dconv("bottleneck", x, 'b', 'b')
    for i in range(4):
        l = 3 - i
        s = skips.pop()
        up = F.conv_transpose2d(x, q(sd[f"decoders.{i}.up.weight"], mw(l)), sd[f"decoders.{i}.up.bias"], stride=(1, 2))
        x = torch.cat([q(up, ma(l)), s], 1)
        x = dconv(f"decoders.{i}.conv", x, mw(l), ma(l))
    return F.conv2d(x, sd["outputs.weight"], sd["outputs.bias"])

xe = torch.from_numpy(make_batch(n_eval, 5, 4096, seed=4321, unique=n_eval)[0])
with torch.no_grad():
    ref = O.unet_forward(sd, xe, train=False)
    for name, mw, ma in (("all bf16", 'b', 'b'), ("L0-1: weights exact, acts bf16", 'x', 'b'), ("L0-1: weights bf16, acts exact", 'b', 'x'),
                         ("L0-1: weights split, acts bf16", 's', 'b'), ("L0-1: weights bf16, acts split", 'b', 's'),
                         ("L0-1: both split", 's', 's'), ("L0-1: both exact", 'x', 'x')):
        z = forward(xe, mw, ma)
        flips = int(((z >= 0) != (ref >= 0)).sum())
        err = float((z - ref).abs().max() / ref.abs().max())
        print(f"{name:34s}: mask pixels differing {flips:5d} of {int((ref >= 0).sum())} positives, logits rel err {err:.2e}", flush=True)
