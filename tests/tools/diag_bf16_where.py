"""Where does the bf16 mask disagreement come from?  CPU emulation of bf16 storage (activations and folded weights rounded
to bf16, fp32 accumulation -- what the bf16 kernels do) with chosen U-Net levels kept exact, on a briefly trained net.
Prints mask pixels that differ from the all-fp32 result per variant.     python tests/tools/diag_bf16_where.py [steps] [n]
(CPU only, a few minutes; decides whether a mixed-precision top level is worth building: DESIGN.md section 4)"""
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
    loss, _, _ = O.train_step(sd, x[i:i + 8], t[i:i + 8], state, 2e-3, 5e-4)
    if s % 10 == 0:
        print("step", s, "loss", round(loss, 4), flush=True)

r = lambda v, on: v.bfloat16().float() if on else v


def dconv(prefix, x, lo_in, lo_w, lo_out):
    """eval-mode DoubleConv with BN folded (as the kernels run it); lo_* = round to bf16 at that point."""
    for conv_idx, bn_idx in ((0, 1), (3, 4)):
        w, b = sd[f"{prefix}.double_conv.{conv_idx}.weight"], sd[f"{prefix}.double_conv.{conv_idx}.bias"]
        bn = f"{prefix}.double_conv.{bn_idx}"
        scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
        shift = (b - sd[bn + ".running_mean"]) * scale + sd[bn + ".bias"]
        first = w.shape[1] == 1
        wf = w * scale.view(-1, 1, 1, 1)
        x = F.relu(F.conv2d(x, wf if first else r(wf, lo_w), shift, padding=(1, 2)))
        x = r(x, lo_out)
    return x


def forward(x, exact_levels=(), exact_last=False):
    lo = lambda l: l not in exact_levels
    skips = []
    for i in range(4):
        s = dconv(f"encoders.{i}.conv", x, lo(i), lo(i), lo(i))
        skips.append(s)
        x = r(F.max_pool2d(s, (1, 2)), lo(i + 1) if i + 1 < 4 else lo(4))
    x = dconv("bottleneck", x, lo(4), lo(4), lo(4))
    for i in range(4):
        l = 3 - i
        s = skips.pop()
        up = F.conv_transpose2d(x, r(sd[f"decoders.{i}.up.weight"], lo(l)), sd[f"decoders.{i}.up.bias"], stride=(1, 2))
        x = torch.cat([r(up, lo(l)), s], 1)
        ex = (not lo(l)) or (exact_last and l == 0)
        x = dconv(f"decoders.{i}.conv", x, not ex, not ex, not ex)
    return F.conv2d(x, sd["outputs.weight"], sd["outputs.bias"])


xe = torch.from_numpy(make_batch(n_eval, 5, 4096, seed=4321, unique=n_eval)[0])
with torch.no_grad():
    ref = O.unet_forward(sd, xe, train=False)
    full = forward(xe, exact_levels=(0, 1, 2, 3, 4))
    print("folded-BN fp32 vs module fp32: mask pixels differing", int(((full >= 0) != (ref >= 0)).sum()), flush=True)
    for name, kw in (("all bf16", {}), ("last DoubleConv exact", {"exact_last": True}), ("level 0 exact", {"exact_levels": (0,)}),
                     ("levels 0-1 exact", {"exact_levels": (0, 1)}), ("levels 0-2 exact", {"exact_levels": (0, 1, 2)})):
        z = forward(xe, **kw)
        flips = int(((torch.sigmoid(z) >= 0.5) != (torch.sigmoid(ref) >= 0.5)).sum())
        err = float((z - ref).abs().max() / ref.abs().max())
        print(f"{name:24s}: mask pixels differing {flips:5d} of {int((ref >= 0).sum())} positives, logits rel err {err:.2e}", flush=True)
