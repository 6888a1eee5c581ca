import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from conftest import load_golden
from dnanet_amd.unet import HipAdam, HipUNet
name = sys.argv[1] if len(sys.argv) > 1 else 'g_d2_k35_f8_h5_w64'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
dev = torch.device('cuda:0')
g = load_golden(name)
m = HipUNet(int(g["depth"]), tuple(int(v) for v in g["kernel"]), int(g["filters"]), dev, dtype=dtype)
m.load_state_dict(g["sd0"]); m.train()
x = torch.from_numpy(g["x"]).to(dev); t = torch.from_numpy(g["t"]).to(dev)
logits = m(x); sums = m.dice_forward(logits, t); m.backward()
print('loss', float(sums[4]), float(g['step0/loss']))
for k, gr in g["grad0"].items():
    scale = float(gr.abs().max()) + 1e-20
    e = float((m.grad(k).cpu() - gr).abs().max()) / scale
    print(f'{k:50s} {e:10.3e}  |ref|max {scale:.3e}')
