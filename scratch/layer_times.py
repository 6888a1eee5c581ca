import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd.unet import HipUNet
dev = torch.device('cuda:0')
h = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = HipUNet(4, (3, 5), 32, dev, dtype='bf16'); m.eval()
x = torch.rand(256, 1, h, 4096, device=dev) * 1000
for _ in range(2): m.infer(x)
m.profile = []
names = [u.conv for u in m.conv_units if u.cin != 1]
for _ in range(3): m.infer(x)
torch.cuda.synchronize()
per = len(m.profile) // 3
agg = {}
for i, (tag, fl, e0, e1) in enumerate(m.profile):
    k = i % per
    d = agg.setdefault(k, [tag, fl, 0.0]); d[2] += e0.elapsed_time(e1) / 3
units = [u for u in m.conv_units if u.cin != 1]
tot = 0
for k in range(per):
    tag, fl, ms = agg[k]; u = units[k]; tot += ms
    print(f'{u.conv:38s} {u.cin:4d}->{u.cout:4d} W={4096 >> u.level:5d} {tag:34s} {ms*1e3:8.1f} us {fl/ms/1e9:8.1f} TF/s')
print('sum conv ms', tot)
