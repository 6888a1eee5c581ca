import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd.unet import HipUNet
dev = torch.device('cuda:0')
h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m = HipUNet(4, (3, 5), 32, dev, dtype='bf16'); m.eval()
x = torch.rand(256, 1, h, 4096, device=dev) * 1000
for _ in range(3): m.infer(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m.infer(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'{os.environ.get("TAG","default")}: bf16 H={h} N=256: {ms:.2f} ms/batch  {256/ms*1e3:.0f} profiles/s')
