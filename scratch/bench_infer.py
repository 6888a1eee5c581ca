import sys, time; sys.path.insert(0, '.')
import torch
from dnanet_amd.unet import HipUNet
dev = torch.device('cuda:0')
FWD = 58_531_512_320
for dtype in ('bf16', 'fp32'):
    for h in (5, 6):
        n = 256 if dtype == 'bf16' else 64
        m = HipUNet(4, (3, 5), 32, dev, dtype=dtype); m.eval()
        x = torch.rand(n, 1, h, 4096, device=dev) * 1000
        for _ in range(2): m.infer(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps): m.infer(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        pps = n / ms * 1e3
        print(f'{dtype} H={h} N={n}: {ms:.2f} ms/batch  {pps:.0f} profiles/s  {pps*FWD*h/5/1e12:.1f} TFLOP/s')
