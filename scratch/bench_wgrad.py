import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd import hip
dev = torch.device('cuda:0')
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
n, h = int(sys.argv[3]) if len(sys.argv) > 3 else 256, int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = [(32, 32, 4096), (32, 64, 2048), (64, 64, 2048), (64, 128, 1024), (128, 128, 1024), (128, 256, 512), (256, 256, 512), (256, 512, 256), (512, 512, 256), (512, 256, 512), (64, 32, 4096)]
dtype = hip.F32 if len(sys.argv) > 2 and sys.argv[2] == 'fp32' else hip.BF16
peak = 2500.0 if dtype == hip.BF16 else 157.3
share = 1 - 2 / (3 * h)
for cin, cout, w in shapes:
    a = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, dev); a.data.normal_()
    dy = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev); dy.data.normal_()
    dw = torch.empty(cout, cin, 3, 5, device=dev)
    def run():
        hip.check(hip.lib.dn_conv2d_wgrad(dy.ref(cout), a.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, w, 3, 5, cin, 0, dtype, hip.stream_ptr()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    fl = 2.0 * n * h * w * cin * cout * 15
    print(f'wgrad {cin:4d}->{cout:4d} W={w:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s nominal, executed frac {fl*share/ms/1e9/peak:.3f}', flush=True)
    del a, dy
