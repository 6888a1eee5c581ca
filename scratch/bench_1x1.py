import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd import hip
dev = torch.device('cuda:0')
n, h = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 6
dtype = hip.BF16
for cin, cout, win in [(512, 512, 256), (256, 256, 512), (64, 64, 2048)]:
    x = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, win, dtype, dev); x.data.normal_()
    y = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, win, dtype, dev)
    w = torch.randn(cout, cin, 1, 1, device=dev)
    nb = hip.lib.dn_packed_weight_bytes(cout, cin, 1, 1, hip.PACK_FWD, dtype)
    pk = torch.empty(nb, dtype=torch.uint8, device=dev)
    hip.check(hip.lib.dn_pack_weight(hip.ptr(w), None, hip.ptr(pk), cout, cin, 1, 1, hip.PACK_FWD, dtype, hip.stream_ptr()))
    b = torch.zeros(hip.channel_pad(cout), device=dev)
    def run():
        hip.check(hip.lib.dn_conv2d(x.ref(cin), hip.ptr(pk), hip.ptr(b), y.ref(cout, 0), n, h, win, 1, 1, 0, dtype, hip.stream_ptr()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = n * h * win * (cin + cout) * 2
    print(f'1x1 {cin:4d}->{cout:4d} W={win:5d}: {ms*1e3:8.1f} us  {byts/ms/1e6:8.1f} GB/s  {2.0*n*h*win*cin*cout/ms/1e9:7.1f} TF/s')
