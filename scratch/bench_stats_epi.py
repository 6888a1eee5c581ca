import os, sys, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd import hip
dev = torch.device('cuda:0')
n = 256; h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dtype = hip.BF16
for cin, cout, w in [(32, 32, 4096), (64, 64, 2048), (128, 128, 1024), (256, 256, 512), (512, 512, 256)]:
    x = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, dev); x.data.normal_()
    y = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev)
    wt = torch.randn(cout, cin, 3, 5, device=dev) * 0.05
    nb = hip.lib.dn_packed_weight_bytes(cout, cin, 3, 5, hip.PACK_FWD, dtype)
    pk = torch.empty(nb, dtype=torch.uint8, device=dev)
    hip.check(hip.lib.dn_pack_weight(hip.ptr(wt), None, hip.ptr(pk), cout, cin, 3, 5, hip.PACK_FWD, dtype, hip.stream_ptr()))
    b = torch.zeros(hip.channel_pad(cout), device=dev)
    cap = n * ((h + 2) // 3) * ((w + 127) // 128) * hip.channel_pad(cout) * 2
    st = torch.empty(cap, device=dev); nt = ctypes.c_int(0)
    def plain(): hip.check(hip.lib.dn_conv2d(x.ref(cin), hip.ptr(pk), hip.ptr(b), y.ref(cout), n, h, w, 3, 5, 0, dtype, hip.stream_ptr()))
    def stats(): hip.check(hip.lib.dn_conv2d_stats(x.ref(cin), hip.ptr(pk), hip.ptr(b), y.ref(cout), n, h, w, 3, 5, dtype, hip.ptr(st), cap, ctypes.byref(nt), hip.stream_ptr()))
    res = []
    for f in (plain, stats, plain, stats):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f'{cin:4d}->{cout:4d} W={w:5d}: plain {res[0]:7.1f} {res[2]:7.1f} us   stats {res[1]:7.1f} {res[3]:7.1f} us   (+{100*(res[3]/res[2]-1):.1f}%)')
