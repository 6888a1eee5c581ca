import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd import hip
dev = torch.device('cuda:0')
n = 256; h = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dtype = hip.F32 if len(sys.argv) > 2 and sys.argv[2] == 'fp32' else hip.BF16
esz = 4 if dtype == hip.F32 else 2
for cin, cout, win in [(512, 256, 256), (256, 128, 512), (128, 64, 1024), (64, 32, 2048)]:
    x = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, win, dtype, dev); x.data.normal_()
    y = hip.PlaneBuffer(n, 2 * hip.planes(cout, dtype), h, 2 * win, dtype, dev)
    w = torch.randn(cin, cout, 1, 2, device=dev)
    nb = hip.lib.dn_packed_weight_bytes(cout, cin, 1, 2, hip.PACK_CONVT_FWD, dtype)
    pk = torch.empty(nb, dtype=torch.uint8, device=dev)
    hip.check(hip.lib.dn_pack_weight(hip.ptr(w), None, hip.ptr(pk), cout, cin, 1, 2, hip.PACK_CONVT_FWD, dtype, hip.stream_ptr()))
    b2 = hip.up_bias2(torch.zeros(cout, device=dev), dtype)
    def run():
        hip.check(hip.lib.dn_conv_transpose_1x2(x.ref(cin), hip.ptr(pk), hip.ptr(b2), y.ref(cout, 0), n, h, win, dtype, hip.stream_ptr()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    pkd = torch.empty(hip.lib.dn_packed_weight_bytes(cout, cin, 1, 2, hip.PACK_CONVT_DGRAD, dtype), dtype=torch.uint8, device=dev)
    hip.check(hip.lib.dn_pack_weight(hip.ptr(w), None, hip.ptr(pkd), cout, cin, 1, 2, hip.PACK_CONVT_DGRAD, dtype, hip.stream_ptr()))
    def run2():
        hip.check(hip.lib.dn_conv_transpose_1x2_dgrad(y.ref(cout, 0), hip.ptr(pkd), x.ref(cin), n, h, win, dtype, hip.stream_ptr()))
    run2(); torch.cuda.synchronize(); e0.record()
    for _ in range(5): run2()
    e1.record(); torch.cuda.synchronize(); ms2 = e0.elapsed_time(e1) / 5
    byts = n * h * win * (cin + 2 * cout) * esz
    print(f'convT {cin:4d}->{cout:4d} Win={win:5d}: {ms*1e3:8.1f} us  {byts/ms/1e6:8.1f} GB/s  {2.0*n*h*win*cin*cout*2/ms/1e9:7.1f} TF/s | dgrad {ms2*1e3:8.1f} us', flush=True)
