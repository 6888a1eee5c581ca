"""Transposed-convolution weight gradient, the four layers of the shipped net at N = 256: python scratch/bench_convt_wgrad.py [rows]
(DN_TUNE_NO_CONVT_GEMM=1: the round-2 kernel)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd import hip
dev = torch.device('cuda:0')
n = 256; h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for dtype in (hip.BF16, hip.F32):
    for cin, cout, win in [(512, 256, 256), (256, 128, 512), (128, 64, 1024), (64, 32, 2048)]:
        x = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, win, dtype, dev); x.data.normal_()
        dy = hip.PlaneBuffer(n, 2 * hip.planes(cout, dtype), h, 2 * win, dtype, dev); dy.data.normal_()
        dw = torch.empty(cin, cout, 1, 2, device=dev)
        ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        def run():
            hip.check(hip.lib.dn_conv_transpose_1x2_wgrad(dy.ref(cout, 0), x.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, win, dtype, hip.stream_ptr()))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        byts = n * h * win * (cin + 2 * cout) * (2 if dtype == hip.BF16 else 4)
        print(f'{"bf16" if dtype == hip.BF16 else "f32 "} convT wgrad {cin:4d}->{cout:4d} Win={win:5d}: {ms*1e3:8.1f} us  {byts/ms/1e6:8.1f} GB/s  {2.0*n*h*win*cin*cout*2/ms/1e9:7.1f} TF/s', flush=True)
        del x, dy, ws
