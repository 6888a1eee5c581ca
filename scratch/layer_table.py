"""Per-layer table of the shipped net at batch 256: forward (eval, folded BN), training forward (+ statistics epilogue), data
gradient and weight gradient of every 3x5 convolution, the transposed convolutions and the first convolution -- time,
TFLOP/s, fraction of the dense bf16 MFMA peak and algorithmic GB/s (one read of the input, one write of the output).
    python scratch/layer_table.py [rows=6] [dtype=bf16] [batch=256] > gpurun_out/layer_table.csv      (on the MI355X box)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from dnanet_amd import hip
dev = torch.device('cuda:0')
h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dname = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
dtype = hip.BF16 if dname == 'bf16' else hip.F32
peak = 2500.0 if dtype == hip.BF16 else 157.3
n, eb = (int(sys.argv[3]) if len(sys.argv) > 3 else 256), (2 if dtype == hip.BF16 else 4)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
stat = torch.empty(64 << 20, dtype=torch.float32, device=dev)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us

def pack(w, mode, kh, kw):
    cout, cin = (w.shape[0], w.shape[1]) if mode in (hip.PACK_FWD, hip.PACK_DGRAD) else (w.shape[1], w.shape[0])
    pk = torch.empty(hip.lib.dn_packed_weight_bytes(cout, cin, kh, kw, mode, dtype), dtype=torch.uint8, device=dev)
    hip.check(hip.lib.dn_pack_weight(hip.ptr(w), None, hip.ptr(pk), cout, cin, kh, kw, mode, dtype, hip.stream_ptr()))
    return pk

print("layer,op,cin,cout,W,us,TFLOP/s,frac_mfma_peak,algorithmic_GB/s,kernel")
def row(name, op, cin, cout, w, us, flop, byts, tag=""):
    print(f"{name},{op},{cin},{cout},{w},{us:.1f},{flop / us / 1e6:.1f},{flop / us / 1e6 / peak:.3f},{byts / us / 1e3:.0f},{tag}", flush=True)

f = 32
c = [f << l for l in range(5)]
W = [4096 >> l for l in range(5)]
layers = []
for l in range(4):
    layers += [(f"encoders.{l}.conv1", 1 if l == 0 else c[l - 1], c[l], W[l]), (f"encoders.{l}.conv2", c[l], c[l], W[l])]
layers += [("bottleneck.conv1", c[3], c[4], W[4]), ("bottleneck.conv2", c[4], c[4], W[4])]
for i in range(4):
    l = 3 - i
    layers += [(f"decoders.{i}.conv1", 2 * c[l], c[l], W[l]), (f"decoders.{i}.conv2", c[l], c[l], W[l])]

for name, cin, cout, w in layers:
    if cin == 1:
        x = torch.rand(n, 1, h, w, device=dev) * 1000
        wf = torch.randn(15, hip.channel_pad(cout), device=dev); b = torch.zeros(hip.channel_pad(cout), device=dev)
        y = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev)
        us = timeit(lambda: hip.check(hip.lib.dn_conv_first(hip.ptr(x), hip.ptr(wf), hip.ptr(b), y.ref(cout), n, h, w, cout, 3, 5, 1, dtype, hip.stream_ptr())))
        row(name, "fwd", cin, cout, w, us, 2.0 * n * h * w * cout * 15, n * h * w * (4 + cout * eb), "conv_first_mfma")
        nt = ctypes.c_int(0)
        us = timeit(lambda: hip.check(hip.lib.dn_conv_first_stats(hip.ptr(x), hip.ptr(wf), hip.ptr(b), y.ref(cout), n, h, w, cout, 3, 5, dtype, hip.ptr(stat), stat.numel(), ctypes.byref(nt), hip.stream_ptr())))
        row(name, "fwd+stats", cin, cout, w, us, 2.0 * n * h * w * cout * 15, n * h * w * (4 + cout * eb), "conv_first_mfma")
        part = torch.empty(hip.first_wgrad_partial_floats(cout, 3, 5), device=dev); dw = torch.empty(cout * 15, device=dev)
        y.data.normal_()
        us = timeit(lambda: hip.check(hip.lib.dn_conv_first_wgrad(hip.ptr(x), y.ref(cout), hip.ptr(dw), hip.ptr(part), part.numel(), n, h, w, 3, 5, dtype, hip.stream_ptr())))
        row(name, "wgrad", cin, cout, w, us, 2.0 * n * h * w * cout * 15, n * h * w * (4 + cout * eb), "conv_first_wgrad")
        continue
    a = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, dev); a.data.normal_()
    y = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev); y.data.normal_()
    wt = torch.randn(cout, cin, 3, 5, device=dev) * 0.05
    pf, pd = pack(wt, hip.PACK_FWD, 3, 5), pack(wt, hip.PACK_DGRAD, 3, 5)
    b = torch.zeros(hip.channel_pad(cout), device=dev)
    flop = 2.0 * n * h * w * cin * cout * 15
    byts = n * h * w * (cin + cout) * eb
    buf = ctypes.create_string_buffer(96)
    hip.check(hip.lib.dn_conv2d_variant(cin, cout, n, h, w, 3, 5, dtype, buf, len(buf))); tag = buf.value.decode().replace(",", ";")
    hip.check(hip.lib.dn_conv2d_variant(cout, cin, n, h, w, 3, 5, dtype, buf, len(buf))); tagd = buf.value.decode().replace(",", ";")
    us = timeit(lambda: hip.check(hip.lib.dn_conv2d(a.ref(cin), hip.ptr(pf), hip.ptr(b), y.ref(cout), n, h, w, 3, 5, 1, dtype, hip.stream_ptr())))
    row(name, "fwd", cin, cout, w, us, flop, byts, tag)
    nt = ctypes.c_int(0)
    us = timeit(lambda: hip.check(hip.lib.dn_conv2d_stats(a.ref(cin), hip.ptr(pf), hip.ptr(b), y.ref(cout), n, h, w, 3, 5, dtype, hip.ptr(stat), stat.numel(), ctypes.byref(nt), hip.stream_ptr())))
    row(name, "fwd+stats", cin, cout, w, us, flop, byts, tag)
    us = timeit(lambda: hip.check(hip.lib.dn_conv2d(y.ref(cout), hip.ptr(pd), None, a.ref(cin), n, h, w, 3, 5, 0, dtype, hip.stream_ptr())))
    row(name, "dgrad", cin, cout, w, us, flop, byts, tagd)
    a.data.normal_()
    dw = torch.empty(cout, cin, 3, 5, device=dev)
    us = timeit(lambda: hip.check(hip.lib.dn_conv2d_wgrad(y.ref(cout), a.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, w, 3, 5, cin, 0, dtype, hip.stream_ptr())))
    row(name, "wgrad", cin, cout, w, us, flop, byts, "conv_wgrad")

for i in range(4):
    l = 3 - i
    cin, cout, win = c[l + 1], c[l], W[l + 1]
    x = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, win, dtype, dev); x.data.normal_()
    y = hip.PlaneBuffer(n, 2 * hip.planes(cout, dtype), h, 2 * win, dtype, dev); y.data.normal_()
    wt = torch.randn(cin, cout, 1, 2, device=dev) * 0.05
    pf, pd = pack(wt, hip.PACK_CONVT_FWD, 1, 2), pack(wt, hip.PACK_CONVT_DGRAD, 1, 2)
    b2 = hip.up_bias2(torch.zeros(cout, device=dev), dtype)
    flop = 2.0 * n * h * win * cin * cout * 2
    byts = n * h * win * (cin + 2 * cout) * eb
    us = timeit(lambda: hip.check(hip.lib.dn_conv_transpose_1x2(x.ref(cin), hip.ptr(pf), hip.ptr(b2), y.ref(cout, 0), n, h, win, dtype, hip.stream_ptr())))
    row(f"decoders.{i}.up", "fwd", cin, cout, win, us, flop, byts, "conv_igemm 1x1")
    us = timeit(lambda: hip.check(hip.lib.dn_conv_transpose_1x2_dgrad(y.ref(cout, 0), hip.ptr(pd), x.ref(cin), n, h, win, dtype, hip.stream_ptr())))
    row(f"decoders.{i}.up", "dgrad", cin, cout, win, us, flop, byts, "conv_igemm 1x1")
    x.data.normal_()
    dw = torch.empty(cin, cout, 1, 2, device=dev)
    us = timeit(lambda: hip.check(hip.lib.dn_conv_transpose_1x2_wgrad(y.ref(cout, 0), x.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, win, dtype, hip.stream_ptr())))
    row(f"decoders.{i}.up", "wgrad", cin, cout, win, us, flop, byts, "conv_wgrad 1x1")
