"""BatchNorm pass timings at the shapes of the shipped net: python scratch/bench_bn.py [rows=6]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd import hip
dev = torch.device('cuda:0'); dtype = hip.BF16; n = 256
h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for c, w in ((32, 4096), (128, 1024), (512, 256)):
    y = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev); y.data.normal_()
    g = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev); g.data.normal_()
    a = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev)
    cp = hip.channel_pad(c)
    v = [torch.ones(cp, device=dev) for _ in range(10)]
    part = torch.empty(hip.partial_floats(c), device=dev); coef = torch.empty(3 * cp, device=dev)
    tot = torch.empty(2 * cp, dtype=torch.float64, device=dev)
    E = n * h * w * c * 2
    us = timeit(lambda: hip.check(hip.lib.dn_bn_train_forward(y.ref(c), a.ref(c), n, h, w, *[hip.ptr(t) for t in v[:4]], 0.1, 1e-5, *[hip.ptr(t) for t in v[4:8]], hip.ptr(part), dtype, hip.stream_ptr())))
    print(f"C={c:4d} W={w:5d}: bn forward (stats+finalize+apply) {us:7.1f} us  {3 * E / us / 1e6:5.2f} TB/s")
    us = timeit(lambda: hip.check(hip.lib.dn_bn_train_backward_totals(g.ref(c), y.ref(c), hip.ptr(v[6]), hip.ptr(v[7]), n, h, w, hip.ptr(v[4]), hip.ptr(v[5]), hip.ptr(part), hip.ptr(tot), dtype, hip.stream_ptr())))
    print(f"              bn backward reduce (+totals)          {us:7.1f} us  {2 * E / us / 1e6:5.2f} TB/s")
    us = timeit(lambda: hip.check(hip.lib.dn_bn_train_backward(g.ref(c), y.ref(c), hip.ptr(v[6]), hip.ptr(v[7]), n, h, w, hip.ptr(v[0]), hip.ptr(v[4]), hip.ptr(v[5]), hip.ptr(v[8]), hip.ptr(v[9]), hip.ptr(coef), hip.ptr(part), dtype, hip.stream_ptr())))
    print(f"              bn backward (reduce+finalize+apply)   {us:7.1f} us  {5 * E / us / 1e6:5.2f} TB/s")
    us = timeit(lambda: hip.check(hip.lib.dn_channel_sum(g.ref(c), hip.ptr(v[8]), hip.ptr(part), n, h, w, dtype, hip.stream_ptr())))
    print(f"              channel sum                            {us:7.1f} us  {E / us / 1e6:5.2f} TB/s")
