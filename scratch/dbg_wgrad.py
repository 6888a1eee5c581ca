import sys; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from dnanet_amd import hip
dev = torch.device('cuda:0')
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def run(n, cin, cout, h, w, kh, kw, dtype):
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-3, 4, (n, cin, h, w), generator=g).float()
    dy = torch.randint(-3, 4, (n, cout, h, w), generator=g).float()
    wt = torch.zeros(cout, cin, kh, kw, requires_grad=True)
    F.conv2d(a, wt, padding=(kh // 2, kw // 2)).backward(dy)
    ref = wt.grad
    ab = hip.PlaneBuffer.from_nchw(a.to(dev), dtype); dyb = hip.PlaneBuffer.from_nchw(dy.to(dev), dtype)
    dw = torch.full((cout, cin, kh, kw), 7.0, device=dev)
    hip.check(hip.lib.dn_conv2d_wgrad(dyb.ref(cout), ab.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, w, kh, kw, cin, 0, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    out = dw.cpu()
    ok = torch.equal(out, ref)
    print(f'dtype {dtype} n{n} cin{cin} cout{cout} h{h} w{w} k{kh}{kw}: equal {ok}  maxdiff {float((out-ref).abs().max())} ref max {float(ref.abs().max())}')
    if not ok:
        print(out[0,0], ref[0,0])
for dtype in (0, 1):
    run(1, 8, 8, 1, 16, 1, 1, dtype)
    run(1, 8, 8, 5, 64, 3, 5, dtype)
    run(2, 32, 32, 5, 256, 3, 5, dtype)
    run(2, 64, 128, 6, 200, 3, 5, dtype)
    run(2, 16, 24, 5, 130, 1, 3, dtype)
