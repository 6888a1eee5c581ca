import sys; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from dnanet_amd import hip
dev = torch.device('cuda:0')
from tests.test_ops_gpu import _pack, _padvec
def run(n, cin, cout, h, w, kh, kw, dtype, x, wt):
    xb = hip.PlaneBuffer.from_nchw(x.to(dev), dtype)
    packed = _pack(hip, wt, None, hip.PACK_FWD, dtype, dev)
    yb = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev, zero=True)
    hip.check(hip.lib.dn_conv2d(xb.ref(cin), hip.ptr(packed), None, yb.ref(cout), n, h, w, kh, kw, 0, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    return yb.to_nchw(cout).cpu()
for dtype in (0, 1):
    # 1x1 identity
    n, c, h, w = 1, 32, 5, 128
    x = torch.arange(n*c*h*w).float().view(n, c, h, w) % 97
    wt = torch.eye(c).view(c, c, 1, 1)
    out = run(n, c, c, h, w, 1, 1, dtype, x, wt)
    print('dtype', dtype, '1x1 identity equal:', torch.equal(out, x))
    if not torch.equal(out, x):
        d = (out != x).nonzero()
        print(' n mismatches', len(d), d[:5])
        print(out[0, :4, 0, :8]); print(x[0, :4, 0, :8])
    # 1x1 permutation weights (asymmetric)
    wt = torch.zeros(c, c, 1, 1); 
    for i in range(c): wt[i, (i * 5 + 3) % c, 0, 0] = i + 1
    out = run(n, c, c, h, w, 1, 1, dtype, x, wt); ref = F.conv2d(x, wt)
    print('dtype', dtype, '1x1 perm equal:', torch.equal(out, ref))
    # 3x5 single tap
    for tap in [(1, 2), (0, 0), (2, 4), (1, 3)]:
        wt = torch.zeros(c, c, 3, 5); 
        for i in range(c): wt[i, i, tap[0], tap[1]] = 1
        out = run(n, c, c, h, w, 3, 5, dtype, x, wt); ref = F.conv2d(x, wt, padding=(1, 2))
        print('dtype', dtype, 'tap', tap, 'equal:', torch.equal(out, ref), 'mismatch frac', float((out != ref).float().mean()))
