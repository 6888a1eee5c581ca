"""One-off fuzz of the convolution entry points against torch on random shapes (integer data: exact)."""
import os, sys, ctypes, random; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dnanet_amd import hip
dev = torch.device('cuda:0')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150

def q(t, dtype): return t if dtype == hip.F32 else t.to(torch.bfloat16).float()
def pack(w, mode, dtype):
    cout, cin, kh, kw = w.shape
    nb = hip.lib.dn_packed_weight_bytes(cout, cin, kh, kw, mode, dtype)
    pk = torch.empty(nb, dtype=torch.uint8, device=dev); wd = w.contiguous().to(dev)
    hip.check(hip.lib.dn_pack_weight(hip.ptr(wd), None, hip.ptr(pk), cout, cin, kh, kw, mode, dtype, hip.stream_ptr()))
    torch.cuda.synchronize(); return pk
def padvec(v, n):
    out = torch.zeros(n, device=dev); out[:v.numel()] = v.to(dev); return out

bad = 0
for case in range(ncase):
    dtype = rng.choice([hip.F32, hip.BF16])
    kh, kw = rng.choice([(3, 5), (3, 5), (1, 3)])
    n = rng.choice([1, 2, 3, 5, 9, 33, 140])
    h = rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 10, 12])
    w = rng.choice([16, 32, 48, 128, 160, 256, 272, 512, 1024])
    cin = rng.choice([8, 16, 24, 32, 40, 64, 96, 128])
    cout = rng.choice([8, 16, 24, 32, 40, 64, 72, 128, 192])
    if n * h * w * cin * cout * kh * kw > 6e10: n = max(1, n // 16)
    g = torch.Generator().manual_seed(case)
    x = torch.randint(-2, 3, (n, cin, h, w), generator=g).float()
    wt = torch.randint(-1, 2, (cout, cin, kh, kw), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2))
    xb = hip.PlaneBuffer.from_nchw(x.to(dev), dtype)
    pk = pack(wt, hip.PACK_FWD, dtype)
    cp = hip.channel_pad(cout)
    bias = padvec(b, cp)
    mode = rng.choice(['plain', 'pool', 'stats', 'head', 'dgrad', 'wgrad'])
    ok = True
    tagbuf = ctypes.create_string_buffer(96)
    hip.lib.dn_conv2d_variant(cin, cout, n, h, w, kh, kw, dtype, tagbuf, 96)
    try:
        if mode == 'plain':
            yb = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev, zero=True)
            hip.check(hip.lib.dn_conv2d(xb.ref(cin), hip.ptr(pk), hip.ptr(bias), yb.ref(cout), n, h, w, kh, kw, 1, dtype, hip.stream_ptr()))
            ok = torch.equal(yb.to_nchw(cout).cpu(), q(ref.relu(), dtype))
        elif mode == 'pool':
            if w % 2: continue
            yb = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev, zero=True)
            pb = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w // 2, dtype, dev, zero=True)
            hip.check(hip.lib.dn_conv2d_pool(xb.ref(cin), hip.ptr(pk), hip.ptr(bias), yb.ref(cout), pb.ref(cout), n, h, w, kh, kw, 1, dtype, hip.stream_ptr()))
            want = q(ref.relu(), dtype)
            ok = torch.equal(yb.to_nchw(cout).cpu(), want) and torch.equal(pb.to_nchw(cout).cpu(), F.max_pool2d(want, (1, 2)))
        elif mode == 'stats':
            yb = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, dev, zero=True)
            cap = n * ((h + 2) // 3) * ((w + 127) // 128) * cp * 2
            st = torch.full((cap,), float('nan'), device=dev); nt = ctypes.c_int(0)
            hip.check(hip.lib.dn_conv2d_stats(xb.ref(cin), hip.ptr(pk), hip.ptr(bias), yb.ref(cout), n, h, w, kh, kw, dtype, hip.ptr(st), cap, ctypes.byref(nt), hip.stream_ptr()))
            y = yb.to_nchw(cout).cpu()
            part = st[:nt.value * cp * 2].view(nt.value, cp, 2).double().sum(0).cpu()
            ok = torch.equal(y, q(ref, dtype)) and torch.allclose(part[:cout, 0], y.double().sum((0, 2, 3)), rtol=1e-6, atol=0.5) \
                and torch.allclose(part[:cout, 1], (y.double() ** 2).sum((0, 2, 3)), rtol=1e-6, atol=0.5)
        elif mode == 'head':
            if not hip.lib.dn_conv2d_head_supported(cin, cout, n, h, w, kh, kw, dtype): continue
            hw = torch.randint(-2, 3, (cout,), generator=g).float(); hb = torch.tensor([0.5], device=dev)
            want = (q(ref.relu(), dtype) * hw.view(1, -1, 1, 1)).sum(1, keepdim=True) + 0.5
            lg = torch.zeros(n, 1, h, w, device=dev); hwp = padvec(hw, cp)
            hip.check(hip.lib.dn_conv2d_head(xb.ref(cin), hip.ptr(pk), hip.ptr(bias), cout, hip.ptr(hwp), hip.ptr(hb), hip.ptr(lg), None, n, h, w, kh, kw, 1, dtype, hip.stream_ptr()))
            ok = torch.equal(lg.cpu(), want)
        elif mode == 'dgrad':
            dy = torch.randint(-2, 3, (n, cout, h, w), generator=g).float()
            refd = F.conv_transpose2d(dy, wt, padding=(kh // 2, kw // 2))
            dyb = hip.PlaneBuffer.from_nchw(dy.to(dev), dtype)
            pkd = pack(wt, hip.PACK_DGRAD, dtype)
            dxb = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, dev, zero=True)
            hip.check(hip.lib.dn_conv2d(dyb.ref(cout), hip.ptr(pkd), None, dxb.ref(cin), n, h, w, kh, kw, 0, dtype, hip.stream_ptr()))
            ok = torch.equal(dxb.to_nchw(cin).cpu(), q(refd, dtype))
        else:
            dy = torch.randint(-2, 3, (n, cout, h, w), generator=g).float()
            wz = torch.zeros(cout, cin, kh, kw, requires_grad=True)
            F.conv2d(x, wz, padding=(kh // 2, kw // 2)).backward(dy)
            dyb = hip.PlaneBuffer.from_nchw(dy.to(dev), dtype)
            dw = torch.zeros(cout, cin, kh, kw, device=dev); ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
            hip.check(hip.lib.dn_conv2d_wgrad(dyb.ref(cout), xb.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, w, kh, kw, cin, 0, dtype, hip.stream_ptr()))
            ok = torch.equal(dw.cpu(), wz.grad)
        torch.cuda.synchronize()
    except RuntimeError as e:
        ok = False; print('ERROR', e)
    if not ok:
        bad += 1
        print('MISMATCH', mode, 'dtype', dtype, 'k', (kh, kw), 'n,h,w', (n, h, w), 'cin,cout', (cin, cout), tagbuf.value.decode())
print(f'{ncase} cases, {bad} mismatches')
