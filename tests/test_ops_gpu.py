"""Per-operator parity of the HIP kernels (through the C ABI) against plain PyTorch fp32 on the CPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _hip():
    from dnanet_amd import hip
    return hip


def _tol(dtype):
    # fp32 MFMA is an exact-fp32 fma chain; bf16 inputs carry 8 significant bits
    return (2e-5, 2e-5) if dtype == 0 else (2e-2, 2e-2)


def _q(ref, dtype):
    """What exact integer arithmetic gives once stored: bf16 mode rounds the fp32 accumulator (RNE)."""
    return ref.bfloat16().float() if dtype == 1 else ref


def _rand(shape, gen, ints=False):
    if ints:
        return torch.randint(-3, 4, shape, generator=gen).float()
    return torch.randn(shape, generator=gen)


def _pack(hip, w, scale, mode, dtype, dev):
    cout, cin, kh, kw = (w.shape[0], w.shape[1], w.shape[2], w.shape[3]) if mode in (0, 1) else \
        (w.shape[1], w.shape[0], 1, 2)
    nbytes = hip.lib.dn_packed_weight_bytes(cout, cin, kh, kw, mode, dtype)
    assert nbytes > 0
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    wd = w.contiguous().to(dev)
    sd = scale.to(dev) if scale is not None else None
    hip.check(hip.lib.dn_pack_weight(hip.ptr(wd), hip.ptr(sd), hip.ptr(packed), cout, cin, kh, kw, mode, dtype,
                                     hip.stream_ptr()))
    return packed


def _padvec(v, n, dev):
    out = torch.zeros(n, dtype=torch.float32, device=dev)
    out[:v.numel()] = v.to(dev)
    return out


CONV_CASES = [
    # n, cin, cout, h, w, kh, kw
    (2, 32, 32, 5, 256, 3, 5),
    (1, 64, 128, 5, 128, 3, 5),
    (2, 16, 8, 5, 64, 3, 5),
    (2, 32, 64, 6, 96, 3, 5),
    (1, 24, 40, 7, 200, 3, 5),
    (2, 8, 16, 5, 130, 1, 3),
    (1, 48, 64, 6, 64, 1, 3),
    (2, 32, 32, 5, 64, 1, 1),
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("ints", [True, False])
def test_conv2d(gpu_device, case, dtype, ints):
    hip = _hip()
    n, cin, cout, h, w, kh, kw = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + dtype)
    x = _rand((n, cin, h, w), g, ints)
    wt = _rand((cout, cin, kh, kw), g, ints) * (1.0 if ints else 0.1)
    b = _rand((cout,), g, ints)
    scale = None if ints else torch.rand(cout, generator=g) + 0.5
    ref = F.conv2d(x, wt * (scale.view(-1, 1, 1, 1) if scale is not None else 1.0), b, padding=(kh // 2, kw // 2)).relu()
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    packed = _pack(hip, wt, scale, hip.PACK_FWD, dtype, gpu_device)
    bias = _padvec(b, hip.channel_pad(cout), gpu_device)
    # write into the middle of a larger buffer to exercise plane offsets
    extra = 2
    yb = hip.PlaneBuffer(n, hip.planes(cout, dtype) + 2 * extra, h, w, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv2d(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), yb.ref(cout, extra), n, h, w, kh, kw, 1,
                                dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    out = yb.to_nchw(cout, extra).cpu()
    if ints:
        assert torch.equal(out, _q(ref, dtype))
    else:
        rtol, atol = _tol(dtype)
        assert torch.allclose(out, ref, rtol=rtol, atol=atol * float(ref.abs().max()))
    # neighbours untouched, padded channels zero
    assert float(yb.data[:, :extra].abs().max()) == 0 and float(yb.data[:, -extra:].abs().max()) == 0
    e = hip.epr(dtype)
    if cout % e:
        assert float(yb.data[:, extra + cout // e, :, :, cout % e:].abs().max()) == 0


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [(2, 32, 32, 5, 64, 3, 5), (1, 64, 32, 6, 96, 3, 5), (2, 16, 24, 5, 40, 1, 3)])
def test_conv2d_dgrad(gpu_device, case, dtype):
    hip = _hip()
    n, cin, cout, h, w, kh, kw = case
    g = torch.Generator().manual_seed(5)
    dy = _rand((n, cout, h, w), g, True)
    wt = _rand((cout, cin, kh, kw), g, True)
    ref = F.conv_transpose2d(dy, wt, padding=(kh // 2, kw // 2))     # = grad of conv2d wrt its input
    dyb = hip.PlaneBuffer.from_nchw(dy.to(gpu_device), dtype)
    packed = _pack(hip, wt, None, hip.PACK_DGRAD, dtype, gpu_device)
    dxb = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv2d(dyb.ref(cout), hip.ptr(packed), None, dxb.ref(cin), n, h, w, kh, kw, 0, dtype,
                                hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dxb.to_nchw(cin).cpu(), _q(ref, dtype))


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [(2, 64, 32, 5, 64), (1, 32, 16, 6, 48), (2, 128, 64, 5, 128), (1, 16, 8, 5, 32)])
def test_conv_transpose(gpu_device, case, dtype):
    hip = _hip()
    n, cin, cout, h, win = case
    g = torch.Generator().manual_seed(7)
    x = _rand((n, cin, h, win), g, True)
    wt = _rand((cin, cout, 1, 2), g, True)
    b = _rand((cout,), g, True)
    ref = F.conv_transpose2d(x, wt, b, stride=(1, 2))
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    packed = _pack(hip, wt, None, hip.PACK_CONVT_FWD, dtype, gpu_device)
    gs = hip.channel_pad(cout)
    bias2 = torch.zeros(2 * gs, device=gpu_device)
    bias2[:cout] = b.to(gpu_device)
    bias2[gs:gs + cout] = b.to(gpu_device)
    # output lands in the first half of a concat buffer (up first, skip second)
    p = hip.planes(cout, dtype)
    yb = hip.PlaneBuffer(n, 2 * p, h, 2 * win, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv_transpose_1x2(xb.ref(cin), hip.ptr(packed), hip.ptr(bias2), yb.ref(cout, 0), n, h, win, dtype,
                                            hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(yb.to_nchw(cout, 0).cpu(), _q(ref, dtype))
    assert float(yb.data[:, p:].abs().max()) == 0
    # data gradient
    dy = _rand((n, cout, h, 2 * win), g, True)
    refdx = F.conv2d(dy, wt.permute(0, 1, 2, 3).reshape(cin, cout, 1, 2), stride=(1, 2))
    dyb = hip.PlaneBuffer.from_nchw(dy.to(gpu_device), dtype)
    packed_d = _pack(hip, wt, None, hip.PACK_CONVT_DGRAD, dtype, gpu_device)
    dxb = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, win, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv_transpose_1x2_dgrad(dyb.ref(cout), hip.ptr(packed_d), dxb.ref(cin), n, h, win, dtype,
                                                  hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dxb.to_nchw(cin).cpu(), _q(refdx, dtype))


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("k", [(3, 5), (1, 3)])
def test_conv_first(gpu_device, dtype, k):
    hip = _hip()
    kh, kw = k
    n, h, w, cout = 2, 5, 300, 24
    g = torch.Generator().manual_seed(9)
    x = torch.randint(0, 16000, (n, 1, h, w), generator=g).float()
    wt = torch.randn(cout, 1, kh, kw, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2)).relu()
    cp = hip.channel_pad(cout)
    wf = torch.zeros(kh * kw, cp)
    wf[:, :cout] = wt.view(cout, kh * kw).t()
    yb = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, gpu_device, zero=True)
    xd, wd, bd = x.to(gpu_device), wf.to(gpu_device), _padvec(b, cp, gpu_device)   # keep alive across the launch
    hip.check(hip.lib.dn_conv_first(hip.ptr(xd), hip.ptr(wd), hip.ptr(bd),
                                    yb.ref(cout), n, h, w, cout, kh, kw, 1, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    out = yb.to_nchw(cout).cpu()
    rtol = 1e-5 if dtype == 0 else 1e-2
    assert torch.allclose(out, ref, rtol=rtol, atol=rtol * float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [0, 1])
def test_maxpool_and_head(gpu_device, dtype):
    hip = _hip()
    n, c, h, w = 2, 40, 5, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g)
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    xq = xb.to_nchw(c).cpu()                     # values as stored (bf16-rounded in bf16 mode)
    yb = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w // 2, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_maxpool_1x2(xb.ref(c), yb.ref(c), n, h, w // 2, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(yb.to_nchw(c).cpu(), F.max_pool2d(xq, (1, 2)))
    hw = torch.randn(1, c, 1, 1, generator=g)
    hb = torch.randn(1, generator=g)
    logits = torch.empty(n, 1, h, w, device=gpu_device)
    probs = torch.empty(n, 1, h, w, device=gpu_device)
    hwd, hbd = _padvec(hw.flatten(), hip.channel_pad(c), gpu_device), hb.to(gpu_device)
    hip.check(hip.lib.dn_head_1x1(xb.ref(c), hip.ptr(hwd), hip.ptr(hbd), hip.ptr(logits), hip.ptr(probs), n, h, w, dtype,
                                  hip.stream_ptr()))
    torch.cuda.synchronize()
    ref = F.conv2d(xq, hw, hb)
    assert torch.allclose(logits.cpu(), ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(probs.cpu(), torch.sigmoid(ref), rtol=1e-5, atol=1e-6)
