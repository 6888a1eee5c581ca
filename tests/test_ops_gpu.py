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
    (2, 32, 32, 6, 256, 3, 5),         # narrow tile, one 6-row tile (conv_igemm<..,ht6,cot32>)
    (2, 128, 128, 4, 128, 1, 1),       # 1x1 GEMM with 2-row x 128-channel tiles
]

# large enough grids to be dispatched to the 8-wave pipelined kernels (api.hip choose_variant): name -> expected variant
CONV8_CASES = [
    ((128, 64, 64, 5, 1024, 3, 5), {0: "conv_igemm8<f32,3x5,ht5,cot64>", 1: "conv_igemm8<bf16,3x5,ht5,cot64>"}),
    ((64, 64, 64, 6, 1024, 3, 5), {0: "conv_igemm8c<f32,3x5,ht3,cot64>", 1: "conv_igemm<bf16,3x5,ht3,cot64>"}),
    ((130, 80, 128, 5, 512, 3, 5), {0: "conv_igemm8<f32,3x5,ht5,cot64>", 1: "conv_igemm8<bf16,3x5,ht5,cot64>"}),
]


def _variant(hip, cin, cout, n, h, w, kh, kw, dtype):
    import ctypes
    buf = ctypes.create_string_buffer(96)
    hip.check(hip.lib.dn_conv2d_variant(cin, cout, n, h, w, kh, kw, dtype, buf, len(buf)))
    return buf.value.decode()


def test_variant_names(gpu_device):
    hip = _hip()
    assert _variant(hip, 32, 32, 2, 6, 256, 3, 5, 1) == "conv_igemm<bf16,3x5,ht6,cot32>"
    assert _variant(hip, 128, 128, 2, 4, 128, 1, 1, 1) == "conv_igemm<bf16,1x1,ht2,cot128>"
    assert _variant(hip, 512, 512, 256, 5, 256, 3, 5, 1) == "conv_igemm8<bf16,3x5,ht5,cot64>"
    assert _variant(hip, 512, 512, 16, 5, 256, 3, 5, 1) == "conv_igemm<bf16,3x5,ht5,cot32>"     # small batch: narrow tiles
    assert _variant(hip, 512, 512, 256, 6, 256, 3, 5, 1) == "conv_igemm<bf16,3x5,ht3,cot64>"


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case,expect", CONV8_CASES)
def test_conv2d_8wave(gpu_device, case, expect, dtype):
    """Integer data: exact in both dtypes, so every tap, halo row, image border and pipeline stage must be right."""
    hip = _hip()
    n, cin, cout, h, w, kh, kw = case
    assert _variant(hip, cin, cout, n, h, w, kh, kw, dtype) == expect[dtype]
    g = torch.Generator().manual_seed(7 + dtype)
    x = torch.randint(-2, 3, (n, cin, h, w), generator=g).float()
    wt = torch.randint(-1, 2, (cout, cin, kh, kw), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2)).relu()
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    packed = _pack(hip, wt, None, hip.PACK_FWD, dtype, gpu_device)
    bias = _padvec(b, hip.channel_pad(cout), gpu_device)
    yb = hip.PlaneBuffer(n, hip.planes(cout, dtype) + 2, h, w, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv2d(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), yb.ref(cout, 1), n, h, w, kh, kw, 1, dtype,
                                hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(yb.to_nchw(cout, 1).cpu(), _q(ref, dtype))
    assert float(yb.data[:, :1].abs().max()) == 0 and float(yb.data[:, -1:].abs().max()) == 0


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
@pytest.mark.parametrize("case", [(2, 32, 32, 5, 64, 3, 5), (3, 16, 24, 6, 200, 3, 5), (130, 32, 64, 5, 512, 3, 5),
                                  (1, 32, 8, 3, 128, 3, 5), (2, 16, 24, 5, 130, 1, 3), (130, 32, 32, 6, 512, 3, 5)])
def test_conv2d_head(gpu_device, case, dtype):
    """dn_conv2d_head = dn_conv2d (ReLU) followed by dn_head_1x1, without the activation in between: logits on integer
    data are exact, probabilities are their sigmoid; shapes whose tile does not span the channels are refused."""
    hip = _hip()
    n, cin, cout, h, w, kh, kw = case
    g = torch.Generator().manual_seed(31 + dtype)
    x = torch.randint(-2, 3, (n, cin, h, w), generator=g).float()
    wt = torch.randint(-1, 2, (cout, cin, kh, kw), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    hw = torch.randint(-2, 3, (cout,), generator=g).float()
    hb = torch.tensor([0.5])
    act = _q(F.conv2d(x, wt, b, padding=(kh // 2, kw // 2)).relu(), dtype)
    want = (act * hw.view(1, -1, 1, 1)).sum(1, keepdim=True) + hb
    assert float(want.abs().max()) < 2 ** 23                       # integer sums stay exact in fp32
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    packed = _pack(hip, wt, None, hip.PACK_FWD, dtype, gpu_device)
    cp = hip.channel_pad(cout)
    bias, hwp, hbd = _padvec(b, cp, gpu_device), _padvec(hw, cp, gpu_device), hb.to(gpu_device)
    assert hip.lib.dn_conv2d_head_supported(cin, cout, n, h, w, kh, kw, dtype) == 1
    logits = torch.full((n, 1, h, w), float("nan"), device=gpu_device)
    probs = torch.full((n, 1, h, w), float("nan"), device=gpu_device)
    hip.check(hip.lib.dn_conv2d_head(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), cout, hip.ptr(hwp), hip.ptr(hbd),
                                     hip.ptr(logits), hip.ptr(probs), n, h, w, kh, kw, 1, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(logits.cpu(), want)
    assert torch.allclose(probs.cpu(), torch.sigmoid(want), rtol=1e-6, atol=1e-7)
    # logits only
    only = torch.zeros_like(logits)
    hip.check(hip.lib.dn_conv2d_head(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), cout, hip.ptr(hwp), hip.ptr(hbd),
                                     hip.ptr(only), None, n, h, w, kh, kw, 1, dtype, hip.stream_ptr()))
    assert torch.equal(only, logits)
    # 128 output channels never fit one tile
    assert hip.lib.dn_conv2d_head_supported(cin, 128, n, h, w, kh, kw, dtype) == 0
    assert hip.lib.dn_conv2d_head(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), 128, hip.ptr(hwp), hip.ptr(hbd),
                                  hip.ptr(only), None, n, h, w, kh, kw, 1, dtype, hip.stream_ptr()) == 2


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [(2, 32, 32, 5, 64, 3, 5), (3, 16, 24, 6, 200, 3, 5), (2, 64, 72, 5, 300, 3, 5),
                                  (1, 32, 128, 3, 128, 3, 5), (2, 16, 40, 5, 130, 1, 3), (130, 32, 32, 5, 512, 3, 5),
                                  (8, 64, 512, 5, 2048, 3, 5), (8, 64, 512, 6, 2048, 3, 5)])   # the last two: 8-wave kernels
def test_conv2d_pool(gpu_device, case, dtype):
    """dn_conv2d_pool: y as dn_conv2d writes it, plus MaxPool2d((1,2)) of y in a second buffer (integer data: exact);
    planes next to both outputs stay untouched."""
    hip = _hip()
    n, cin, cout, h, w, kh, kw = case
    g = torch.Generator().manual_seed(23 + dtype)
    x = torch.randint(-2, 3, (n, cin, h, w), generator=g).float()
    wt = torch.randint(-1, 2, (cout, cin, kh, kw), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2)).relu()
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    packed = _pack(hip, wt, None, hip.PACK_FWD, dtype, gpu_device)
    bias = _padvec(b, hip.channel_pad(cout), gpu_device)
    pl = hip.planes(cout, dtype)
    yb = hip.PlaneBuffer(n, pl + 2, h, w, dtype, gpu_device, zero=True)
    pb = hip.PlaneBuffer(n, pl + 2, h, w // 2, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv2d_pool(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), yb.ref(cout, 1), pb.ref(cout, 1), n, h, w,
                                     kh, kw, 1, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    want = _q(ref, dtype)
    assert torch.equal(yb.to_nchw(cout, 1).cpu(), want)
    assert torch.equal(pb.to_nchw(cout, 1).cpu(), F.max_pool2d(want, (1, 2)))
    for buf in (yb, pb):
        assert float(buf.data[:, :1].abs().max()) == 0 and float(buf.data[:, -1:].abs().max()) == 0
    e = hip.epr(dtype)
    if cout % e:
        assert float(pb.data[:, 1 + cout // e, :, :, cout % e:].abs().max()) == 0
    # odd width is refused
    assert hip.lib.dn_conv2d_pool(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), yb.ref(cout, 1), pb.ref(cout, 1), n, h, w - 1,
                                  kh, kw, 1, dtype, hip.stream_ptr()) != 0


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [(2, 32, 32, 5, 64, 3, 5), (3, 16, 24, 6, 200, 3, 5), (2, 64, 72, 5, 300, 3, 5),
                                  (1, 32, 128, 3, 128, 3, 5), (2, 16, 40, 5, 130, 1, 3), (130, 32, 32, 5, 512, 3, 5),
                                  (8, 64, 512, 5, 2048, 3, 5), (8, 64, 512, 6, 2048, 3, 5)])   # the last two: 8-wave kernels
def test_conv2d_stats(gpu_device, case, dtype):
    """dn_conv2d_stats: same y as dn_conv2d without ReLU, and per-tile (sum, sum of squares) of the *stored* values
    whose totals equal a separate pass over y (integer data: exact)."""
    import ctypes
    hip = _hip()
    n, cin, cout, h, w, kh, kw = case
    g = torch.Generator().manual_seed(11 + dtype)
    x = torch.randint(-2, 3, (n, cin, h, w), generator=g).float()
    wt = torch.randint(-1, 2, (cout, cin, kh, kw), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2))
    if cin == 64 and cout == 512:
        tag = _variant(hip, cin, cout, n, h, w, kh, kw, dtype)
        assert tag.startswith("conv_igemm8") == (h == 5 or dtype == 0), tag
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    packed = _pack(hip, wt, None, hip.PACK_FWD, dtype, gpu_device)
    cp = hip.channel_pad(cout)
    bias = _padvec(b, cp, gpu_device)
    yb = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, gpu_device, zero=True)
    cap = n * ((h + 2) // 3) * ((w + 127) // 128) * cp * 2
    stats = torch.full((cap,), float("nan"), device=gpu_device)
    ntiles = ctypes.c_int(0)
    hip.check(hip.lib.dn_conv2d_stats(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), yb.ref(cout), n, h, w, kh, kw, dtype,
                                      hip.ptr(stats), cap, ctypes.byref(ntiles), hip.stream_ptr()))
    torch.cuda.synchronize()
    y = yb.to_nchw(cout).cpu()
    assert torch.equal(y, _q(ref, dtype))
    t = ntiles.value
    assert 0 < t and t * cp * 2 <= cap
    part = stats[:t * cp * 2].view(t, cp, 2).double().sum(0).cpu()
    assert torch.equal(part[:cout, 0], y.double().sum((0, 2, 3)))
    assert torch.equal(part[:cout, 1], (y.double() ** 2).sum((0, 2, 3)))
    assert float(part[cout:].abs().max()) == 0 if cp > cout else True
    # too small a buffer is refused, not overrun
    assert hip.lib.dn_conv2d_stats(xb.ref(cin), hip.ptr(packed), hip.ptr(bias), yb.ref(cout), n, h, w, kh, kw, dtype,
                                   hip.ptr(stats), 16, ctypes.byref(ntiles), hip.stream_ptr()) != 0


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


BN_SUMS_CASES = [
    # n, cin (channels of dx = of the BatchNorm unit), cout (channels of dy), h, w -> the tile the dispatcher picks
    (2, 32, 32, 5, 64), (3, 24, 16, 6, 200), (2, 72, 64, 5, 300), (1, 128, 32, 3, 128), (2, 64, 64, 6, 130), (2, 32, 64, 6, 256),
    (130, 32, 32, 5, 512), (2, 8, 8, 7, 96), (2, 130, 70, 4, 33), (3, 5, 3, 7, 257), (1, 40, 200, 2, 129), (2, 3, 2, 1, 7),
    (8, 512, 64, 5, 2048), (8, 512, 64, 6, 2048),                  # 8-wave kernels (fp32: both; bf16: the 5-row one)
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", BN_SUMS_CASES)
def test_conv2d_dgrad_bn_sums(gpu_device, case, dtype):
    """dn_conv2d_dgrad_bn_sums: dx bitwise the plain data gradient's, and per-tile sums whose totals are BatchNorm
    backward's reduction over the stored dx (against fp64 on the CPU and against dn_bn_train_backward_totals); then
    dn_bn_train_backward_sums = dn_bn_train_backward on the same tensors."""
    import ctypes
    hip = _hip()
    dev = gpu_device
    n, cin, cout, h, w = case
    kh, kw = 3, 5
    g = torch.Generator().manual_seed(23 + dtype)
    dy = torch.randint(-1, 2, (n, cout, h, w), generator=g).float()
    wt = torch.randint(-1, 2, (cout, cin, kh, kw), generator=g).float()
    if cin == 512:
        tag = _variant(hip, cout, cin, n, h, w, kh, kw, dtype)
        assert tag.startswith("conv_igemm8") == (h == 5 or dtype == 0), tag
    dyb = hip.PlaneBuffer.from_nchw(dy.to(dev), dtype)
    packed = _pack(hip, wt, None, hip.PACK_DGRAD, dtype, dev)
    plain = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, dev, zero=True)
    hip.check(hip.lib.dn_conv2d(dyb.ref(cout), hip.ptr(packed), None, plain.ref(cin), n, h, w, kh, kw, 0, dtype, hip.stream_ptr()))
    # the unit the gradient feeds: values for which y*scale + shift and (y - mean)*invstd are exact in fp32 and never 0
    y = torch.randint(-4, 5, (n, cin, h, w), generator=g).float()
    yb = hip.PlaneBuffer.from_nchw(y.to(dev), dtype)
    cp = hip.channel_pad(cin)
    pick = lambda vals: torch.tensor(vals)[torch.randint(0, len(vals), (cin,), generator=g)]
    scale, shift = pick([-2.0, -1.0, -0.5, 0.5, 1.0, 2.0]), torch.randint(-3, 4, (cin,), generator=g).float() + 0.25
    mean, invstd = torch.randint(-4, 5, (cin,), generator=g).float() * 0.5, pick([0.5, 1.0, 2.0])
    vec = torch.full((4 * cp,), float("nan"), device=dev)           # the padding entries are never read into a result
    vs = [vec[i * cp:(i + 1) * cp] for i in range(4)]
    for dst, src in zip(vs, (scale, shift, mean, invstd)):
        dst[:cin] = src.to(dev)
    dxb = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, dev, zero=True)
    cap = n * ((h + 2) // 3) * ((w + 127) // 128) * cp * 2
    sums = torch.full((cap,), float("nan"), device=dev)
    ntiles = ctypes.c_int(0)
    args = (dyb.ref(cout), hip.ptr(packed), dxb.ref(cin), yb.ref(cin), hip.ptr(vs[0]), hip.ptr(vs[1]), hip.ptr(vs[2]), hip.ptr(vs[3]),
            n, h, w, kh, kw, dtype, hip.ptr(sums))
    hip.check(hip.lib.dn_conv2d_dgrad_bn_sums(*args, cap, ctypes.byref(ntiles), hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dxb.data, plain.data)
    dx = dxb.to_nchw(cin).cpu().double()
    t = ntiles.value
    assert 0 < t and t * cp * 2 <= cap
    got = sums[:t * cp * 2].view(t, cp, 2).double().sum(0).cpu()
    bc = lambda v: v.double().view(1, -1, 1, 1)
    dz = dx * ((y.double() * bc(scale) + bc(shift)) > 0)
    want = torch.stack([dz.sum((0, 2, 3)), (dz * (y.double() - bc(mean)) * bc(invstd)).sum((0, 2, 3))], 1)
    assert float(want.abs().max()) > 0
    assert torch.allclose(got[:cin], want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    # the two-pass path on the same tensors
    part = torch.empty(hip.partial_floats(cin), device=dev)
    tot_a, tot_b = torch.zeros(2 * cp, dtype=torch.float64, device=dev), torch.zeros(2 * cp, dtype=torch.float64, device=dev)
    hip.check(hip.lib.dn_bn_train_backward_totals(dxb.ref(cin), yb.ref(cin), hip.ptr(vs[0]), hip.ptr(vs[1]), n, h, w, hip.ptr(vs[2]),
                                                  hip.ptr(vs[3]), hip.ptr(part), hip.ptr(tot_a), dtype, hip.stream_ptr()))
    hip.check(hip.lib.dn_bn_backward_sums_totals(hip.ptr(sums), t, cin, hip.ptr(tot_b), hip.stream_ptr()))
    torch.cuda.synchronize()
    a, b = tot_a.view(cp, 2)[:cin].cpu(), tot_b.view(cp, 2)[:cin].cpu()
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-6 * float(a.abs().max()))
    assert float(tot_b.view(cp, 2)[cin:].abs().max()) == 0 if cp > cin else True
    gamma = (torch.rand(cin, generator=g) + 0.5).to(dev)
    outs = []
    for fused in (False, True):
        gb = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, w, dtype, dev, zero=True)
        gb.data.copy_(dxb.data)
        dgam, dbet, coef = torch.empty(cin, device=dev), torch.empty(cin, device=dev), torch.empty(3 * cp, device=dev)
        common = (gb.ref(cin), yb.ref(cin), hip.ptr(vs[0]), hip.ptr(vs[1]), n, h, w, hip.ptr(gamma), hip.ptr(vs[2]), hip.ptr(vs[3]),
                  hip.ptr(dgam), hip.ptr(dbet), hip.ptr(coef))
        if fused:
            hip.check(hip.lib.dn_bn_train_backward_sums(*common, hip.ptr(sums), t, dtype, hip.stream_ptr()))
        else:
            hip.check(hip.lib.dn_bn_train_backward(*common, hip.ptr(part), dtype, hip.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((gb.to_nchw(cin).cpu(), dgam.cpu(), dbet.cpu()))
    tol = 1e-5 if dtype == 0 else 1e-2
    for u, v in zip(*outs):
        assert torch.allclose(u, v, rtol=tol, atol=tol * float(v.abs().max()))
    # too small a buffer, a bias-carrying or non-3x5 call are refused, not overrun
    assert hip.lib.dn_conv2d_dgrad_bn_sums(*args, 16, ctypes.byref(ntiles), hip.stream_ptr()) != 0
    bad = list(args); bad[11], bad[12] = 1, 3
    assert hip.lib.dn_conv2d_dgrad_bn_sums(*bad, cap, ctypes.byref(ntiles), hip.stream_ptr()) != 0


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [(2, 64, 32, 5, 64), (1, 32, 16, 6, 48), (2, 128, 64, 5, 128), (1, 16, 8, 5, 32),
                                  (2, 48, 64, 6, 256), (2, 128, 40, 6, 256)])   # 128-row tiles, two K planes per barrier (odd / even plane counts)
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
    bias2 = hip.up_bias2(b.to(gpu_device), dtype)
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
    # the same with BatchNorm backward's sums of the unit whose activation the transposed convolution read
    import ctypes
    dev = gpu_device
    y = torch.randint(-4, 5, (n, cin, h, win), generator=g).float()
    yb2 = hip.PlaneBuffer.from_nchw(y.to(dev), dtype)
    cp = hip.channel_pad(cin)
    pick = lambda vals: torch.tensor(vals)[torch.randint(0, len(vals), (cin,), generator=g)]
    scale, shift = pick([-2.0, -1.0, -0.5, 0.5, 1.0, 2.0]), torch.randint(-3, 4, (cin,), generator=g).float() + 0.25
    mean, invstd = torch.randint(-4, 5, (cin,), generator=g).float() * 0.5, pick([0.5, 1.0, 2.0])
    vec = torch.full((4 * cp,), float("nan"), device=dev)
    vs = [vec[i * cp:(i + 1) * cp] for i in range(4)]
    for dst, src in zip(vs, (scale, shift, mean, invstd)):
        dst[:cin] = src.to(dev)
    cap = n * ((h * win + 127) // 128 + h) * cp * 2
    sums, nt = torch.full((cap,), float("nan"), device=dev), ctypes.c_int(0)
    dxs = hip.PlaneBuffer(n, hip.planes(cin, dtype), h, win, dtype, dev, zero=True)
    args = (dyb.ref(cout), hip.ptr(packed_d), dxs.ref(cin), yb2.ref(cin), hip.ptr(vs[0]), hip.ptr(vs[1]), hip.ptr(vs[2]), hip.ptr(vs[3]),
            n, h, win, dtype, hip.ptr(sums))
    hip.check(hip.lib.dn_conv_transpose_1x2_dgrad_bn_sums(*args, cap, ctypes.byref(nt), hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dxs.data, dxb.data) and 0 < nt.value and nt.value * cp * 2 <= cap
    got = sums[:nt.value * cp * 2].view(nt.value, cp, 2).double().sum(0).cpu()[:cin]
    bc = lambda v: v.double().view(1, -1, 1, 1)
    dz = dxs.to_nchw(cin).cpu().double() * ((y.double() * bc(scale) + bc(shift)) > 0)
    want = torch.stack([dz.sum((0, 2, 3)), (dz * (y.double() - bc(mean)) * bc(invstd)).sum((0, 2, 3))], 1)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    assert hip.lib.dn_conv_transpose_1x2_dgrad_bn_sums(*args, 16, ctypes.byref(nt), hip.stream_ptr()) != 0


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
@pytest.mark.parametrize("shape", [(2, 5, 300, 24), (3, 6, 256, 64), (2, 3, 128, 32), (2, 7, 200, 8), (1, 10, 4096, 32)])
@pytest.mark.parametrize("k", [(3, 5), (1, 3)])
def test_conv_first_exact_and_stats(gpu_device, dtype, shape, k):
    """The matrix-core first convolution on integer data (exact in fp32 accumulation; raw-RFU-sized inputs), every row
    tiling (5 / 6 / 3 / 1 rows), several channel tiles; dn_conv_first_stats gives the same y without ReLU plus per-tile
    sums whose totals equal a pass over the stored y."""
    import ctypes
    hip = _hip()
    kh, kw = k
    n, h, w, cout = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randint(0, 30000, (n, 1, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (cout, 1, kh, kw), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2))
    cp = hip.channel_pad(cout)
    wf = torch.zeros(kh * kw, cp)
    wf[:, :cout] = wt.view(cout, kh * kw).t()
    xd, wd, bd = x.to(gpu_device), wf.to(gpu_device), _padvec(b, cp, gpu_device)
    yb = hip.PlaneBuffer(n, hip.planes(cout, dtype) + 2, h, w, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv_first(hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), yb.ref(cout, 1), n, h, w, cout, kh, kw, 1, dtype,
                                    hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(yb.to_nchw(cout, 1).cpu(), _q(ref.relu(), dtype))
    assert float(yb.data[:, :1].abs().max()) == 0 and float(yb.data[:, -1:].abs().max()) == 0
    cap = n * h * ((w + 127) // 128) * cp * 2
    stats = torch.full((cap,), float("nan"), device=gpu_device)
    ntiles = ctypes.c_int(0)
    y2 = hip.PlaneBuffer(n, hip.planes(cout, dtype), h, w, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_conv_first_stats(hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), y2.ref(cout), n, h, w, cout, kh, kw, dtype,
                                          hip.ptr(stats), cap, ctypes.byref(ntiles), hip.stream_ptr()))
    torch.cuda.synchronize()
    y = y2.to_nchw(cout).cpu()
    assert torch.equal(y, _q(ref, dtype))
    t = ntiles.value
    assert 0 < t and t * cp * 2 <= cap
    part = stats[:t * cp * 2].view(t, cp, 2).double().sum(0).cpu()
    ysum, ysq = y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))
    # per-tile partials are fp32 sums of numbers up to ~1e11: equal to the exact totals to fp32 rounding per tile
    assert torch.allclose(part[:cout, 0], ysum, rtol=1e-6, atol=1.0) and torch.allclose(part[:cout, 1], ysq, rtol=1e-6, atol=1.0)
    assert hip.lib.dn_conv_first_stats(hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), y2.ref(cout), n, h, w, cout, kh, kw, dtype,
                                       hip.ptr(stats), 8, ctypes.byref(ntiles), hip.stream_ptr()) != 0


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


WGRAD_CASES = [(1, 8, 8, 1, 16, 1, 1), (1, 8, 8, 5, 64, 3, 5), (2, 32, 32, 5, 256, 3, 5), (2, 64, 128, 6, 200, 3, 5),
               (2, 16, 24, 5, 130, 1, 3), (3, 40, 72, 7, 96, 3, 5),
               # large enough for a split-K count that is a multiple of the row tiles: blocks of one image height, which
               # leave out the products with the zero halo row (3-, 5- and 1-row strips, two / two / seven tiles high)
               (16, 32, 64, 6, 512, 3, 5), (16, 32, 64, 10, 256, 3, 5), (8, 32, 64, 7, 512, 3, 5), (16, 32, 64, 5, 512, 3, 5)]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv2d_wgrad(gpu_device, case, dtype):
    """Split-K MFMA weight gradient (bf16: transposing LDS reads) is exact on small-integer data."""
    hip = _hip()
    n, cin, cout, h, w, kh, kw = case
    g = torch.Generator().manual_seed(1)
    a = _rand((n, cin, h, w), g, True)
    dy = _rand((n, cout, h, w), g, True)
    wt = torch.zeros(cout, cin, kh, kw, requires_grad=True)
    F.conv2d(a, wt, padding=(kh // 2, kw // 2)).backward(dy)
    ab = hip.PlaneBuffer.from_nchw(a.to(gpu_device), dtype)
    dyb = hip.PlaneBuffer.from_nchw(dy.to(gpu_device), dtype)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=gpu_device)
    dw = torch.full((cout, cin, kh, kw), 7.0, device=gpu_device)
    hip.check(hip.lib.dn_conv2d_wgrad(dyb.ref(cout), ab.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, w, kh, kw, cin,
                                      0, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu(), wt.grad)


@pytest.mark.parametrize("dtype", [0, 1])
def test_conv_transpose_wgrad_and_bias(gpu_device, dtype):
    hip = _hip()
    n, cin, cout, h, win = 2, 64, 32, 5, 96
    g = torch.Generator().manual_seed(2)
    x = _rand((n, cin, h, win), g, True)
    dy = _rand((n, cout, h, 2 * win), g, True)
    wt = torch.zeros(cin, cout, 1, 2, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    F.conv_transpose2d(x, wt, b, stride=(1, 2)).backward(dy)
    xb = hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype)
    dyb = hip.PlaneBuffer.from_nchw(dy.to(gpu_device), dtype)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=gpu_device)
    dw = torch.full((cin, cout, 1, 2), 7.0, device=gpu_device)
    hip.check(hip.lib.dn_conv_transpose_1x2_wgrad(dyb.ref(cout), xb.ref(cin), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, win,
                                                  dtype, hip.stream_ptr()))
    db = torch.empty(cout, device=gpu_device)
    part = torch.empty(hip.partial_floats(cout), device=gpu_device)
    hip.check(hip.lib.dn_channel_sum(dyb.ref(cout), hip.ptr(db), hip.ptr(part), n, h, 2 * win, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu(), wt.grad)
    assert torch.equal(db.cpu(), b.grad)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [(3, 64, 32, 5, 128), (2, 128, 64, 6, 64), (2, 512, 256, 4, 32), (2, 272, 136, 4, 64),
                                  (1, 40, 24, 8, 64), (9, 256, 128, 6, 64), (5, 32, 16, 2, 2048)])
def test_conv_transpose_wgrad_large_tile_gemm(gpu_device, dtype, case):
    """The split-K GEMM of convt_wgrad.h (taken when H * Win is whole 64-pixel strips): every tile size (64, 128, 256 rows /
    columns), partially filled tiles, more splits than strips, operands that are plane ranges inside wider buffers -- exact on
    small-integer data against autograd, and bitwise equal to the 1x1 kernel it replaces (DN_TUNE switch not needed: the
    sums are integers)."""
    hip = _hip()
    n, cin, cout, h, win = case
    g = torch.Generator().manual_seed(7)
    x = _rand((n, cin, h, win), g, True)
    dy = _rand((n, cout, h, 2 * win), g, True)
    wt = torch.zeros(cin, cout, 1, 2, requires_grad=True)
    F.conv_transpose2d(x, wt, None, stride=(1, 2)).backward(dy)
    # x sits behind 3 foreign planes, dY in the first half of a concat-gradient buffer (as HipUNet.backward passes them)
    px, py = hip.planes(cin, dtype), hip.planes(cout, dtype)
    xb = hip.PlaneBuffer(n, px + 3, h, win, dtype, gpu_device)
    xb.data.fill_(3.0)
    xb.data[:, 3:].copy_(hip.PlaneBuffer.from_nchw(x.to(gpu_device), dtype).data)
    dyb = hip.PlaneBuffer(n, 2 * py, h, 2 * win, dtype, gpu_device)
    dyb.data.fill_(5.0)
    dyb.data[:, :py].copy_(hip.PlaneBuffer.from_nchw(dy.to(gpu_device), dtype).data)
    ws = torch.empty(160 << 20, dtype=torch.uint8, device=gpu_device)
    dw = torch.full((cin, cout, 1, 2), 7.0, device=gpu_device)
    hip.check(hip.lib.dn_conv_transpose_1x2_wgrad(dyb.ref(cout, 0), xb.ref(cin, 3), hip.ptr(dw), hip.ptr(ws), ws.numel(), n, h, win,
                                                  dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu(), wt.grad)


@pytest.mark.parametrize("dtype", [0, 1])
def test_batchnorm_train_forward_backward(gpu_device, dtype):
    hip = _hip()
    n, c, h, w = 3, 24, 5, 64
    g = torch.Generator().manual_seed(4)
    y = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    yb = hip.PlaneBuffer.from_nchw(y.to(gpu_device), dtype)
    yq = yb.to_nchw(c).cpu().requires_grad_(True)       # the values the kernel sees
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(c, generator=g).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    a_ref = F.relu(F.batch_norm(yq, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5))
    cp = hip.channel_pad(c)
    dev = gpu_device
    gd, bd, rmd, rvd = (t.detach().to(dev) for t in (gamma, beta, rm, rv))
    vec = torch.zeros(4 * cp, device=dev)
    mean, invstd, scale, shift = (vec[i * cp:(i + 1) * cp] for i in range(4))
    part = torch.empty(hip.partial_floats(c), device=dev)
    ab = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
    hip.check(hip.lib.dn_bn_train_forward(yb.ref(c), ab.ref(c), n, h, w, hip.ptr(gd), hip.ptr(bd), hip.ptr(rmd), hip.ptr(rvd), 0.1,
                                          1e-5, hip.ptr(mean), hip.ptr(invstd), hip.ptr(scale), hip.ptr(shift), hip.ptr(part),
                                          dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == 0 else 1e-2
    assert torch.allclose(ab.to_nchw(c).cpu(), a_ref.detach(), rtol=tol, atol=tol)
    assert torch.allclose(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    # backward: gradient wrt y, gamma, beta
    gout = torch.randn(n, c, h, w, generator=g)
    gb = hip.PlaneBuffer.from_nchw(gout.to(dev), dtype)
    gq = gb.to_nchw(c).cpu()
    # use the kernel's own (stored) activation for the ReLU mask, as the library does
    a_store = ab.to_nchw(c).cpu()
    (a_ref * (a_store > 0).float()).backward(gq) if False else a_ref.backward(gq * ((a_store > 0) == (a_ref > 0)).float())
    dgam, dbet = torch.empty(c, device=dev), torch.empty(c, device=dev)
    coef = torch.empty(3 * cp, device=dev)
    hip.check(hip.lib.dn_bn_train_backward(gb.ref(c), yb.ref(c), hip.ptr(scale), hip.ptr(shift), n, h, w, hip.ptr(gd),
                                           hip.ptr(mean), hip.ptr(invstd),
                                           hip.ptr(dgam), hip.ptr(dbet), hip.ptr(coef), hip.ptr(part), dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    gt = 2e-5 if dtype == 0 else 2e-2
    assert torch.allclose(dgam.cpu(), gamma.grad, rtol=gt, atol=gt * float(gamma.grad.abs().max()))
    assert torch.allclose(dbet.cpu(), beta.grad, rtol=gt, atol=gt * float(beta.grad.abs().max()))
    assert torch.allclose(gb.to_nchw(c).cpu(), yq.grad, rtol=gt, atol=gt * float(yq.grad.abs().max()))


@pytest.mark.parametrize("dtype", [0, 1])
def test_bn_apply_from_tile_statistics_with_pool(gpu_device, dtype):
    """dn_bn_train_apply / dn_bn_train_apply_pool from per-tile (sum, sum of squares) partials: the activation equals
    dn_bn_train_forward's bit for bit when the partials add up to the same statistics, and the pooled output is
    MaxPool2d((1,2)) of that activation."""
    hip = _hip()
    n, c, h, w = 3, 40, 5, 96
    dev = gpu_device
    g = torch.Generator().manual_seed(6)
    y = torch.randint(-8, 9, (n, c, h, w), generator=g).float()       # integers: every partition of the sums is exact
    yb = hip.PlaneBuffer.from_nchw(y.to(dev), dtype)
    cp = hip.channel_pad(c)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)

    def run(fn, *extra):
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        vec = torch.zeros(4 * cp, device=dev)
        ab = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
        args = [yb.ref(c), ab.ref(c)] + ([extra[0]] if extra[0] is not None else []) + [n, h, w, hip.ptr(gd), hip.ptr(bd), hip.ptr(rm), hip.ptr(rv), 0.1, 1e-5] + \
               [hip.ptr(vec[i * cp:(i + 1) * cp]) for i in range(4)] + list(extra[1:]) + [dtype, hip.stream_ptr()]
        hip.check(fn(*args))
        torch.cuda.synchronize()
        return ab.to_nchw(c).cpu(), rm.cpu(), rv.cpu()

    gd, bd = gamma.to(dev), beta.to(dev)
    part = torch.empty(hip.partial_floats(c), device=dev)
    a_ref, rm_ref, rv_ref = run(hip.lib.dn_bn_train_forward, None, hip.ptr(part))
    # two "tiles" per image row pair: sums over disjoint pixel sets, laid out [tile][cpad][2]
    tiles = []
    for img in range(n):
        for half in (slice(0, w // 2), slice(w // 2, w)):
            blk = y[img, :, :, half].double()
            t = torch.zeros(cp, 2)
            t[:c, 0], t[:c, 1] = blk.sum((1, 2)), (blk ** 2).sum((1, 2))
            tiles.append(t)
    tile_part = torch.stack(tiles).float().to(dev)
    a1, rm1, rv1 = run(hip.lib.dn_bn_train_apply, None, hip.ptr(tile_part), len(tiles))
    assert torch.equal(a1, a_ref) and torch.equal(rm1, rm_ref) and torch.equal(rv1, rv_ref)
    pb = hip.PlaneBuffer(n, hip.planes(c, dtype) + 2, h, w // 2, dtype, dev, zero=True)
    a2, _, _ = run(hip.lib.dn_bn_train_apply_pool, pb.ref(c, 1), hip.ptr(tile_part), len(tiles))
    assert torch.equal(a2, a_ref)
    assert torch.equal(pb.to_nchw(c, 1).cpu(), F.max_pool2d(a_ref, (1, 2)))
    assert float(pb.data[:, :1].abs().max()) == 0 and float(pb.data[:, -1:].abs().max()) == 0


@pytest.mark.parametrize("dtype", [0, 1])
def test_maxpool_backward_with_skip(gpu_device, dtype):
    hip = _hip()
    n, c, h, w = 2, 24, 5, 64
    g = torch.Generator().manual_seed(6)
    x = torch.randint(-2, 3, (n, c, h, w), generator=g).float()      # many ties: first element must win
    dpool = torch.randn(n, c, h, w // 2, generator=g)
    dskip = torch.randn(n, c, h, w, generator=g)
    xb, dpb, dsb = (hip.PlaneBuffer.from_nchw(t.to(gpu_device), dtype) for t in (x, dpool, dskip))
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr, (1, 2)).backward(dpb.to_nchw(c).cpu())
    ref = xr.grad + dsb.to_nchw(c).cpu()
    gb = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, gpu_device, zero=True)
    hip.check(hip.lib.dn_maxpool_1x2_backward(xb.ref(c), dpb.ref(c), dsb.ref(c), gb.ref(c), n, h, w // 2, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(gb.to_nchw(c).cpu(), _q(ref, dtype))


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("shape", [(2, 24, 5, 64), (3, 32, 6, 200), (40, 64, 5, 512), (1, 8, 3, 34)])
def test_pool_and_head_backward_with_bn_sums(gpu_device, dtype, shape):
    """dn_maxpool_1x2_backward_bn_sums / dn_head_1x1_backward_bn_sums: g (and dw, db) bitwise those of the plain entry points,
    and block sums whose totals are BatchNorm backward's reduction over the stored g: dn_bn_train_backward_sums on them gives
    what dn_bn_train_backward gives on the same tensors."""
    import ctypes
    hip = _hip()
    dev = gpu_device
    n, c, h, w = shape
    gen = torch.Generator().manual_seed(31 + dtype)
    cp = hip.channel_pad(c)
    # the BatchNorm unit whose activation gradient the producers write
    y = torch.randn(n, c, h, w, generator=gen) * 2 + 0.5
    yb = hip.PlaneBuffer.from_nchw(y.to(dev), dtype)
    yq = yb.to_nchw(c).cpu().double()
    vec = torch.full((4 * cp,), float("nan"), device=dev)
    vs = [vec[i * cp:(i + 1) * cp] for i in range(4)]
    scale, shift = torch.randn(c, generator=gen), torch.randn(c, generator=gen)
    mean, invstd = torch.randn(c, generator=gen), torch.rand(c, generator=gen) + 0.5
    for dst, src in zip(vs, (scale, shift, mean, invstd)):
        dst[:c] = src.to(dev)
    bn = (yb.ref(c), hip.ptr(vs[0]), hip.ptr(vs[1]), hip.ptr(vs[2]), hip.ptr(vs[3]))
    cap = 512 * cp * 2
    part = torch.empty(max(hip.partial_floats(c), 4096), device=dev)
    bc = lambda v: v.double().view(1, -1, 1, 1)

    def check_sums(gbuf, sums, nblk):
        gq = gbuf.to_nchw(c).cpu().double()
        t = yq * bc(scale) + bc(shift)
        safe = (t.abs() > 1e-4).double()        # elements whose sign cannot depend on the rounding of the fp32 fma
        dz = gq * (t > 0)
        xhat = (yq - bc(mean)) * bc(invstd)
        want = torch.stack([dz.sum((0, 2, 3)), (dz * xhat).sum((0, 2, 3))], 1)
        got = sums[:nblk * cp * 2].view(nblk, cp, 2).double().sum(0).cpu()[:c]
        slack = float(((1 - safe) * gq.abs() * (1 + xhat.abs())).sum((0, 2, 3)).max())
        assert torch.allclose(got, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()) + slack), float((got - want).abs().max())
        gamma = (torch.rand(c, generator=gen) + 0.5).to(dev)
        outs = []
        for fused in (False, True):
            gb2 = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
            gb2.data.copy_(gbuf.data)
            dgam, dbet, coef = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(3 * cp, device=dev)
            common = (gb2.ref(c), yb.ref(c), hip.ptr(vs[0]), hip.ptr(vs[1]), n, h, w, hip.ptr(gamma), hip.ptr(vs[2]), hip.ptr(vs[3]),
                      hip.ptr(dgam), hip.ptr(dbet), hip.ptr(coef))
            if fused:
                hip.check(hip.lib.dn_bn_train_backward_sums(*common, hip.ptr(sums), nblk, dtype, hip.stream_ptr()))
            else:
                hip.check(hip.lib.dn_bn_train_backward(*common, hip.ptr(part), dtype, hip.stream_ptr()))
            torch.cuda.synchronize()
            outs.append((gb2.to_nchw(c).cpu(), dgam.cpu(), dbet.cpu()))
        tol = 1e-5 if dtype == 0 else 1e-2
        for u, v in zip(*outs):
            assert torch.allclose(u, v, rtol=tol, atol=tol * float(v.abs().max()))

    # ---- pooling backward + skip gradient
    if w % 2 == 0:
        x = torch.randint(-2, 3, (n, c, h, w), generator=gen).float()
        dpool, dskip = torch.randn(n, c, h, w // 2, generator=gen), torch.randn(n, c, h, w, generator=gen)
        xb, dpb, dsb = (hip.PlaneBuffer.from_nchw(t.to(dev), dtype) for t in (x, dpool, dskip))
        for with_skip in (True, False):
            ds = dsb.ref(c) if with_skip else hip.DnTensor(None, 0, 0, c)
            plain = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
            hip.check(hip.lib.dn_maxpool_1x2_backward(xb.ref(c), dpb.ref(c), ds, plain.ref(c), n, h, w // 2, dtype, hip.stream_ptr()))
            gb = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
            sums, nblk = torch.full((cap,), float("nan"), device=dev), ctypes.c_int(0)
            hip.check(hip.lib.dn_maxpool_1x2_backward_bn_sums(xb.ref(c), dpb.ref(c), ds, gb.ref(c), *bn, n, h, w // 2, dtype, hip.ptr(sums),
                                                              cap, ctypes.byref(nblk), hip.stream_ptr()))
            torch.cuda.synchronize()
            assert torch.equal(gb.data, plain.data) and 0 < nblk.value <= 512
            check_sums(gb, sums, nblk.value)
        assert hip.lib.dn_maxpool_1x2_backward_bn_sums(xb.ref(c), dpb.ref(c), dsb.ref(c), gb.ref(c), *bn, n, h, w // 2, dtype, hip.ptr(sums),
                                                       8, ctypes.byref(nblk), hip.stream_ptr()) != 0
    # ---- head backward
    dlog = torch.randn(n, 1, h, w, generator=gen).to(dev)
    act = hip.PlaneBuffer.from_nchw(torch.randn(n, c, h, w, generator=gen).to(dev), dtype)
    hw_ = _padvec(torch.randn(c, generator=gen), cp, dev)
    res = []
    for fused in (False, True):
        gb = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
        dw, db = torch.zeros(c, device=dev), torch.zeros(1, device=dev)
        sums, nblk = torch.full((cap,), float("nan"), device=dev), ctypes.c_int(0)
        if fused:
            hip.check(hip.lib.dn_head_1x1_backward_bn_sums(hip.ptr(dlog), act.ref(c), hip.ptr(hw_), gb.ref(c), hip.ptr(dw), hip.ptr(db),
                                                           hip.ptr(part), *bn, n, h, w, dtype, hip.ptr(sums), cap, ctypes.byref(nblk),
                                                           hip.stream_ptr()))
        else:
            hip.check(hip.lib.dn_head_1x1_backward(hip.ptr(dlog), act.ref(c), hip.ptr(hw_), gb.ref(c), hip.ptr(dw), hip.ptr(db),
                                                   hip.ptr(part), n, h, w, dtype, hip.stream_ptr()))
        torch.cuda.synchronize()
        res.append((gb, dw.clone(), db.clone(), sums, nblk.value))
    assert torch.equal(res[0][0].data, res[1][0].data) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    check_sums(res[1][0], res[1][3], res[1][4])


def test_dice_and_adam(gpu_device):
    hip = _hip()
    g = torch.Generator().manual_seed(8)
    z = torch.randn(3, 1, 5, 333, generator=g) * 3
    t = (torch.rand(3, 1, 5, 333, generator=g) < 0.05).to(torch.int8)
    zr = z.clone().requires_grad_(True)
    p = torch.sigmoid(zr).view(-1)
    loss = 1 - (2 * (p * t.view(-1)).sum() + 1) / (p.sum() + t.sum() + 1)
    loss.backward()
    zd, td = z.to(gpu_device), t.to(gpu_device)
    sums = torch.zeros(8, device=gpu_device)
    part = torch.empty(8192, device=gpu_device)
    dz = torch.empty_like(zd)
    hip.check(hip.lib.dn_dice_forward(hip.ptr(zd), hip.ptr(td), zd.numel(), hip.ptr(sums), hip.ptr(part), hip.stream_ptr()))
    hip.check(hip.lib.dn_dice_backward(hip.ptr(zd), hip.ptr(td), zd.numel(), hip.ptr(sums), hip.ptr(dz), hip.stream_ptr()))
    torch.cuda.synchronize()
    assert abs(float(sums[4]) - float(loss)) < 1e-6
    assert int(sums[3]) == int(((torch.sigmoid(z) >= 0.5) == (t != 0)).sum())
    assert torch.allclose(dz.cpu(), zr.grad, rtol=1e-4, atol=1e-9)
    # Adam with coupled weight decay against torch.optim.Adam
    n = 1000
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-2, weight_decay=5e-4)
    pd, md, vd = p0.to(gpu_device), torch.zeros(n, device=gpu_device), torch.zeros(n, device=gpu_device)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        gd = gr.to(gpu_device)
        hip.check(hip.lib.dn_adam_step(hip.ptr(pd), hip.ptr(gd), hip.ptr(md), hip.ptr(vd), n, 1e-2, 0.9, 0.999, 1e-8, 5e-4, step,
                                       1.0, 1, hip.stream_ptr()))
        torch.cuda.synchronize()
        assert float(gd.abs().max()) == 0.0
    assert torch.allclose(pd.cpu(), ref.data, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [0, 1])
def test_pack_batched_equals_single_packs(gpu_device, dtype):
    """dn_pack_batched (one launch for a whole job table) writes the images dn_pack_weight writes one by one, and the
    fp32 side images (padded vector, doubled up-conv bias, tap-major first-conv weight)."""
    hip = _hip()
    g = torch.Generator().manual_seed(3)
    w1 = torch.randn(40, 24, 3, 5, generator=g).to(gpu_device)
    w2 = torch.randn(64, 32, 1, 2, generator=g).to(gpu_device)          # ConvTranspose2d layout [cin][cout][1][2]
    b = torch.randn(40, generator=g).to(gpu_device)
    wf = torch.randn(24, 1, 3, 5, generator=g).to(gpu_device)
    specs = [(w1, 40, 24, 15, hip.PACK_FWD), (w1, 40, 24, 15, hip.PACK_DGRAD), (w2, 32, 64, 2, hip.PACK_CONVT_FWD),
             (w2, 32, 64, 2, hip.PACK_CONVT_DGRAD), (b, 40, 1, 1, hip.PACK_VEC), (b, 40, 1, 1, hip.PACK_UP_BIAS),
             (wf, 24, 1, 15, hip.PACK_FIRST)]
    outs, jobs = [], []
    for w, cout, cin, taps, mode in specs:
        if mode <= hip.PACK_CONVT_DGRAD:
            kh, kw = (3, 5) if taps == 15 else (1, 2)
            n = hip.lib.dn_packed_weight_bytes(cout, cin, kh, kw, mode, dtype)
        else:
            cp = hip.channel_pad(cout)
            n = 4 * {hip.PACK_VEC: cp, hip.PACK_UP_BIAS: 2 * cp, hip.PACK_FIRST: taps * cp}[mode]
        out = torch.full((n + 64,), 0xAB, dtype=torch.uint8, device=gpu_device)      # guard bytes behind every image
        outs.append((out, n))
        jobs.append(hip.DnPackJob(hip.ptr(w), None, hip.ptr(out), cout, cin, taps, mode, 0, 0, 0, 0))
    table = (hip.DnPackJob * len(jobs))(*jobs)
    blocks = hip.lib.dn_pack_jobs_prepare(table, len(jobs), dtype)
    assert blocks > 0
    dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).clone().to(gpu_device)
    hip.check(hip.lib.dn_pack_batched(hip.ptr(dev), len(jobs), blocks, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    for (w, cout, cin, taps, mode), (out, n) in zip(specs, outs):
        assert bool((out[n:] == 0xAB).all()), mode
        if mode <= hip.PACK_CONVT_DGRAD:
            ref = _pack(hip, w, None, mode, dtype, gpu_device)
            assert torch.equal(out[:n], ref.view(torch.uint8).reshape(-1)), mode
        else:
            got = out[:n].view(torch.float32).cpu()
            cp = hip.channel_pad(cout)
            if mode == hip.PACK_VEC:
                want = torch.zeros(cp); want[:cout] = w.cpu()
            elif mode == hip.PACK_UP_BIAS:
                want = hip.up_bias2(w, dtype).cpu()       # [output plane][k][channels of the plane]
            else:
                want = torch.zeros(taps, cp); want[:, :cout] = w.cpu().reshape(cout, taps).t(); want = want.reshape(-1)
            assert torch.equal(got, want), mode
    bad = (hip.DnPackJob * 1)(hip.DnPackJob(hip.ptr(b), None, hip.ptr(outs[0][0]), 40, 1, 1, 9, 0, 0, 0, 0))
    assert hip.lib.dn_pack_jobs_prepare(bad, 1, dtype) == -1
