"""bf16x3: hi + lo split bf16 tensors and weights (three MFMAs per product).  The operators against torch fp32 on the CPU
(tolerance: the 2^-17 relative error of a two-term bf16 split, far below bf16's 2^-9), the whole net against the fp32 CPU
oracle at the logit level, and masks / called alleles at the shipped size."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dnanet_amd import hip  # noqa: E402
from dnanet_amd.hip import BF16, PlaneBuffer  # noqa: E402


def split_buffer(x: torch.Tensor, extra_planes: int = 0) -> PlaneBuffer:
    """(N,C,H,W) fp32 on the device -> [hi planes | lo planes] bf16 buffer"""
    hi = x.bfloat16().float()
    lo = (x - hi).bfloat16().float()
    bh, bl = PlaneBuffer.from_nchw(hi, BF16), PlaneBuffer.from_nchw(lo, BF16)
    buf = PlaneBuffer(x.shape[0], 2 * bh.planes_total + extra_planes, x.shape[2], x.shape[3], BF16, x.device, zero=True)
    buf.data[:, :bh.planes_total] = bh.data
    buf.data[:, bh.planes_total:2 * bh.planes_total] = bl.data
    return buf


def joined(buf: PlaneBuffer, channels: int, plane0: int, lo_off: int) -> torch.Tensor:
    return buf.to_nchw(channels, plane0) + buf.to_nchw(channels, plane0 + lo_off)


def terms(w: torch.Tensor, dim: int, split_input: bool) -> torch.Tensor:
    e = 16
    pad = (-w.shape[dim]) % e
    if pad:
        shape = list(w.shape); shape[dim] = pad
        w = torch.cat([w, torch.zeros(shape, device=w.device)], dim)
    hi = w.bfloat16().float()
    return torch.cat([hi, hi, w - hi] if split_input else [hi, w - hi], dim).contiguous()


def pack(w, cout, cin, kh, kw, mode):
    pk = torch.empty(hip.lib.dn_packed_weight_bytes(cout, cin, kh, kw, mode, BF16), dtype=torch.uint8, device=w.device)
    hip.check(hip.lib.dn_pack_weight(hip.ptr(w), None, hip.ptr(pk), cout, cin, kh, kw, mode, BF16, hip.stream_ptr()))
    return pk


@pytest.mark.parametrize("n,h,w,cin,cout,split_in,pool", [(2, 5, 256, 32, 64, True, True), (3, 6, 128, 64, 32, True, False),
                                                           (2, 5, 512, 48, 64, False, False), (16, 5, 256, 64, 64, True, True),
                                                           (2, 3, 384, 16, 32, True, True), (1, 4, 128, 40, 24, True, False),
                                                           (2, 5, 160, 32, 32, True, True), (1, 7, 96, 64, 128, False, True), (300, 1, 64, 32, 64, True, False)])
def test_conv2d_x3_vs_torch(gpu_device, n, h, w, cin, cout, split_in, pool):
    torch.manual_seed(cin + cout)
    dev = gpu_device
    x = torch.randn(n, cin, h, w) * 3
    wt = torch.randn(cout, cin, 3, 5) / (cin * 15) ** 0.5
    b = torch.randn(cout)
    if not split_in:
        x = x.bfloat16().float()
    want = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=(1, 2))).float()
    xp = hip.planes(cin, BF16)
    xbuf = split_buffer(x.to(dev)) if split_in else PlaneBuffer.from_nchw(x.to(dev), BF16)
    pk = pack(terms(wt.to(dev), 1, split_in), cout, (3 if split_in else 2) * xp * 16, 3, 5, hip.PACK_FWD)
    bias = torch.zeros(hip.channel_pad(cout), device=dev); bias[:cout] = b.to(dev)
    yp = hip.planes(cout, BF16)
    ybuf = PlaneBuffer(n, 2 * yp + 1, h, w, BF16, dev, zero=True)          # lo part one plane further than needed: lo_off is free
    pbuf = PlaneBuffer(n, 2 * yp, h, w // 2, BF16, dev, zero=True)
    pooled = pbuf.ref3(cout, 0, yp) if pool else None
    hip.check(hip.lib.dn_conv2d_x3(xbuf.ref3(cin, 0, xp if split_in else 0), hip.ptr(pk), hip.ptr(bias), ybuf.ref3(cout, 0, yp + 1),
                                   ctypes.byref(pooled) if pool else None, n, h, w, 3, 5, 1, hip.stream_ptr()))
    got = joined(ybuf, cout, 0, yp + 1).cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 3e-5 * scale, float((got - want).abs().max()) / scale
    if pool:
        gp = joined(pbuf, cout, 0, yp).cpu()
        assert float((gp - F.max_pool2d(want, (1, 2))).abs().max()) <= 3e-5 * scale


@pytest.mark.parametrize("n,h,win,cin,cout,split_in", [(2, 5, 128, 64, 32, True), (2, 6, 256, 128, 64, False), (3, 5, 64, 32, 16, True),
                                                       (2, 6, 128, 256, 128, True)])
def test_conv_transpose_x3_vs_torch(gpu_device, n, h, win, cin, cout, split_in):
    torch.manual_seed(cin)
    dev = gpu_device
    x = torch.randn(n, cin, h, win) * 2
    if not split_in:
        x = x.bfloat16().float()
    wt = torch.randn(cin, cout, 1, 2) / cin ** 0.5
    b = torch.randn(cout)
    want = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=(1, 2)).float()
    xp, yp = hip.planes(cin, BF16), hip.planes(cout, BF16)
    xbuf = split_buffer(x.to(dev)) if split_in else PlaneBuffer.from_nchw(x.to(dev), BF16)
    pk = pack(terms(wt.to(dev), 0, split_in), cout, (3 if split_in else 2) * xp * 16, 1, 2, hip.PACK_CONVT_FWD)
    # into the "up" half of a split concat buffer [up_hi | skip_hi | up_lo | skip_lo]
    cat = PlaneBuffer(n, 4 * yp, h, 2 * win, BF16, dev, zero=True)
    hip.check(hip.lib.dn_conv_transpose_1x2_x3(xbuf.ref3(cin, 0, xp if split_in else 0), hip.ptr(pk), hip.ptr(hip.up_bias2(b.to(dev), BF16)),
                                               cat.ref3(cout, 0, 2 * yp), n, h, win, hip.stream_ptr()))
    got = joined(cat, cout, 0, 2 * yp).cpu()
    assert float((got - want).abs().max()) <= 3e-5 * float(want.abs().max())
    assert float(cat.data[:, yp:2 * yp].abs().max()) == 0 and float(cat.data[:, 3 * yp:].abs().max()) == 0      # skip half untouched


def test_conv_first_x3_vs_torch(gpu_device):
    torch.manual_seed(3)
    dev = gpu_device
    n, h, w, cout = 3, 6, 512, 32
    x = torch.randint(-50, 30000, (n, 1, h, w)).float()
    wt = torch.randn(cout, 1, 3, 5) * 1e-3
    b = torch.randn(cout)
    want = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=(1, 2))).float()
    wf = torch.zeros(15, hip.channel_pad(cout), device=dev)
    wf[:, :cout] = wt.reshape(cout, 15).t().to(dev)
    bias = torch.zeros(hip.channel_pad(cout), device=dev); bias[:cout] = b.to(dev)
    yp = hip.planes(cout, BF16)
    ybuf = PlaneBuffer(n, 2 * yp, h, w, BF16, dev, zero=True)
    hip.check(hip.lib.dn_conv_first_x3(hip.ptr(x.to(dev)), hip.ptr(wf), hip.ptr(bias), ybuf.ref3(cout, 0, yp), n, h, w, cout, 3, 5, 1,
                                       hip.stream_ptr()))
    got = joined(ybuf, cout, 0, yp).cpu()
    assert float((got - want).abs().max()) <= 3e-5 * float(want.abs().max())


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16x3@2", "bf16x3@1"])
@pytest.mark.parametrize("depth,filters,h,w,n,kernel", [(4, 32, 5, 4096, 4, (3, 5)), (2, 8, 6, 256, 3, (3, 5)), (3, 16, 5, 512, 18, (3, 5)),
                                                        (4, 8, 5, 256, 2, (1, 3)), (3, 64, 6, 128, 5, (1, 3))])
def test_net_logits_vs_oracle(gpu_device, dtype, depth, filters, h, w, n, kernel):
    from oracle import unet_oracle as O
    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipUNet
    sd = O.init_state_dict(depth, kernel, filters, seed=5)
    g = torch.Generator().manual_seed(1)
    for k, v in sd.items():                      # BatchNorm statistics away from the initial (0, 1)
        if k.endswith("running_mean"): v.copy_(torch.randn(v.shape, generator=g) * 0.2)
        if k.endswith("running_var"): v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    x = torch.from_numpy(make_batch(n, h, w, seed=11, unique=n)[0])
    with torch.no_grad():
        want = O.unet_forward(sd, x, train=False)
    m = HipUNet(depth, kernel, filters, gpu_device, dtype=dtype)
    m.load_state_dict(sd)
    m.eval()
    logits, probs = m.infer(x.to(gpu_device))
    err = float((logits.cpu() - want).abs().max() / want.abs().max())
    full = m.split_levels == depth + 1
    assert err <= (5e-5 if full else 2e-2), err
    assert torch.allclose(probs.cpu(), torch.sigmoid(logits.cpu()), atol=1e-6)
    with pytest.raises(RuntimeError):
        m.train(); m(x.to(gpu_device))


def test_masks_and_alleles_vs_oracle_shipped_size(gpu_device):
    """Briefly trained shipped-size net, 2 real + 32 synthetic profiles: bf16x3 masks and called alleles against the fp32 CPU
    oracle (fp32 kernels: 0 / 0; plain bf16: 40-80 pixels, 2-10 alleles -- tests/test_parity_bf16_gpu.py)."""
    from oracle import unet_oracle as O
    from dnanet_amd.parity import briefly_trained_state, disagreement
    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipUNet
    sd = briefly_trained_state(4, (3, 5), 32, gpu_device, steps=40)
    x = torch.from_numpy(make_batch(32, 5, 4096, seed=4321, unique=32)[0])
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_real_d4_k35_f8.npz")
    real = torch.from_numpy(np.load(golden)["x"]).float()
    if real.shape[2:] == x.shape[2:]:
        x = torch.cat([real, x])
    with torch.no_grad():
        ref = torch.sigmoid(O.unet_forward(sd, x, train=False))
    xd = x.to(gpu_device)
    out = {}
    for dtype in ("bf16x3", "bf16x3@2"):
        m = HipUNet(4, (3, 5), 32, gpu_device, dtype=dtype)
        m.load_state_dict(sd)
        m.eval()
        out[dtype] = disagreement(m.infer(xd)[1], ref, xd)
    print(out)
    d = out["bf16x3"]
    assert d["mask_pixels_differing"] == 0 and d.get("alleles_differing", 0) == 0, d
    assert d["max_abs_prob_diff"] < 1e-4
    d = out["bf16x3@2"]
    assert d["mask_pixels_differing"] <= 40 and d.get("alleles_differing", 0) <= 4, d
