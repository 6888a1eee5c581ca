"""The one entry point that allocates nothing: dn_bn_train_forward_noalloc (include/dnanet_hip.h) against the entry point
it is a variant of, dn_bn_train_forward, bit for bit -- two calls in a row on the same GPU, the second with a smaller
input (the built-in scratch keeps the first call's partials), then the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bn(fn_name, y_nchw, dtype, gamma, beta, with_ws):
    from dnanet_amd import hip
    dev = y_nchw.device
    n, c, h, w = y_nchw.shape
    cp = hip.channel_pad(c)
    y = hip.PlaneBuffer.from_nchw(y_nchw, dtype)
    a = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
    vec = lambda v: torch.cat([v, torch.zeros(cp - c, device=dev)]).contiguous()
    g, b = vec(gamma), vec(beta)
    rm, rv = torch.zeros(cp, device=dev), torch.ones(cp, device=dev)
    outs = [torch.full((cp,), 7.0, device=dev) for _ in range(4)]
    args = [y.ref(c), a.ref(c), n, h, w, hip.ptr(g), hip.ptr(b), hip.ptr(rm), hip.ptr(rv), 0.1, 1e-5] + [hip.ptr(o) for o in outs]
    if with_ws:
        ws = torch.full((hip.partial_floats(c),), float("nan"), device=dev)
        args.append(hip.ptr(ws))
    hip.check(getattr(hip.lib, fn_name)(*args, dtype, hip.stream_ptr()))
    torch.cuda.synchronize()
    return [a.data.clone(), rm, rv] + outs


@pytest.mark.parametrize("dtype_name", ["fp32", "bf16"])
def test_no_alloc_entry_point_matches_workspace_entry_point(gpu_device, dtype_name):
    from dnanet_amd import hip
    dtype = hip.F32 if dtype_name == "fp32" else hip.BF16
    gen = torch.Generator().manual_seed(5)
    # the first call fills the built-in scratch with the partials of a large, wide input; the second is smaller in every
    # dimension (fewer blocks, fewer channels), so stale partials of the first are right next to the ones it reads
    for (n, c, h, w) in ((6, 96, 5, 512), (2, 40, 5, 128)):
        y = (torch.randn(n, c, h, w, generator=gen) * 3 + 1).to(gpu_device)
        if dtype == hip.BF16:
            y = y.bfloat16().float()
        gamma = (torch.rand(c, generator=gen) + 0.5).to(gpu_device)
        beta = torch.randn(c, generator=gen).to(gpu_device)
        ref = _bn("dn_bn_train_forward", y, dtype, gamma, beta, with_ws=True)
        got = _bn("dn_bn_train_forward_noalloc", y, dtype, gamma, beta, with_ws=False)
        for r, g_ in zip(ref, got):
            assert torch.equal(r, g_)
        # and against torch: batch statistics of y
        mean = y.mean(dim=(0, 2, 3))
        assert torch.allclose(got[3][:c], mean, rtol=1e-5, atol=1e-5)


def test_no_alloc_entry_point_rejects_more_channels_than_its_scratch_holds(gpu_device):
    from dnanet_amd import hip
    c = 520
    y = torch.zeros(1, c, 1, 16, device=gpu_device)
    with pytest.raises(hip.HipError, match="at most 512"):
        _bn("dn_bn_train_forward_noalloc", y, hip.F32, torch.ones(c, device=gpu_device), torch.zeros(c, device=gpu_device), False)


def test_no_alloc_entry_point_under_hip_graph_capture(gpu_device):
    """The callers it is for: code under HIP graph capture.  The first call on a device looks up the address of the built-in
    scratch (hipGetSymbolAddress), so one call is made before the capture (include/dnanet_hip.h says so); then the call is
    captured, replayed twice on new data, and compared with the workspace entry point; nothing is allocated by the capture
    or the replays."""
    from dnanet_amd import hip
    dev, dtype = gpu_device, hip.BF16
    n, c, h, w = 4, 64, 5, 256
    cp = hip.channel_pad(c)
    gen = torch.Generator().manual_seed(9)
    y_static = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
    a_static = hip.PlaneBuffer(n, hip.planes(c, dtype), h, w, dtype, dev, zero=True)
    gamma, beta = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
    rm, rv = torch.zeros(cp, device=dev), torch.ones(cp, device=dev)
    outs = [torch.zeros(cp, device=dev) for _ in range(4)]

    def call():
        hip.check(hip.lib.dn_bn_train_forward_noalloc(y_static.ref(c), a_static.ref(c), n, h, w, hip.ptr(gamma), hip.ptr(beta),
                                                      hip.ptr(rm), hip.ptr(rv), 0.1, 1e-5, *[hip.ptr(o) for o in outs], dtype,
                                                      hip.stream_ptr()))

    call()                                   # warm-up outside the capture: resolves the scratch address on this device
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    mallocs = lambda: torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    with torch.cuda.graph(graph):
        call()
    replay_mallocs = 0
    for _ in range(2):
        y = (torch.randn(n, c, h, w, generator=gen) * 2 + 0.5).to(dev).bfloat16().float()
        y_static.data.copy_(hip.PlaneBuffer.from_nchw(y, dtype).data)
        rm.zero_(); rv.fill_(1.0)
        before = mallocs()
        graph.replay()
        torch.cuda.synchronize()
        replay_mallocs += mallocs() - before
        got = [a_static.data.clone(), rm.clone(), rv.clone()] + [o.clone() for o in outs]
        ref = _bn("dn_bn_train_forward", y, dtype, gamma[:c], beta[:c], with_ws=True)
        for r, g_ in zip(ref, got):
            assert torch.equal(r, g_)
    assert replay_mallocs == 0
