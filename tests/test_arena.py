"""The persistent per-shape buffers of the passes (dnanet_amd.hip.StepArena, HipUNet._arena): slot reuse, in-pass release,
least-recently-used eviction (CPU), and on the GPU: no allocator traffic in steady state, results independent of what a
previous pass at another batch shape left in the arena."""
import pytest
import torch


def test_step_arena_hands_back_the_same_slots_every_pass():
    from dnanet_amd.hip import StepArena
    a = StepArena("cpu")
    first = [a.begin().take((4, 8), torch.float32), a.take((3,), torch.float64), a.take((4, 8), torch.float32)]
    assert a.allocations == 3 and first[0].data_ptr() != first[2].data_ptr()
    again = [a.begin().take((4, 8), torch.float32), a.take((3,), torch.float64), a.take((4, 8), torch.float32)]
    assert a.allocations == 3 and all(x.data_ptr() == y.data_ptr() for x, y in zip(first, again))
    assert a.nbytes == 2 * 4 * 8 * 4 + 3 * 8
    # a pass that takes another route than last time (a key its owner did not foresee) re-allocates from the point of divergence
    a.begin()
    assert a.take((4, 8), torch.float32).data_ptr() == first[0].data_ptr()
    other = a.take((5,), torch.float64)
    assert other.shape == (5,) and a.allocations == 4 and len(a.slots) == 2
    z = a.begin().take((2, 2), torch.float32, zero=True)
    z.fill_(3.0)
    assert float(a.begin().take((2, 2), torch.float32, zero=True).abs().max()) == 0.0      # zero=True clears a reused slot


def test_step_arena_release_reuses_a_buffer_inside_a_pass_deterministically():
    from dnanet_amd.hip import StepArena
    a = StepArena("cpu")
    ptrs = []
    for _ in range(3):
        a.begin()
        x = a.take((16,), torch.float32)
        y = a.take((16,), torch.float32)
        a.release(x)
        z = a.take((16,), torch.float32)                 # gets x back: nothing new
        w = a.take((16,), torch.float32)                 # the free list is empty again: a third slot
        ptrs.append((x.data_ptr(), y.data_ptr(), z.data_ptr(), w.data_ptr()))
        assert z.data_ptr() == x.data_ptr() and len({x.data_ptr(), y.data_ptr(), w.data_ptr()}) == 3
    assert ptrs[0] == ptrs[1] == ptrs[2] and a.allocations == 3


def test_model_arenas_are_keyed_by_shape_and_evicted_least_recently_used_first():
    from dnanet_amd.unet import HipUNet
    m = HipUNet(1, (3, 5), 8, "cpu")
    m.arena_budget_bytes = 3000
    a = m._arena("eval", 2, 5, 64); a.take((256,), torch.float32)       # 1024 bytes each
    b = m._arena("eval", 3, 5, 64); b.take((256,), torch.float32)
    assert m._arena("eval", 2, 5, 64) is a                               # a is now the most recently used
    c = m._arena("train", 2, 5, 64); c.take((256,), torch.float32)
    assert m.arena_stats() == {"arenas": 3, "bytes": 3072, "allocations": 3}
    d = m._arena("eval", 4, 5, 64)                                       # over budget: b (least recently used) goes
    keys = [k[:2] for k in m._arenas]
    assert ("eval", 3) not in keys and ("eval", 2) in keys and ("train", 2) in keys and m._arenas[list(m._arenas)[-1]] is d
    m.release_arenas()
    assert m.arena_stats()["arenas"] == 0


@pytest.mark.gpu
def test_training_and_inference_allocate_nothing_in_steady_state(gpu_device):
    """the step's buffers are a persistent arena -- after the first step at a shape, a step makes no device
    allocation (hipMalloc) and the arena does not grow; inference likewise.  The API's own return values (logits, loss
    sums) are ordinary small torch allocations served from the allocator's cache."""
    from dnanet_amd.synthetic import make_batch
    from dnanet_amd.unet import HipAdam, HipUNet
    xn, tn = make_batch(8, 5, 512, seed=3, unique=8)
    x, t = torch.from_numpy(xn).to(gpu_device), torch.from_numpy(tn).to(gpu_device)
    for dtype in ("fp32", "bf16"):
        torch.manual_seed(0)
        m = HipUNet(3, (3, 5), 16, gpu_device, dtype=dtype)
        m.train()
        opt = HipAdam(m, lr=1e-3, weight_decay=5e-4)

        def step():
            logits = m(x); sums = m.dice_forward(logits, t); m.backward(); opt.step(); opt.zero_grad(); return sums

        step(); step()
        torch.cuda.synchronize()
        stats0, mallocs0 = m.arena_stats(), torch.cuda.memory_stats(gpu_device).get("num_device_alloc", 0)
        for _ in range(5):
            sums = step()
        torch.cuda.synchronize()
        assert m.arena_stats() == stats0 and stats0["arenas"] == 1 and stats0["allocations"] > 40
        assert torch.cuda.memory_stats(gpu_device).get("num_device_alloc", 0) == mallocs0
        assert 0.0 < float(sums[4]) < 1.0
        m.eval()
        m.infer(x); m.infer(x[:3]); m.infer(x)
        torch.cuda.synchronize()
        stats1, mallocs1 = m.arena_stats(), torch.cuda.memory_stats(gpu_device).get("num_device_alloc", 0)
        assert stats1["arenas"] == 3                      # train @ 8, eval @ 8, eval @ 3
        ref = m.infer(x)[0].clone()
        m.infer(x[:3])
        out = m.infer(x)[0]
        torch.cuda.synchronize()
        assert torch.equal(out, ref)                      # whatever the pass at the other shape left behind does not matter
        assert m.arena_stats() == stats1 and torch.cuda.memory_stats(gpu_device).get("num_device_alloc", 0) == mallocs1
