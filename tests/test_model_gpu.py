"""The reference's model API end to end on the GPU: predict_batch -> Prediction -> allele caller, and fit()."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu


def _images(x, t=None, panel=None):
    from dnanet_amd.data import Annotation, ArrayImage
    out = []
    for i in range(x.shape[0]):
        data = np.transpose(x[i], (1, 2, 0)).astype(np.int16)                   # (H,W,1) like HIDImage.data
        ann = Annotation(image=np.transpose(t[i], (1, 2, 0)).astype(np.int8)) if t is not None else None
        out.append(ArrayImage(data=data, annotation=ann, path=f"profile_{i}.hid", _scaler=np.linspace(65, 475, x.shape[3]),
                              _panel=panel))
    return out


def test_predict_batch_real_profiles_probabilities_masks_alleles(gpu_device):
    """The two real RD profiles of the reference's test resources through predict_batch (fp32 mode):
    probabilities within tolerance of the reference's, masks and called alleles identical."""
    from dnanet_amd.allele_caller import NearestBasePairCaller
    from dnanet_amd.data import Panel
    from dnanet_amd.model import DNANet_UNet
    g = load_golden("g_real_d4_k35_f8")
    panel = Panel(os.path.join(GOLDEN_DIR, "panel.xml"))
    model = DNANet_UNet(4, (3, 5), 8, device="cuda", apply_allele_caller=True)
    model._model.load_state_dict(g["sd0"])
    images = _images(g["x"], g["t"], panel)
    preds = model.predict_batch(images)
    ref_probs = torch.sigmoid(torch.from_numpy(g["logits_eval"])).numpy()
    for i, p in enumerate(preds):
        assert p.image.shape == (5, 4096, 1) and p.image.dtype == np.float32
        assert p.meta["original_image_path"] == f"profile_{i}.hid"
        ref = np.transpose(ref_probs[i], (1, 2, 0))
        assert np.abs(p.image - ref).max() < 1e-5
        assert np.array_equal(p.image >= 0.5, ref >= 0.5)
        expect = NearestBasePairCaller().translate_pixels_to_alleles(images[i]._scaler[None], ref, images[i].data, panel)
        got = p.meta["called_alleles"]
        key = lambda ms: sorted((m.dye_row, m.name, tuple(sorted((a.name, a.height) for a in m.alleles))) for m in ms)
        assert key(got) == key(expect)
    # predict() == predict_batch()[0]
    one = model.predict(images[1])
    assert np.array_equal(one.image, preds[1].image)


def test_predict_batch_bf16_mask_agreement(gpu_device):
    from dnanet_amd.model import DNANet_UNet
    g = load_golden("g_real_d4_k35_f8")
    model = DNANet_UNet(4, (3, 5), 8, device="cuda", apply_allele_caller=False, dtype="bf16")
    model._model.load_state_dict(g["sd0"])
    preds = model.predict_batch(_images(g["x"]))
    ref = torch.sigmoid(torch.from_numpy(g["logits_eval"])).numpy()
    agree = np.mean([(np.transpose(ref[i], (1, 2, 0)) >= 0.5) == (p.image >= 0.5) for i, p in enumerate(preds)])
    assert agree > 0.995, agree       # bf16 mode: reported agreement, not identity


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16x3@2"])
def test_predict_batch_bf16x3_masks_and_alleles_of_the_reference(gpu_device, dtype):
    """The real RD profiles through predict_batch in the hi + lo split bf16 modes: the reference's masks and called alleles,
    probabilities to 3e-4 (plain bf16: agreement only, see above); fit() refuses the inference-only dtype."""
    from dnanet_amd.allele_caller import NearestBasePairCaller
    from dnanet_amd.data import Panel
    from dnanet_amd.model import DNANet_UNet
    g = load_golden("g_real_d4_k35_f8")
    panel = Panel(os.path.join(GOLDEN_DIR, "panel.xml"))
    model = DNANet_UNet(4, (3, 5), 8, device="cuda", apply_allele_caller=True, dtype=dtype)
    model._model.load_state_dict(g["sd0"])
    images = _images(g["x"], g["t"], panel)
    preds = model.predict_batch(images)
    ref_probs = torch.sigmoid(torch.from_numpy(g["logits_eval"])).numpy()
    key = lambda ms: sorted((m.dye_row, m.name, tuple(sorted((a.name, a.height) for a in m.alleles))) for m in ms)
    for i, p in enumerate(preds):
        ref = np.transpose(ref_probs[i], (1, 2, 0))
        assert np.abs(p.image - ref).max() < (3e-4 if dtype == "bf16x3" else 2e-3)
        assert np.array_equal(p.image >= 0.5, ref >= 0.5)
        expect = NearestBasePairCaller().translate_pixels_to_alleles(images[i]._scaler[None], ref, images[i].data, panel)
        assert key(p.meta["called_alleles"]) == key(expect)
    with pytest.raises(RuntimeError, match="inference mode"):
        model.fit(images, num_epochs=1, batch_size=2)


def test_fit_loop_matches_oracle_loop(gpu_device):
    """fit(): chunking, dropped remainder, per-epoch ExponentialLR, Adam -- against the oracle's restatement of the
    reference loop (trainable_unet.py:183-203,273-274,339-347)."""
    from dnanet_amd.model import DNANet_UNet
    from oracle import unet_oracle as O
    g = load_golden("g_d2_k35_f8_h5_w64")
    rng = np.random.default_rng(5)
    x = np.concatenate([g["x"], np.abs(rng.normal(0, 300, size=(3, 1, 5, 64))).round().astype(np.float32)])
    t = np.concatenate([g["t"], (rng.random((3, 1, 5, 64)) < 0.1).astype(np.int8)])
    images = _images(x, t)
    model = DNANet_UNet(2, (3, 5), 8, device="cuda", apply_allele_caller=False)
    model._model.load_state_dict(g["sd0"])
    model.fit(images, batch_size=2, num_epochs=2, learning_rate=1e-3, use_scheduler=True)
    sd = OrderedDict((k, v.clone()) for k, v in g["sd0"].items())
    state = O.AdamState(O.trainable_keys(sd), sd)
    for epoch in range(2):
        lr = O.exponential_lr(1e-3, epoch)
        for s in range(len(images) // 2):              # 5 images, batch 2 -> 2 steps, the 5th image is never used
            xs = torch.from_numpy(x[2 * s:2 * s + 2]); ts = torch.from_numpy(t[2 * s:2 * s + 2])
            O.train_step(sd, xs, ts, state, lr, 5e-4)
    got = model._model.state_dict()
    for k, v in sd.items():
        if v.dtype == torch.int64:
            assert int(got[k]) == int(v) == 4
        elif "running" in k:
            assert torch.allclose(got[k].cpu(), v, rtol=1e-4, atol=1e-5), k
        else:
            d = (got[k].cpu() - v).abs()
            assert float(d.max()) <= 2.5 * 1e-3 * 4 and float((d > 2e-4 + 2e-3 * v.abs()).float().mean()) < 1e-2, k


def test_the_reference_smoke_test_against_the_oracle_loop(gpu_device):
    """The reference's one hot-path test (``tests/models/segmentation/test_unet.py:12-19``): ``DNANet_UNet(4, (1, 3))`` with its
    default 64 filters, ``fit(batch_size=1, num_epochs=5)`` on the two real RD profiles, then ``predict_batch`` -- there a smoke
    test without numeric assertions, here the same calls checked against the oracle running the reference's loop."""
    from oracle import unet_oracle as O
    from dnanet_amd.model import DNANet_UNet
    g = load_golden("g_real_d4_k35_f8")
    x, t = g["x"], g["t"]
    images = _images(x, t)
    sd0 = O.init_state_dict(4, (1, 3), 64, seed=9)
    model = DNANet_UNet(4, (1, 3), device="cuda", apply_allele_caller=False)
    assert model._model.num_filters == 64
    model._model.load_state_dict(sd0)
    model.fit(images, batch_size=1, num_epochs=5)                     # fit's defaults: lr 0.005, weight decay 5e-4, no scheduler
    sd = OrderedDict((k, v.clone()) for k, v in sd0.items())
    state = O.AdamState(O.trainable_keys(sd), sd)
    losses = []
    for epoch in range(5):
        for s in range(2):
            loss, _, _ = O.train_step(sd, torch.from_numpy(x[s:s + 1]), torch.from_numpy(t[s:s + 1]), state, 0.005, 5e-4)
            losses.append(loss)
    got = model._model.state_dict()
    assert int(got["bottleneck.double_conv.1.num_batches_tracked"]) == 10
    # ten Adam steps at lr 5e-3 from the same start: parameters move by up to 5e-2; trajectories agree to a fraction of that
    moved = max(float((sd[k] - sd0[k]).abs().max()) for k in sd if sd[k].dtype != torch.int64)
    worst = max(float((got[k].cpu() - sd[k]).abs().max()) for k in sd if sd[k].dtype != torch.int64 and "running" not in k)
    assert moved > 1e-2 and worst < 0.1 * moved, (moved, worst)
    preds = model.predict_batch(images)
    ref = torch.sigmoid(O.unet_forward(sd, torch.from_numpy(x), train=False)).numpy()
    for i, p in enumerate(preds):
        # ten aggressive steps on single-profile batches amplify fp32 round-off at the sharp peaks: typical pixels agree
        assert p.image.shape == (5, 4096, 1)
        assert np.median(np.abs(p.image - np.transpose(ref[i], (1, 2, 0)))) < 2e-2
    assert losses[-1] < losses[0]
    # the prediction path itself, on the oracle's end state: exact to fp32 round-off
    model._model.load_state_dict(sd)
    for i, p in enumerate(model.predict_batch(images)):
        assert np.abs(p.image - np.transpose(ref[i], (1, 2, 0))).max() < 1e-4


def test_fit_learns_and_checkpoints(gpu_device, tmp_path):
    from dnanet_amd.model import DNANet_UNet
    from dnanet_amd.synthetic import make_batch
    x, t = make_batch(12, 5, 256, seed=3, unique=12)
    images = _images(x, t)
    torch.manual_seed(0)
    model = DNANet_UNet(2, (3, 5), 8, device="cuda", apply_allele_caller=False, dtype="bf16")

    def train_mode_loss():      # batch-statistics loss on the training images (no update)
        model._model.train()
        vals = [float(model.step(b, [])) for b in ([images[:4], images[4:8]])]
        model._model._saved = None
        return sum(vals) / len(vals)

    before = train_mode_loss()
    model.fit(images[:8], batch_size=4, num_epochs=10, learning_rate=5e-3, validation_set=images[8:], min_delta=-99999,
              checkpoint_dir=str(tmp_path), save_best=True, use_scheduler=True)
    after = train_mode_loss()
    assert after < before - 0.02, (before, after)
    assert not (tmp_path / "best_model").exists()      # restored and removed (trainable_unet.py:290-293)
    model.save(tmp_path / "final")
    sd = torch.load(tmp_path / "final" / "DNANet_UNet_checkpoint.pt", weights_only=True)
    assert len(sd) == 7 * 10 + 2 * 2 + 2


def test_data_parallel_step_world1_equals_plain_step(gpu_device):
    """The DP trainer (RCCL process group of one rank here) must reproduce the plain step bit for bit."""
    import torch.distributed as dist
    from dnanet_amd.distributed import DataParallelTrainer
    from dnanet_amd.unet import HipAdam, HipUNet
    g = load_golden("g_d2_k35_f8_h5_w64")
    x, t = torch.from_numpy(g["x"]).to(gpu_device), torch.from_numpy(g["t"]).to(gpu_device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu_device)
    try:
        results = []
        for dp in (False, True):
            m = HipUNet(2, (3, 5), 8, gpu_device, dtype="fp32")
            m.load_state_dict(g["sd0"])
            m.train()
            opt = HipAdam(m, lr=1e-3, weight_decay=5e-4)
            if dp:
                tr = DataParallelTrainer(m, opt, min_bucket_elems=10000)
                for _ in range(2):
                    sums = tr.step(x, t)
            else:
                for _ in range(2):
                    logits = m(x); sums = m.dice_forward(logits, t); m.backward(); opt.step(); opt.zero_grad()
            results.append((m.flat_params.clone(), float(sums[4])))
        assert torch.equal(results[0][0], results[1][0])
        assert abs(results[0][1] - results[1][1]) < 1e-7
    finally:
        dist.destroy_process_group()


def test_fit_data_parallel_world1_equals_plain_fit(gpu_device, tmp_path):
    """fit() inside a process group (here one rank, forced with data_parallel=True): the slice / all-reduce / broadcast
    path -- with a validation set, save_best and the scheduler -- gives the parameters, statistics and checkpoint of
    the plain loop; a global batch that the ranks cannot share is refused."""
    import torch.distributed as dist
    from dnanet_amd.model import DNANet_UNet
    from dnanet_amd.synthetic import make_batch
    x, t = make_batch(10, 5, 256, seed=12, unique=10)
    images = _images(x, t)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    states = []
    for dp in (False, True):
        torch.manual_seed(0)
        model = DNANet_UNet(2, (3, 5), 8, device="cuda", apply_allele_caller=False)
        if dp:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu_device)
        try:
            ckpt = tmp_path / f"dp{int(dp)}"
            model.fit(images[:8], batch_size=4, num_epochs=3, learning_rate=2e-3, validation_set=images[8:], min_delta=-99999,
                      checkpoint_dir=str(ckpt), save_best=True, use_scheduler=True, hip_graph=False,
                      **({"data_parallel": True} if dp else {}))
            model.save(str(ckpt))
            assert os.path.exists(ckpt / "DNANet_UNet_checkpoint.pt") and not os.path.exists(ckpt / "best_model")
            if dp:
                with pytest.raises(ValueError):
                    model.fit(images[:8], batch_size=4, num_epochs=1, hip_graph=True, data_parallel=True)
        finally:
            if dp:
                dist.destroy_process_group()
        states.append({k: v.detach().cpu().clone() for k, v in model._model.state_dict().items()})
    for k, v in states[0].items():
        assert torch.equal(states[1][k], v), k
    with pytest.raises(RuntimeError):
        DNANet_UNet(2, (3, 5), 8, device="cuda", apply_allele_caller=False).fit(images[:4], batch_size=4, num_epochs=1,
                                                                                data_parallel=True)


def test_resident_dataset_batches_equal_per_step_tensorisation(gpu_device):
    """Input pipeline: batches gathered from the HBM-resident copy are the tensors get_inputs /
    get_targets build from the images themselves -- in dataset order (a view), shuffled (a gather), and a batch
    holding an unknown image falls back to the per-step path."""
    from dnanet_amd.model import DNANet_UNet
    from dnanet_amd.synthetic import make_batch
    x, t = make_batch(7, 5, 256, seed=9, unique=7)
    images = _images(x, t)
    model = DNANet_UNet(2, (3, 5), 8, device="cuda", apply_allele_caller=False)
    plain = [(model.get_inputs(b).clone(), model.get_targets(b).clone())
             for b in (images[2:5], [images[6], images[0], images[3]])]
    assert model.make_resident(images[:4], images[4:])
    for (px, pt), b in zip(plain, (images[2:5], [images[6], images[0], images[3]])):
        rx, rt = model.get_inputs(b), model.get_targets(b)
        assert rx.dtype == px.dtype and rt.dtype == pt.dtype and torch.equal(rx, px) and torch.equal(rt, pt)
    stranger = _images(x[:1] + 1, t[:1])[0]
    mixed = model.get_inputs([images[1], stranger])
    assert torch.equal(mixed[0], plain[0][0][0] * 0 + model.get_inputs([images[1]])[0]) and float(mixed[1, 0, 0, 0]) == x[0, 0, 0, 0] + 1
    model.release_resident()
    assert model._resident is None


def test_fit_hip_graph_replay_equals_launch_by_launch(gpu_device, caplog):
    """fit() replays the whole training step from a HIP graph for small batches; parameters, running statistics and
    the reported losses equal the launch-by-launch loop (the kernels are deterministic; the only difference allowed
    is the device-side pow() of Adam's bias correction)."""
    import logging
    from dnanet_amd import unet as U
    from dnanet_amd.model import DNANet_UNet
    from dnanet_amd.synthetic import make_batch
    x, t = make_batch(9, 5, 256, seed=4, unique=9)
    images = _images(x, t)
    states, losses, replays = [], [], []
    real_call = U.GraphedTrainStep.__call__
    for use_graph in (False, True):
        torch.manual_seed(0)
        model = DNANet_UNet(3, (3, 5), 8, device="cuda", apply_allele_caller=False)
        count = [0]

        def counting(self, *a, _c=count, **k):
            _c[0] += 1
            return real_call(self, *a, **k)
        U.GraphedTrainStep.__call__ = counting
        try:
            with caplog.at_level(logging.INFO, logger="dnanet"):
                caplog.clear()
                model.fit(images, batch_size=4, num_epochs=3, learning_rate=2e-3, use_scheduler=True, hip_graph=use_graph)
                losses.append([r.getMessage() for r in caplog.records if "Training loss" in r.getMessage()])
        finally:
            U.GraphedTrainStep.__call__ = real_call
        replays.append(count[0])
        states.append({k: v.detach().cpu().clone() for k, v in model._model.state_dict().items()})
    assert replays == [0, 5]                      # 2 steps x 3 epochs, the very first one is the eager warm-up
    assert len(losses[0]) == 3
    for a, b in zip(losses[0], losses[1]):
        assert abs(float(a.split()[-1]) - float(b.split()[-1])) < 1e-5
    for k, v in states[0].items():
        if v.dtype == torch.int64:
            assert int(v) == int(states[1][k]) == 6, k
        else:
            assert torch.allclose(states[1][k], v, rtol=1e-5, atol=1e-7), k


def test_device_pixel_counts_equal_host_metrics(gpu_device):
    """the confusion counts predict_batch takes on the GPU give the same pixel metrics as the
    reference-style boolean passes over the host copies (integers: exact), including pixels at exactly 0.5."""
    from dnanet_amd import metrics as M
    from dnanet_amd.model import DNANet_UNet
    from dnanet_amd.prediction import Prediction
    from dnanet_amd.synthetic import make_batch
    x, t = make_batch(6, 5, 256, seed=21, unique=6)
    rng = np.random.default_rng(0)
    t = (rng.random(t.shape) < 0.2).astype(np.int8)          # plenty of positives so every count is non-trivial
    images = _images(x, t)
    torch.manual_seed(3)
    model = DNANet_UNet(2, (3, 5), 8, device="cuda", apply_allele_caller=False)
    preds = model.predict_batch(images)
    assert all(hasattr(p, "_pixel_counts") for p in preds)
    bare = [Prediction(image=p.image, meta=p.meta) for p in preds]
    pixel_metrics = {k: v for k, v in M.METRICS.items() if not k.startswith("allele")}     # the allele metrics read the caller's output
    for name, fn in pixel_metrics.items():
        assert fn(images=images, predictions=preds) == fn(images=images, predictions=bare), name
    # the kernel itself on crafted values: 0.5 is positive for ">=" and negative for ">"
    from dnanet_amd import hip
    probs = torch.tensor([[0.5, 0.5, 0.7, 0.2, 0.49999997, 0.9, 0.1, 0.5]], device="cuda")
    truth = torch.tensor([[1, 0, 1, 1, 0, 0, 0, 2]], dtype=torch.int8, device="cuda")
    counts = torch.zeros((1, 7), dtype=torch.int32, device="cuda")
    hip.check(hip.lib.dn_pixel_confusion(hip.ptr(probs), hip.ptr(truth), 1, 8, 0.5, 0.5, hip.ptr(counts), hip.stream_ptr()))
    assert counts.cpu().tolist() == [[2, 2, 1, 1, 1, 2, 1]]
    # an annotation that is not a 0/1 mask: no device counts for that profile, host path as before; float 0/1 masks are fine
    images[0].annotation.image = images[0].annotation.image.astype(np.float64) * 0.5
    images[1].annotation.image = images[1].annotation.image.astype(np.float64)
    again = model.predict_batch(images)
    assert not hasattr(again[0], "_pixel_counts") and all(hasattr(p, "_pixel_counts") for p in again[1:])
    bare = [Prediction(image=p.image, meta=p.meta) for p in again]
    for name, fn in pixel_metrics.items():
        assert fn(images=images, predictions=again) == fn(images=images, predictions=bare), name


def test_empty_and_short_inputs(gpu_device):
    """Edge cases of the model API: an empty batch predicts nothing; a dataset shorter than one batch still trains ONE step per
    epoch on the short batch (``steps_per_epoch = max(1, len // batch_size)``, trainable_unet.py:183-184, with ``chunks`` keeping the
    remainder); one image is a batch of one."""
    from dnanet_amd.model import DNANet_UNet
    from dnanet_amd.synthetic import make_batch
    x, t = make_batch(3, 5, 256, seed=1, unique=3)
    images = _images(x, t)
    model = DNANet_UNet(2, (3, 5), 8, device="cuda", apply_allele_caller=False)
    assert model.predict_batch([]) == []
    before = model._model.flat_params.clone()
    model.fit(images, batch_size=4, num_epochs=2)
    assert not torch.equal(model._model.flat_params, before)
    assert int(model._model.state_dict()["bottleneck.double_conv.1.num_batches_tracked"]) == 2
    one = model.predict(images[0])
    assert one.image.shape == (5, 256, 1) and np.array_equal(one.image, model.predict_batch(images)[0].image)
    # a validation set shorter than a batch: zero validation steps per epoch (len // batch_size, :186-187), training goes on
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.fit(images, batch_size=2, num_epochs=2, validation_set=images[:1])
    assert int(model._model.state_dict()["bottleneck.double_conv.1.num_batches_tracked"]) >= 4
