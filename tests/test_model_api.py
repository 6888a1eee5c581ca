"""Host-side boundary: registry / load_model, constructor and fit() contract, checkpoint wire
format, batching helper, Prediction.  No GPU needed."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_names, load_golden
from dnanet_amd.config_io import get_config_path, load_model, load_training_config, parse_config
from dnanet_amd.data import Annotation, ArrayImage, chunks
from dnanet_amd.model import BinaryAccuracy, DNANet_UNet
from dnanet_amd.prediction import Prediction


@pytest.fixture(autouse=True)
def _cwd(monkeypatch):
    monkeypatch.chdir(ROOT)


def test_load_model_from_file_and_name():
    for src in ("config/models/unet.yaml", "unet"):
        m = load_model(src) if torch.cuda.is_available() else _cpu_load(src)
        assert isinstance(m, DNANet_UNet)
        assert m._model.depth == 4 and m._model.kernel_size == (3, 5) and m._model.num_filters == 32
        assert m.allele_caller is not None
        assert len(m._model.state_dict()) == 136
        assert sum(p.numel() for p in m._model.parameters()) == 12_123_745


def _cpu_load(src):
    import dnanet_amd.config_io as cio
    import dnanet_amd.model as mm
    old = mm.TORCH_DEFAULT_DEVICE
    mm.TORCH_DEFAULT_DEVICE = "cpu"
    try:
        return cio.load_model(src)
    finally:
        mm.TORCH_DEFAULT_DEVICE = old


def test_get_config_path():
    assert get_config_path("unet", "models") == __import__("pathlib").Path("config/models/unet.yaml")
    assert str(get_config_path("x.yaml", None, "/r")) == "/r/config/x.yaml"


def test_training_config_values():
    cfg = load_training_config("segmentation")
    assert cfg["batch_size"] == 16 and cfg["num_epochs"] == 15 and cfg["learning_rate"] == 1e-4
    assert cfg["use_scheduler"] is True and cfg["min_delta"] == -99999 and cfg["steps_per_epoch"] is None


def test_load_model_directory_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        load_model(tmp_path)                           # no yaml
    (tmp_path / "a.yaml").write_text("model: {name: unet, depth: 2, kernel_size: [3, 5], num_filters: 8, device: cpu}\n")
    with pytest.raises(FileNotFoundError):
        load_model(tmp_path)                           # no checkpoint
    (tmp_path / "b.yaml").write_text("x: 1\n")
    with pytest.raises(ValueError):
        load_model(tmp_path)                           # two yamls


def test_directory_form_loads_checkpoint(tmp_path):
    g = load_golden("g_d2_k35_f8_h5_w64")
    (tmp_path / "config.yaml").write_text(
        "model: {name: unet, depth: 2, kernel_size: [3, 5], num_filters: 8, device: cpu}\ntraining: {batch_size: 4}\n")
    torch.save(g["sd0"], tmp_path / "DNANet_UNet_checkpoint.pt")      # a reference-format checkpoint
    m = load_model(tmp_path)
    sd = m._model.state_dict()
    assert list(sd.keys()) == list(g["sd0"].keys())
    for k, v in g["sd0"].items():
        assert sd[k].dtype == v.dtype and tuple(sd[k].shape) == tuple(v.shape) and torch.equal(sd[k].cpu(), v), k


@pytest.mark.parametrize("name", golden_names()[:2])
def test_checkpoint_roundtrip_is_reference_wire_format(tmp_path, name):
    g = load_golden(name)
    m = DNANet_UNet(int(g["depth"]), tuple(int(v) for v in g["kernel"]), int(g["filters"]), device="cpu")
    m._model.load_state_dict(g["sd0"])
    m.save(tmp_path)
    path = tmp_path / "DNANet_UNet_checkpoint.pt"
    assert path.exists()
    sd = torch.load(path, weights_only=True)
    assert list(sd.keys()) == list(g["sd0"].keys())
    for k, v in g["sd0"].items():
        assert sd[k].dtype == v.dtype and torch.equal(sd[k], v), k
    m2 = DNANet_UNet(int(g["depth"]), tuple(int(v) for v in g["kernel"]), int(g["filters"]), device="cpu")
    m2.load(tmp_path)
    assert torch.equal(m2._model.flat_params, m._model.flat_params)
    with pytest.raises(RuntimeError):
        bad = dict(g["sd0"]); bad.pop("outputs.bias")
        m2._model.load_state_dict(bad)             # strict, like nn.Module.load_state_dict


def test_fit_argument_contract():
    m = DNANet_UNet(2, (3, 5), 8, device="cpu")
    with pytest.raises(TypeError, match="validation_set"):
        m.fit([], save_best=True)
    with pytest.raises(TypeError, match="checkpoint_dir"):
        m.fit([], save_best=True, validation_set=[1])
    with pytest.raises(TypeError, match="checkpoint_dir"):
        m.fit([], tensorboard=True)
    import inspect
    sig = inspect.signature(m.fit)
    defaults = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert defaults == dict(batch_size=4, num_epochs=10, learning_rate=0.005, weight_decay=0.0005, tensorboard=False,
                            validation_set=None, min_delta=0.01, steps_per_epoch=None, use_evaluation_metric=True,
                            checkpoint_dir=None, save_best=False, use_scheduler=False)


def test_no_cpu_fallback():
    m = DNANet_UNet(2, (3, 5), 8, device="cpu")
    img = ArrayImage(data=np.zeros((5, 64, 1), dtype=np.int16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m._model(torch.zeros(1, 1, 5, 64))
    with pytest.raises(Exception):
        m.predict(img)
    with pytest.raises(ValueError):
        DNANet_UNet(2, (3, 5), 12, device="cpu")       # filters must be a multiple of 8
    with pytest.raises(RuntimeError):
        m._model.to("meta")


def test_default_init_statistics():
    m = DNANet_UNet(2, (3, 5), 8, device="cpu")
    sd = m._model.state_dict()
    w = sd["encoders.1.conv.double_conv.0.weight"]
    bound = 1 / (w.shape[1] * 15) ** 0.5
    assert 0.9 * bound < float(w.abs().max()) <= bound
    assert float(sd["encoders.0.conv.double_conv.1.weight"].min()) == 1 and float(sd["bottleneck.double_conv.4.bias"].abs().max()) == 0
    assert float(sd["bottleneck.double_conv.4.running_var"].min()) == 1 and int(sd["bottleneck.double_conv.4.num_batches_tracked"]) == 0


def test_get_inputs_targets_layout():
    m = DNANet_UNet(2, (3, 5), 8, device="cpu")
    rng = np.random.default_rng(0)
    data = rng.integers(0, 3000, size=(5, 64, 1)).astype(np.int64)
    img = ArrayImage(data=data, annotation=Annotation(image=(data > 2000).astype(np.int8)))
    x = m.get_input(img)
    assert x.shape == (1, 5, 64) and x.dtype == torch.float32 and torch.equal(x[0], torch.from_numpy(data[:, :, 0]).float())
    t = m.get_targets([img, img])
    assert t.shape == (2, 1, 5, 64) and t.dtype == torch.int8


def test_chunks_semantics():
    assert list(chunks(range(10), 2)) == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9]]
    assert list(chunks(range(11), 2))[-1] == [10]
    assert list(chunks(range(11), 2, skip_remainder=True))[-1] == [8, 9]
    assert list(chunks([], 3)) == []


def test_prediction_roundtrip_and_metric():
    p = Prediction(image=np.arange(6, dtype=np.float32).reshape(1, 6, 1), meta={"original_image_path": "x"})
    q = Prediction.from_dict(p.to_dict())
    assert np.array_equal(p.image, q.image) and q.meta == p.meta
    with pytest.raises(ValueError):
        p.label
    acc = BinaryAccuracy()
    acc.update(torch.tensor([0.5, 0.2, 0.9, 0.4]), torch.tensor([1, 0, 0, 1]))
    assert acc.compute() == 0.5


def test_parse_config_registry():
    made = parse_config({"model": {"name": "unet", "depth": 2, "kernel_size": [1, 3], "num_filters": 8, "device": "cpu"},
                         "other": {"name": "x", "v": [1, {"name": "y"}]}}, {"model": {"unet": DNANet_UNet}})
    assert isinstance(made["model"], DNANet_UNet) and made["other"] == {"v": [1, {}]}


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports every entry point include/dnanet_hip.h declares."""
    import re
    from dnanet_amd import hip
    header = open(os.path.join(ROOT, "include", "dnanet_hip.h")).read()
    declared = set(re.findall(r"^(?:int|long|const char\*)\s+(dn_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(hip.lib, sym), f"{sym} declared in include/dnanet_hip.h but not exported"
    assert declared == set(hip.exported_symbols())
    assert hip.lib.dn_abi_version() == 4
