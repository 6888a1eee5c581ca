"""Host logic around the path: array datasets (split contract of the reference's SimpleDataset), config registry for
datasets / metrics, command-line flags of train.py / evaluate.py / cross_validate.py; on the GPU a tiny end-to-end run."""
import json
import os
import sys

import numpy as np
import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dnanet_amd.config_io import load_dataset, load_evaluation_config  # noqa: E402
from dnanet_amd.data import ArrayDataset, NpyDirDataset, SyntheticDataset  # noqa: E402


def test_split_is_seeded_and_partitions():
    ds = ArrayDataset(list(range(20)))
    a, b = ds.split(0.75, seed=7)
    a2, b2 = ds.split(0.75, seed=7)
    assert list(a) == list(a2) and list(b) == list(b2)
    assert len(a) == 15 and len(b) == 5 and sorted(list(a) + list(b)) == list(range(20))
    with pytest.raises(ValueError):
        ds.split(1.0)
    # the known answer of the reference's own test (tests/data/data_models/test_base.py:15-30): 20 images, fraction 0.8, seed 42
    # -> images 4, 14, 5, 19 form the second part, in that order; `-rs 42` reproduces the reference's partitions
    d1, d2 = ArrayDataset([f"dummy_path_{i}" for i in range(20)]).split(fraction=0.8, seed=42)
    assert len(d1) == 16 and list(d2) == ["dummy_path_4", "dummy_path_14", "dummy_path_5", "dummy_path_19"]
    d1, d2 = ArrayDataset([f"dummy_path_{i}" for i in range(1, 20)]).split(fraction=0.78)
    assert (len(d1), len(d2)) == (14, 5)


def test_split_grouped_by_replicas_and_number_of_contributors():
    """`group_replicas_in_split: true` (the reference's config/data/dnanet_rd.yaml; hid_dataset.py:344-412): the replicas of a
    mixture never straddle a split, every number of contributors is divided in the same proportion, seeded runs repeat.  (The
    reference pins its version on a cached dataset whose arrow file is a Git-LFS pointer here: structural checks only.)"""
    from collections import Counter
    from dnanet_amd.data import ArrayImage, rd_prefix
    names = [f"1{letter}{noc}_{well}_{rep:02d}.hid" for noc in (2, 3, 4, 5) for letter in "ABCDE" for rep, well in enumerate(("A01", "B02", "C03"))]
    ds = ArrayDataset([ArrayImage(data=None, path=n) for n in names], group_replicas_in_split=True)
    prefix = lambda im: rd_prefix(os.path.basename(im.path)[:-4])
    a, b = ds.split(0.8, seed=42)
    assert len(a) + len(b) == 60 and not {prefix(im) for im in a} & {prefix(im) for im in b}
    assert Counter(prefix(im)[-1] for im in b) == {"2": 3, "3": 3, "4": 3, "5": 3}          # one mixture of five per NoC
    a2, b2 = ds.split(0.8, seed=42)
    assert [im.path for im in a] == [im.path for im in a2] and [im.path for im in b] == [im.path for im in b2]
    assert [im.path for im in ds.split(0.8, seed=43)[1]] != [im.path for im in b]
    pairs = ds.split_k_fold(5, seed=42)
    assert sorted(im.path for _, t in pairs for im in t) == sorted(names)
    for train, test in pairs:
        assert not {prefix(im) for im in train} & {prefix(im) for im in test}
        assert Counter(prefix(im)[-1] for im in test) == {"2": 3, "3": 3, "4": 3, "5": 3}
    with pytest.raises(ValueError):
        ArrayDataset([ArrayImage(data=None, path="ladder_01.hid")], group_replicas_in_split=True).split(0.5)
    assert rd_prefix("1A2_A01_01") == "1A2"


def test_k_fold_covers_every_image_once_as_test():
    ds = ArrayDataset(list(range(23)))
    pairs = ds.split_k_fold(5, seed=42)
    assert len(pairs) == 5
    tests = [x for _, t in pairs for x in t]
    assert sorted(tests) == list(range(23))
    sizes = sorted(len(t) for _, t in pairs)
    assert sizes[-1] - sizes[0] <= 1
    for train, test in pairs:
        assert not set(train) & set(test) and len(train) + len(test) == 23
    again = ds.split_k_fold(5, seed=42)
    assert [sorted(t) for _, t in pairs] == [sorted(t) for _, t in again]


def test_synthetic_and_npy_datasets(tmp_path):
    ds = SyntheticDataset(4, rows=5, width=256, seed=3)
    assert len(ds) == 4 and ds[0].data.shape == (5, 256, 1) and ds[0].data.dtype == np.int16
    assert ds[0].annotation.image.shape == (5, 256, 1) and set(np.unique(ds[0].annotation.image)) <= {0, 1}
    for i, im in enumerate(ds):
        np.save(tmp_path / f"p{i}.npy", im.data)
        np.save(tmp_path / f"p{i}_annotation.npy", im.annotation.image)
    back = NpyDirDataset(str(tmp_path))
    assert len(back) == 4 and np.array_equal(back[2].data, ds[2].data)
    assert np.array_equal(back[2].annotation.image, ds[2].annotation.image)
    with pytest.raises(FileNotFoundError):
        NpyDirDataset(str(tmp_path / "missing"))


def test_config_registry(tmp_path, monkeypatch):
    monkeypatch.chdir(ROOT)
    ds = load_dataset("synthetic")
    assert len(ds) == 64 and ds[0].data.shape == (5, 4096, 1)
    metrics = load_evaluation_config("segmentation")
    assert list(metrics) == ["pixel_precision", "pixel_recall", "pixel_f1_score", "average_binary_iou_threshold_0.5"]
    assert metrics["average_binary_iou_threshold_0.5"].keywords == {"threshold": 0.5}
    # the shipped model configs parse into the model class with the precision they name (parameter tree only on the CPU)
    from dnanet_amd.config_io import load_model
    for name, (dtype, split) in {"unet": (0, 0), "unet_bf16": (1, 0), "unet_bf16x3": (1, 5)}.items():
        cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "models", name + ".yaml")))
        cfg["model"]["device"] = "cpu"
        path = tmp_path / (name + ".yaml")
        path.write_text(yaml.safe_dump(cfg))
        m = load_model(str(path))
        assert (m._model.dtype, m._model.split_levels, m._model.num_filters) == (dtype, split, 32)


def test_cli_flags_match_the_reference():
    import cross_validate
    import evaluate
    import train
    t = train.get_parser().parse_args("-d d -m m -t t -s 0.8 -o out -v 0.1 -c ck -rs 3".split())
    assert (t.data_config, t.model_config, t.training_config, t.split, t.output_dir, t.validation_config, t.checkpoint_dir,
            t.seed) == ("d", "m", "t", 0.8, "out", "0.1", "ck", 3)
    e = evaluate.get_parser().parse_args("-d d -m m -e e -o out -c ck -p -s 0.2 -rs 3".split())
    assert e.save_preds is True and e.split == 0.2 and e.evaluation_config == "e"
    c = cross_validate.get_parser().parse_args("-d d -m m -t t -e e -k 5".split())
    assert c.nr_folds == 5 and c.gpus == 1 and c.only_fold is None
    with pytest.raises(SystemExit):
        train.get_parser().parse_args(["-d", "d"])


@pytest.mark.gpu
def test_train_evaluate_cross_validate_end_to_end(tmp_path, monkeypatch, gpu_device):
    import cross_validate
    import evaluate
    import train
    monkeypatch.chdir(ROOT)
    cfg = tmp_path / "cfg"
    cfg.mkdir()
    (cfg / "data.yaml").write_text(yaml.safe_dump({"dataset": {"name": "synthetic", "n": 12, "rows": 5, "width": 256, "seed": 5}}))
    (cfg / "model.yaml").write_text(yaml.safe_dump({"model": {"name": "unet", "depth": 2, "kernel_size": [3, 5], "num_filters": 8,
                                                              "apply_allele_caller": False, "device": "cuda"}}))
    (cfg / "train.yaml").write_text(yaml.safe_dump({"training": {"batch_size": 4, "num_epochs": 2, "learning_rate": 1e-3,
                                                                 "tensorboard": False, "use_evaluation_metric": True,
                                                                 "use_scheduler": True, "save_best": False}}))
    out = tmp_path / "run"
    train.run(str(cfg / "data.yaml"), str(cfg / "model.yaml"), str(cfg / "train.yaml"), split=0.75, output_dir=str(out), seed=1)
    assert (out / "DNANet_UNet_checkpoint.pt").is_file() and (out / "config.yaml").is_file()
    ev = tmp_path / "eval"
    res = evaluate.run(str(cfg / "data.yaml"), str(cfg / "model.yaml"), os.path.join(ROOT, "config/evaluation/segmentation.yaml"),
                       output_dir=str(ev), checkpoint_dir=str(out), save_preds=True, split=0.25, seed=1)
    assert set(res) >= {"pixel_precision", "pixel_recall", "pixel_f1_score"}
    preds = json.load(open(ev / "predictions.json"))
    assert len(preds) == 3 and np.array(preds[0]["image"]).shape == (5, 256, 1)
    # the same fp32-trained checkpoint evaluated through the hi + lo split bf16 kernels: the same masks, hence the same metrics
    (cfg / "model_x3.yaml").write_text(yaml.safe_dump({"model": {"name": "unet", "depth": 2, "kernel_size": [3, 5], "num_filters": 8,
                                                                 "apply_allele_caller": False, "device": "cuda", "dtype": "bf16x3"}}))
    res3 = evaluate.run(str(cfg / "data.yaml"), str(cfg / "model_x3.yaml"), os.path.join(ROOT, "config/evaluation/segmentation.yaml"),
                        output_dir=str(tmp_path / "eval_x3"), checkpoint_dir=str(out), save_preds=False, split=0.25, seed=1)
    assert res3.keys() == res.keys() and all(abs(res3[k] - res[k]) < 2e-3 or (np.isnan(res3[k]) and np.isnan(res[k])) for k in res)
    cv = tmp_path / "cv"
    agg = cross_validate.run(str(cfg / "data.yaml"), str(cfg / "model.yaml"), str(cfg / "train.yaml"),
                             os.path.join(ROOT, "config/evaluation/segmentation.yaml"), 3, output_dir=str(cv))
    assert all((cv / f"fold_{i}" / "metrics.txt").is_file() for i in (1, 2, 3)) and (cv / "agg_metrics.txt").is_file()
    assert all(np.isfinite(v) or np.isnan(v) for v in agg.values())
    # folds as worker processes (one per GPU slot; on a one-GPU box both slots map to device 0): same split, same files
    cv2 = tmp_path / "cv2"
    cross_validate.run(str(cfg / "data.yaml"), str(cfg / "model.yaml"), str(cfg / "train.yaml"),
                       os.path.join(ROOT, "config/evaluation/segmentation.yaml"), 3, output_dir=str(cv2), gpus=2)
    assert all((cv2 / f"fold_{i}" / "metrics.txt").is_file() for i in (1, 2, 3)) and (cv2 / "agg_metrics.txt").is_file()
    for i in (1, 2, 3):       # deterministic folds and seeded models: the worker run reproduces the in-process run
        a = json.load(open(cv / f"fold_{i}" / "metrics.txt")); b = json.load(open(cv2 / f"fold_{i}" / "metrics.txt"))
        assert a.keys() == b.keys()
