"""Arrow cache of parsed profiles (reference DNAnet/data/caching.py): written the way the reference writes
it (Hugging Face ``datasets``, same columns and feature types as tests/resources/cached_hid_dataset_M1/dataset_info.json
of the reference), read back with pyarrow alone."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

datasets = pytest.importorskip("datasets")


def _images(n, panel):
    from dnanet_amd.data import Allele, Annotation, ArrayImage, Marker
    from dnanet_amd.synthetic import make_profiles
    x, t = make_profiles(n, 5, 4096, seed=8)
    out = []
    for i in range(n):
        called = [Marker(0, "D3S1358", [Allele("15", height=812.0 + i), Allele("16", height=90.0)])]
        out.append(ArrayImage(data=x[i], annotation=Annotation(image=t[i]), path=f"/data/{1 + i % 5}A{2 + i % 4}_A0{i}_01.hid",
                              meta={"called_alleles": called}, _scaler=np.linspace(65, 475, 4096) + 0.01 * i, _panel=panel))
    return out


def test_cache_round_trip_matches_reference_schema(tmp_path):
    from dnanet_amd.caching import ArrowCacheDataset, write_cache
    from dnanet_amd.data import Panel
    panel = Panel(os.path.join(GOLDEN_DIR, "panel.xml"))
    images = _images(5, panel)
    cache = tmp_path / "cache"
    write_cache(cache, images)
    # the directory is what the reference's load_from_disk expects, with the reference's feature schema
    info = json.load(open(cache / "dataset_info.json"))["features"]
    assert info["images"] == {"shape": [5, 4096, 1], "dtype": "int16", "_type": "Array3D"}
    assert info["annotations"] == {"shape": [5, 4096, 1], "dtype": "int8", "_type": "Array3D"}
    assert info["scalers"]["feature"] == {"dtype": "float64", "_type": "Value"} and info["hid_paths"]["dtype"] == "string"
    assert [d["filename"] for d in json.load(open(cache / "state.json"))["_data_files"]] == ["data-00000-of-00001.arrow"]
    hf = datasets.load_from_disk(str(cache))
    assert len(hf) == 5
    ds = ArrowCacheDataset(cache)
    assert len(ds) == 5 and ds.block_inputs.shape == (5, 5, 4096, 1) and ds.block_inputs.dtype == np.int16
    assert ds.block_targets.dtype == np.int8 and ds.block_inputs.flags["C_CONTIGUOUS"]
    for a, b in zip(images, ds):
        assert np.array_equal(a.data, b.data) and b.data.dtype == np.int16
        assert np.array_equal(a.annotation.image, b.annotation.image) and b.annotation.image.dtype == np.int8
        assert b.path == a.path and np.array_equal(a._scaler, b._scaler)
        assert b.meta["called_alleles"][0].alleles[0].height == a.meta["called_alleles"][0].alleles[0].height
        assert b.meta["noc"] == f"{a.path.split('/')[-1][2]}p"
        assert np.shares_memory(b.data, ds.block_inputs)                 # views into the one block
    assert ds[0]._panel is ds[1]._panel                                   # one Panel object per kit
    assert ds[0]._panel.dye_bp_to_allele_mapping == panel.dye_bp_to_allele_mapping
    assert len(ArrowCacheDataset(cache, limit=3)) == 3


def test_directory_of_caches_and_errors(tmp_path):
    from dnanet_amd.caching import ArrowCacheDataset, write_cache
    from dnanet_amd.data import Panel
    panel = Panel(os.path.join(GOLDEN_DIR, "panel.xml"))
    images = _images(4, panel)
    write_cache(tmp_path / "all" / "part_a", images[:3])
    write_cache(tmp_path / "all" / "part_b", images[3:])
    ds = ArrowCacheDataset(tmp_path / "all")
    assert [im.path for im in ds] == [im.path for im in images]
    assert len(ArrowCacheDataset(tmp_path / "all", limit=2)) == 2
    with pytest.raises(FileNotFoundError):
        ArrowCacheDataset(tmp_path / "nowhere")
    # the reference checkout's own cache fixture: state.json is there, the arrow blob is a missing large file
    broken = tmp_path / "broken"
    broken.mkdir()
    (broken / "state.json").write_text(json.dumps({"_data_files": [{"filename": "data-00000-of-00001.arrow"}]}))
    (broken / "x.arrow").write_text("")
    with pytest.raises(FileNotFoundError, match="missing"):
        ArrowCacheDataset(broken)


def test_cached_dataset_loads_through_the_data_config(tmp_path):
    from dnanet_amd.caching import write_cache
    from dnanet_amd.config_io import load_dataset
    from dnanet_amd.data import Panel
    write_cache(tmp_path / "c", _images(3, Panel(os.path.join(GOLDEN_DIR, "panel.xml"))))
    cfg = tmp_path / "cached.yaml"
    cfg.write_text(f"dataset:\n  name: arrow_cache\n  cache_path: {tmp_path / 'c'}\n  limit: 2\n")
    assert len(load_dataset(str(cfg))) == 2


def test_reference_data_config_name_hid_reads_the_cache(tmp_path):
    """The reference's own data config (dataset name `hid`, its parser keys and all) works when it points at a cache:
    `use_cache: true` + `cache_path`; without a cache the build says that parsing .hid files is the reference's job."""
    import yaml
    from conftest import ROOT
    from dnanet_amd.caching import write_cache
    from dnanet_amd.config_io import load_dataset
    from dnanet_amd.data import Panel
    panel = Panel(os.path.join(GOLDEN_DIR, "panel.xml"))
    images = _images(8, panel)
    write_cache(tmp_path / "cache", images)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "data", "dnanet_rd.yaml")))
    assert cfg["dataset"]["name"] == "hid" and cfg["dataset"]["group_replicas_in_split"] is True
    cfg["dataset"]["cache_path"] = str(tmp_path / "cache")
    (tmp_path / "data.yaml").write_text(yaml.safe_dump(cfg))
    ds = load_dataset(str(tmp_path / "data.yaml"))
    assert len(ds) == 8 and ds.group_replicas_in_split
    a, b = ds.split(0.5, seed=1)
    stem = lambda im: os.path.basename(im.path).split("_")[0]
    assert len(a) + len(b) == 8 and not {stem(im) for im in a} & {stem(im) for im in b}
    cfg["dataset"]["use_cache"] = False
    (tmp_path / "raw.yaml").write_text(yaml.safe_dump(cfg))
    with pytest.raises(NotImplementedError, match="arrow cache"):
        load_dataset(str(tmp_path / "raw.yaml"))


def test_noc_from_rd_file_name_follows_the_reference_regex():
    """``get_noc_from_rd_file_name`` (``DNAnet/utils.py:35-40,62-70``): the first three characters must look like '1A2' with a
    mixture letter A-F; nothing else about the name matters."""
    from dnanet_amd.caching import noc_from_rd_file_name
    assert noc_from_rd_file_name("1A2_A01_01.hid") == "2p"
    assert noc_from_rd_file_name("1A2.hid") == "2p"                # no '_' parts needed
    assert noc_from_rd_file_name("/some/dir/3F5_H12_07.hid") == "5p"
    assert noc_from_rd_file_name("1G2_A01_01.hid") is None          # G is not a mixture letter of the R&D set
    assert noc_from_rd_file_name("Ladder_A01.hid") is None
