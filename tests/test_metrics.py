"""Pixel metrics against the known answers the reference's own test pins
(tests/evaluation/segmentation/test_pixel_metrics.py:31-95 of the reference)."""
import os

import numpy as np
import pytest

from dnanet_amd.data import Annotation, ArrayImage
from dnanet_amd.metrics import average_binary_iou, pixel_f1_score, pixel_precision, pixel_recall
from dnanet_amd.prediction import Prediction

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _image(mask):
    return ArrayImage(data=np.zeros(mask.shape, np.int16), annotation=Annotation(image=mask))


def test_toy_image_known_answers():
    image = _image(np.array([[1, 1, 1], [0, 1, 0], [1, 0, 0]]))
    prediction = Prediction(image=np.array([[0.6, 0.8, 0.1], [0.2, 0.7, 0.6], [0.9, 0.2, 0.7]]))
    tp, fn, fp = 4, 1, 2
    precision, recall = tp / (tp + fp), tp / (tp + fn)
    assert pixel_precision([image], [prediction]) == precision
    assert pixel_recall([image], [prediction]) == recall
    assert pixel_f1_score([image], [prediction]) == 2 * (precision * recall / (precision + recall))
    assert average_binary_iou([image], [prediction]) == tp / (tp + fn + fp)


@pytest.fixture
def real_mask():
    # the analyst's annotation of profile 1A2_A01_01 (638 positive pixels), the array the reference's test loads
    return np.load(os.path.join(GOLDEN, "allele_inputs_1A2_A01_01.npz"))["mask"].astype(np.float64)


def test_real_annotation_known_answers(real_mask):
    image = _image(real_mask.copy())
    assert int(real_mask.sum()) == 638
    for metric in (pixel_precision, pixel_recall, pixel_f1_score, average_binary_iou):
        assert metric([image], [Prediction(image=real_mask.copy())]) == 1.0
    noisy = real_mask.copy()
    noisy[0, :100, 0] = 1                    # 100 false positives
    assert pixel_precision([image], [Prediction(image=noisy)]) == pytest.approx(0.865, rel=1e-3)
    assert pixel_f1_score([image], [Prediction(image=noisy)]) == pytest.approx(0.927, rel=1e-3)
    assert average_binary_iou([image], [Prediction(image=noisy)]) == pytest.approx(0.865, rel=1e-3)
    fewer = real_mask.copy()
    ones = np.where(fewer[0, :, 0] == 1)[0]
    fewer[0, np.random.default_rng(0).choice(ones, size=100, replace=False), 0] = 0   # 100 false negatives
    assert pixel_recall([image], [Prediction(image=fewer)]) == pytest.approx(0.844, rel=1e-3)


def test_iou_counts_empty_images():
    empty = _image(np.zeros((2, 4, 1)))
    full = _image(np.ones((2, 4, 1)))
    preds = [Prediction(image=np.zeros((2, 4, 1))), Prediction(image=np.ones((2, 4, 1)))]
    assert average_binary_iou([empty, full], preds) == 0.5    # empty union adds 0 but is counted


# ---- allele-level metrics --------------------------------------------------------------------------------------------
def _pair(annotated, called, manual=False):
    from dnanet_amd.data import ArrayImage
    from dnanet_amd.prediction import Prediction
    meta = {"called_alleles": annotated}
    if manual:
        meta["called_alleles_manual"] = annotated
    return ArrayImage(data=None, meta=meta), Prediction(image=None, meta={"called_alleles": called})


def test_allele_metrics_known_answers_of_the_reference_test():
    """tests/evaluation/segmentation/test_allele_metrics.py:31-48 of the reference: two annotated markers with two alleles each,
    the prediction holds the first marker only -> tp 2, fn 2, fp 0."""
    from dnanet_amd.data import Allele, Marker
    from dnanet_amd.metrics import allele_f1_score, allele_precision, allele_recall
    m1 = Marker(name="AMEL", dye_row=0, alleles=[Allele("13"), Allele("15.1")])
    m2 = Marker(name="Penta E", dye_row=0, alleles=[Allele("14"), Allele("15.1")])
    image, prediction = _pair([m1, m2], [m1])
    assert allele_precision([image], [prediction]) == 1.0
    assert allele_recall([image], [prediction]) == 0.5
    assert allele_f1_score([image], [prediction]) == 2 * (1.0 * 0.5) / 1.5
    # nothing called, nothing annotated: 0, not a division error
    empty_i, empty_p = _pair([], [])
    assert allele_precision([empty_i], [empty_p]) == 0 and allele_recall([empty_i], [empty_p]) == 0
    assert allele_f1_score([empty_i], [empty_p]) == 0


def test_allele_metric_thresholds_and_errors():
    """The per-locus RFU filters (DNAnet/evaluation/utils.py:10-126 restated).  No fixture or test of the reference covers them
    and its module cannot be imported here without a missing third-party package: hand-derived answers, parity unpinned."""
    from dnanet_amd.data import Allele, Marker
    from dnanet_amd.metrics import (THRESHOLDS_PER_LOCUS, allele_precision, allele_recall, locus_allele_names,
                                    min_rfu_per_locus)
    assert len(THRESHOLDS_PER_LOCUS) == 27 and THRESHOLDS_PER_LOCUS["D18S51"] == {"low": 50, "high": 140}
    assert min_rfu_per_locus("TH01", "high") == {"TH01": 85} and min_rfu_per_locus("TH01", None, 30) == {"TH01": 30}
    assert min_rfu_per_locus(None, "between")["SE33"] == {"low": 80, "high": 135}
    with pytest.raises(ValueError):
        min_rfu_per_locus("NOPE")
    with pytest.raises(ValueError):
        min_rfu_per_locus(None, "medium")
    called = [Marker(name="TH01", dye_row=1, alleles=[Allele("6", height=40), Allele("7", height=60), Allele("9.3", height=85),
                                                      Allele("10", height=3000)]),
              Marker(name="FGA", dye_row=3, alleles=[Allele("20", height=41), Allele("21", height=500)])]
    assert locus_allele_names(called) == {"TH01_6", "TH01_7", "TH01_9.3", "TH01_10", "FGA_20", "FGA_21"}
    assert locus_allele_names(called, low_or_high="low") == {"TH01_7", "TH01_9.3", "TH01_10", "FGA_20", "FGA_21"}       # >= 45 / >= 40
    assert locus_allele_names(called, low_or_high="high") == {"TH01_9.3", "TH01_10", "FGA_21"}                         # >= 85 / >= 95
    assert locus_allele_names(called, low_or_high="between") == {"TH01_7", "FGA_20"}                                   # strictly inside
    assert locus_allele_names(called, locus="FGA", min_rfu=100) == {"FGA_21"}
    as_dicts = [{"name": "TH01", "alleles": [{"name": "6", "height": 40}]}]
    assert locus_allele_names(as_dicts) == {"TH01_6"}
    with pytest.raises(ValueError):                      # a threshold but no height
        locus_allele_names([Marker(name="TH01", dye_row=1, alleles=[Allele("6")])], min_rfu=10)
    with pytest.raises(ValueError):                      # the same allele twice
        locus_allele_names([Marker(name="TH01", dye_row=1, alleles=[Allele("6"), Allele("6")])])
    # donor alleles (no heights) are never filtered; the predictions are
    donors = [Marker(name="TH01", dye_row=1, alleles=[Allele("7"), Allele("10")])]
    image, prediction = _pair(donors, called, manual=True)
    assert allele_recall([image], [prediction], low_or_high_threshold="high") == 0.5          # of {7, 10} only 10 is called >= 85
    assert allele_precision([image], [prediction], low_or_high_threshold="high") == 1 / 3
    # analyst annotations with heights are filtered like the predictions
    analyst = [Marker(name="TH01", dye_row=1, alleles=[Allele("7", height=61), Allele("10", height=2990)])]
    image, prediction = _pair(analyst, called)
    assert allele_recall([image], [prediction], low_or_high_threshold="between") == 1.0       # {7} vs {7, FGA_20}
    assert allele_precision([image], [prediction], low_or_high_threshold="between") == 0.5


def test_evaluation_config_with_allele_metrics():
    import os
    from conftest import ROOT
    from dnanet_amd.config_io import load_evaluation_config
    metrics = load_evaluation_config(os.path.join(ROOT, "config", "evaluation", "segmentation_alleles.yaml"))
    assert len(metrics) == 13 and "allele_recall_low_or_high_threshold_between" in metrics
    assert metrics["allele_precision_low_or_high_threshold_high"].keywords == {"low_or_high_threshold": "high"}


def test_allele_metrics_on_the_real_profiles_through_the_caller():
    """The reference's known-answer set-up (tests/test_allele_callers.py:7-61) carried through to the metrics: the analyst's
    annotation mask of a real profile fed as the "prediction" reproduces the annotated alleles, so every allele metric of the
    reference's evaluation config is 1 (or 0 where the RFU band holds no allele at all); dropping a dye row lowers recall only."""
    from conftest import GOLDEN_DIR, ROOT
    from dnanet_amd.allele_caller import NearestBasePairCaller
    from dnanet_amd.cli import write_metrics
    from dnanet_amd.config_io import load_evaluation_config
    from dnanet_amd.data import Panel
    panel = Panel(os.path.join(GOLDEN_DIR, "panel.xml"))
    caller, scaler = NearestBasePairCaller(), np.linspace(65, 475, 4096)
    images, exact, partial = [], [], []
    for stem in ("1A2_A01_01", "1A2_E01_13"):
        z = np.load(os.path.join(GOLDEN_DIR, f"allele_inputs_{stem}.npz"))
        rfu, mask = z["rfu"], z["mask"].astype(np.float32)
        truth = caller.translate_pixels_to_alleles(scaler[None], mask, rfu, panel)
        image = ArrayImage(data=rfu, annotation=Annotation(image=mask.astype(np.int8)), meta={"called_alleles": truth},
                           _scaler=scaler, _panel=panel)
        images.append(image)
        exact.append(caller.call_alleles(image, Prediction(image=mask, meta={})))
        cut = mask.copy()
        cut[0] = 0                                            # nothing predicted on the first dye row
        partial.append(caller.call_alleles(image, Prediction(image=cut, meta={})))
    metrics = load_evaluation_config(os.path.join(ROOT, "config", "evaluation", "segmentation_alleles.yaml"))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        full = write_metrics(metrics, images, exact, os.path.join(d, "m.txt"))
        part = write_metrics(metrics, images, partial, os.path.join(d, "p.txt"))
    allele = {k: v for k, v in full.items() if k.startswith("allele")}
    assert len(allele) == 9 and all(v in (0.0, 1.0) for v in allele.values()) and allele["allele_f1_score"] == 1.0
    assert allele["allele_precision"] == allele["allele_recall"] == 1.0
    assert part["allele_precision"] == 1.0 and 0.5 < part["allele_recall"] < 1.0
    assert full["pixel_f1_score"] == 1.0 and part["pixel_recall"] < 1.0


def test_human_analysis_model_scores_like_a_network():
    """``-m human_analysis``: the analyst's alleles as predictions (minus AMEL / DYS markers) against donor alleles."""
    from conftest import ROOT
    from dnanet_amd.config_io import load_model
    from dnanet_amd.data import Allele, Marker
    from dnanet_amd.metrics import allele_precision, allele_recall
    model = load_model(os.path.join(ROOT, "config", "models", "human_analysis.yaml"))
    donors = [Marker(name="TH01", dye_row=1, alleles=[Allele("7"), Allele("10")]), Marker(name="FGA", dye_row=3, alleles=[Allele("21")])]
    manual = [Marker(name="TH01", dye_row=1, alleles=[Allele("7", height=300), Allele("9.3", height=90)]),
              Marker(name="AMEL", dye_row=0, alleles=[Allele("X", height=900)])]
    image = ArrayImage(data=None, path="a.hid", meta={"called_alleles": donors, "called_alleles_manual": manual})
    (pred,) = model.predict_batch([image])
    assert pred.image is None and pred.meta["original_image_path"] == "a.hid"
    assert [m.name for m in pred.meta["called_alleles"]] == ["TH01"]
    assert allele_precision([image], [pred]) == 0.5 and allele_recall([image], [pred]) == 1 / 3
    with pytest.raises(ValueError):
        model.predict(ArrayImage(data=None, meta={}))
