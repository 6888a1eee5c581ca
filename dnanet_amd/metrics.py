"""Pixel-level segmentation metrics on (image, prediction) pairs.

Same definitions as the reference's ``DNAnet/evaluation/segmentation/pixel_metrics.py:9-74`` -- a pixel is predicted
positive when its probability is ``>= 0.5`` (``> threshold`` for the IoU), counts are pooled over all images for
precision / recall / F1, the IoU is averaged per image (images with an empty union add 0 but still count) -- computed
from one confusion count per image instead of one boolean pass per term.  Known answers of the reference's own test
(``tests/evaluation/segmentation/test_pixel_metrics.py:31-47``) are in ``tests/test_metrics.py``.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np


def _confusion(images: Sequence, predictions: Sequence, threshold: float = 0.5, strict: bool = False) -> np.ndarray:
    """[tp, fp, fn] summed over all pairs (annotation values other than 0 / 1 count for neither class, as in the
    reference, which compares ``== 1`` and ``== 0``)."""
    out = np.zeros(3, dtype=np.int64)
    for image, prediction in zip(images, predictions):
        device = _device_counts(image, prediction, threshold, strict)
        if device is not None:
            out += device
            continue
        truth = np.asarray(image.annotation.image)
        prob = np.asarray(prediction.image)
        pos = prob > threshold if strict else prob >= threshold
        out[0] += np.count_nonzero(pos & (truth == 1))
        out[1] += np.count_nonzero(pos & (truth == 0))
        out[2] += np.count_nonzero(~pos & (truth == 1))
    return out


def _device_counts(image, prediction, threshold: float, strict: bool):
    """{tp, fp, fn} that ``DNANet_UNet.predict_batch`` already counted on the GPU for exactly this (image, prediction)
    pair and threshold, else None."""
    tag = getattr(prediction, "_pixel_counts", None)
    if tag is None or tag[0] != id(image) or tag[1] != threshold:
        return None
    return np.asarray(tag[2][3:6] if strict else tag[2][0:3], dtype=np.int64)


def pixel_precision(images: Sequence, predictions: Sequence) -> float:
    tp, fp, _ = _confusion(images, predictions)
    return tp / (tp + fp)


def pixel_recall(images: Sequence, predictions: Sequence) -> float:
    tp, _, fn = _confusion(images, predictions)
    return tp / (tp + fn)


def pixel_f1_score(images: Sequence, predictions: Sequence) -> float:
    p, r = pixel_precision(images, predictions), pixel_recall(images, predictions)
    return 2 * p * r / (p + r) if p + r else 0


def average_binary_iou(images: Sequence, predictions: Sequence, threshold: float = 0.5) -> float:
    total, n = 0.0, 0
    for image, prediction in zip(images, predictions):
        device = _device_counts(image, prediction, threshold, strict=True)
        if device is not None:           # 0/1 masks: intersection = tp, union = tp + fp + fn
            inter, union = float(device[0]), float(device.sum())
        else:
            truth = np.asarray(image.annotation.image, dtype=np.float64)
            pred = (np.asarray(prediction.image) > threshold).astype(np.float64)
            inter = float(np.sum(truth * pred))
            union = float(np.sum(truth) + np.sum(pred) - inter)
        if union > 0:
            total += inter / union
        n += 1
    return total / n


# ---- allele-level metrics: the AlleleCaller's output against the annotated alleles ------------------------------------
# Same definitions as the reference's ``DNAnet/evaluation/segmentation/allele_metrics.py:8-84`` with the per-locus detection
# thresholds and filters of ``DNAnet/evaluation/utils.py:10-126``: a called allele is the string "<locus>_<allele>"; true
# positives and totals are pooled over all (image, prediction) pairs; an RFU filter (one value, or the kit's low / high
# detection threshold per locus, or "between" the two) applies to the predictions, and to the annotations too unless they are
# donor alleles without heights (``called_alleles_manual`` in the meta).  Known answers of the reference's own test
# (``tests/evaluation/segmentation/test_allele_metrics.py:31-48``) are in ``tests/test_metrics.py``.
_DETECTION_THRESHOLDS = {        # (low, high) RFU -> loci: the kit's dye-dependent detection thresholds
    (45, 95): ("AMEL", "D3S1358", "D1S1656", "D2S441", "D10S1248", "D13S317", "Penta E"),
    (50, 140): ("D16S539", "D18S51", "D2S1338", "CSF1PO", "Penta D"),
    (45, 85): ("TH01", "vWA", "D21S11", "D7S820", "D5S818", "TPOX"),
    (80, 135): ("D8S1179", "D12S391", "D19S433", "SE33", "D22S1045"),
    (40, 95): ("DYS391", "FGA", "DYS576", "DYS570"),
}
THRESHOLDS_PER_LOCUS = {locus: {"low": lo, "high": hi} for (lo, hi), loci in _DETECTION_THRESHOLDS.items() for locus in loci}


def min_rfu_per_locus(locus=None, low_or_high=None, min_rfu=None) -> dict:
    """locus -> None (no filter) | a minimum RFU | {"low", "high"} (keep what lies strictly between), for one locus or all."""
    if locus and locus not in THRESHOLDS_PER_LOCUS:
        raise ValueError(f"Unknown locus {locus} found when trying to set locus-specific threshold.")
    if low_or_high and low_or_high not in ("low", "high", "between"):
        raise ValueError(f"Unexpected argument for `low_or_high`, found {low_or_high}, expected `low`, `high` or `between` (or None).")
    loci = [locus] if locus else list(THRESHOLDS_PER_LOCUS)
    if low_or_high == "between":
        return {name: THRESHOLDS_PER_LOCUS[name] for name in loci}
    if low_or_high:
        return {name: THRESHOLDS_PER_LOCUS[name][low_or_high] for name in loci}
    return {name: min_rfu for name in loci}


def locus_allele_names(markers: Sequence, locus=None, min_rfu=None, low_or_high=None) -> set:
    """``[Marker(D5, [13, 15.1]), Marker(D7, [14])] -> {"D5_13", "D5_15.1", "D7_14"}``, filtered by locus and RFU.  Markers may
    be ``Marker`` objects or their dict form; a name listed twice is an error, as in the reference."""
    limits = min_rfu_per_locus(locus, low_or_high, min_rfu)
    names = []
    for marker in markers:
        m_name = marker["name"] if isinstance(marker, dict) else marker.name
        alleles = marker["alleles"] if isinstance(marker, dict) else marker.alleles
        for allele in alleles:
            if locus and m_name != locus:
                continue
            a_name, height = (allele["name"], allele["height"]) if isinstance(allele, dict) else (allele.name, allele.height)
            limit = limits[m_name]
            if limit:
                if height is None:
                    raise ValueError("Found rfu threshold, but no allele height to compare with. Set `min_rfu` and `low_or_high` "
                                     "to None to discard threshold.")
                if isinstance(limit, dict):
                    if not limit["low"] < height < limit["high"]:
                        continue
                elif height < limit:
                    continue
            names.append(f"{m_name}_{a_name}")
    unique = set(names)
    if len(unique) != len(names):
        raise ValueError(f"Found non-unique locus-allele combinations for {markers}")
    return unique


def _allele_counts(images: Sequence, predictions: Sequence, locus_name, low_or_high_threshold, min_rfu) -> Tuple[int, int, int]:
    """(true positives, predicted alleles, annotated alleles) pooled over the pairs."""
    donor = "called_alleles_manual" in images[0].meta            # donor alleles carry no peak heights: never filtered
    tp = n_pred = n_true = 0
    for image, prediction in zip(images, predictions):
        pred = locus_allele_names(prediction.meta["called_alleles"], locus_name, min_rfu, low_or_high_threshold)
        true = locus_allele_names(image.meta["called_alleles"], locus_name, None if donor else min_rfu,
                                  None if donor else low_or_high_threshold)
        tp += len(pred & true)
        n_pred += len(pred)
        n_true += len(true)
    return tp, n_pred, n_true


def allele_precision(images: Sequence, predictions: Sequence, locus_name=None, low_or_high_threshold=None, min_rfu=None) -> float:
    tp, n_pred, _ = _allele_counts(images, predictions, locus_name, low_or_high_threshold, min_rfu)
    return tp / n_pred if n_pred else 0


def allele_recall(images: Sequence, predictions: Sequence, locus_name=None, low_or_high_threshold=None, min_rfu=None) -> float:
    tp, _, n_true = _allele_counts(images, predictions, locus_name, low_or_high_threshold, min_rfu)
    return tp / n_true if n_true else 0


def allele_f1_score(images: Sequence, predictions: Sequence) -> float:
    tp, n_pred, n_true = _allele_counts(images, predictions, None, None, None)
    p, r = (tp / n_pred if n_pred else 0), (tp / n_true if n_true else 0)
    return 2 * p * r / (p + r) if p + r else 0


METRICS = {"pixel_precision": pixel_precision, "pixel_recall": pixel_recall, "average_binary_iou": average_binary_iou,
           "pixel_f1_score": pixel_f1_score, "allele_precision": allele_precision, "allele_recall": allele_recall,
           "allele_f1_score": allele_f1_score}
