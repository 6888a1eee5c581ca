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


METRICS = {"pixel_precision": pixel_precision, "pixel_recall": pixel_recall, "average_binary_iou": average_binary_iou,
           "pixel_f1_score": pixel_f1_score}
