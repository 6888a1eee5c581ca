"""``Prediction`` -- the boundary data type handed to the allele caller and the metrics
(same fields and JSON form as ``DNAnet/models/prediction.py:7-76``)."""
from __future__ import annotations

from typing import Any, Mapping, MutableMapping, Optional

import numpy as np


class Prediction:
    def __init__(self, classification: Optional[Mapping[str, float]] = None, image: Optional[np.ndarray] = None,
                 meta: Optional[MutableMapping[str, Any]] = None):
        self.classification = classification
        self.image = image            # (H, W, 1) fp32 probabilities
        self.meta = meta or dict()

    @property
    def label(self) -> str:
        if not self.classification:
            raise ValueError("Can't determine label without classification")
        return max(self.classification, key=self.classification.get)

    @property
    def confidence(self) -> float:
        if not self.classification:
            raise ValueError("Can't return a confidence score without classification")
        return self.classification[self.label]

    def to_dict(self):
        return {"classification": self.classification,
                "image": None if self.image is None else self.image.tolist(),
                "meta": self.meta}

    @classmethod
    def from_dict(cls, data):
        return cls(classification=data["classification"], image=np.array(data["image"]), meta=data["meta"])

    def __hash__(self):
        cls_items = tuple(self.classification.items()) if self.classification else ()
        return hash((cls_items, None if self.image is None else self.image.tobytes()))

    def __repr__(self) -> str:
        return f"Prediction(classification={self.classification}, image={self.image}, meta={self.meta})"

    __str__ = __repr__
