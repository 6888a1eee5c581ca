"""Nearest-base-pair allele caller: the consumer of the U-Net's probability maps.

Same contract as ``NearestBasePairCaller`` (reference ``DNAnet/allele_callers.py:38-125``): per dye row,
pixels with probability >= 0.5 (``:80``) are grouped into runs of consecutive scan columns (``:87``); a
run's mean base pair (``:90``) selects the panel allele with the nearest base pair on that dye
(``:122-123``, first-listed wins a tie); the allele's height is the highest RFU seen in any of its runs
(``:94-98``).  This implementation finds runs with one vectorised edge detection per dye and the nearest
allele by binary search on the dye's sorted base-pair table instead of Python loops over pixels/alleles.

``call_alleles_batch`` is the batched device path used by ``predict_batch``: the run finding, mean base pair, nearest
allele and peak height of every run of every profile are computed by ``dn_call_alleles`` on the probability maps while
they are still in HBM; the host only turns the small per-allele tables into ``Marker`` objects.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Tuple

import numpy as np

from .data import Allele, Marker, Panel
from .prediction import Prediction

LOGGER = logging.getLogger("dnanet")

NON_AUTOSOMAL_MARKERS = ["AMEL", "DYS391", "DYS576", "DYS570"]   # allele_callers.py:15


class _DyeTable:
    """Sorted base pairs of one dye with their (marker, allele) names and panel order."""

    def __init__(self, mapping: Dict[float, Tuple[str, str]]):
        keys = list(mapping.keys())
        bps = np.asarray(keys, dtype=np.float64)
        self.order = np.argsort(bps, kind="stable")     # order[j] = position in the panel (tie-break)
        self.bps = bps[self.order]
        self.names = [mapping[keys[i]] for i in self.order]

    def nearest(self, bp: float) -> Tuple[str, str]:
        i = int(np.searchsorted(self.bps, bp))
        cands = [j for j in (i - 1, i) if 0 <= j < len(self.bps)]
        # equal base pairs sit next to each other after the stable sort; widen to all equals
        best = min(cands, key=lambda j: (abs(self.bps[j] - bp), self.order[j]))
        d = abs(self.bps[best] - bp)
        j = best
        while j - 1 >= 0 and abs(self.bps[j - 1] - bp) == d:
            j -= 1
            if self.order[j] < self.order[best]:
                best = j
        j = best
        while j + 1 < len(self.bps) and abs(self.bps[j + 1] - bp) == d:
            j += 1
            if self.order[j] < self.order[best]:
                best = j
        return self.names[best]


class NearestBasePairCaller:
    def __init__(self):
        self._tables: Dict[int, Dict[int, _DyeTable]] = {}

    def _table(self, panel: Panel, dye: int) -> _DyeTable:
        per_panel = self._tables.setdefault(id(panel), {})
        if dye not in per_panel:
            per_panel[dye] = _DyeTable(panel.dye_bp_to_allele_mapping[dye])
        return per_panel[dye]

    def call_alleles(self, image, prediction: Prediction) -> Prediction:
        called = self.translate_pixels_to_alleles(image._scaler[np.newaxis, :], prediction.image, image.data,
                                                  image._panel)
        if "called_alleles_manual" in image.meta:
            # ground-truth annotations hold no DYS / AMEL markers (allele_callers.py:54-58)
            called = [m for m in called if m.name not in NON_AUTOSOMAL_MARKERS]
        prediction.meta["called_alleles"] = called
        return prediction

    def translate_pixels_to_alleles(self, scaler: np.ndarray, prediction_image: np.ndarray, image: np.ndarray,
                                    panel: Panel) -> List[Marker]:
        scaler = np.asarray(scaler)
        bp_of_col = scaler.reshape(-1)
        loci: Dict[Tuple[int, str], Dict[str, None]] = {}
        rfus: Dict[Tuple[str, str], int] = {}
        for dye, dye_pred in enumerate(prediction_image):
            positive = np.asarray(dye_pred)[:, 0] >= 0.5
            if not positive.any():
                LOGGER.warning(f"No predictions present in dye row {dye}")
                continue
            edges = np.diff(np.concatenate(([0], positive.view(np.int8), [0])))
            starts, stops = np.flatnonzero(edges == 1), np.flatnonzero(edges == -1)
            table = self._table(panel, dye)
            rfu_row = np.asarray(image)[dye, :, 0]
            for a, b in zip(starts, stops):
                mean_bp = np.mean(bp_of_col[a:b])
                marker_name, allele_name = table.nearest(float(mean_bp))
                loci.setdefault((dye, marker_name), {})[allele_name] = None
                key = (marker_name, allele_name)
                rfus[key] = int(max(rfus.get(key, 0), rfu_row[a:b].max()))
        return [Marker(dye, marker_name, alleles=[Allele(name=a, height=rfus[(marker_name, a)]) for a in alleles])
                for (dye, marker_name), alleles in loci.items()]

    # ---- batched device path ---------------------------------------------------------------------
    def _device_tables(self, panel: Panel, n_rows: int, device):
        """Concatenated per-dye tables of a panel on the device: (bp, order, offsets, amax)."""
        import torch
        cache = self.__dict__.setdefault("_dev_tables", {})
        key = (id(panel), n_rows, str(device))
        if key not in cache:
            bps, orders, offs = [], [], [0]
            for dye in range(n_rows):
                if dye in panel.dye_bp_to_allele_mapping:
                    t = self._table(panel, dye)
                    bps.append(t.bps); orders.append(np.asarray(t.order, dtype=np.int32))
                    offs.append(offs[-1] + len(t.bps))
                else:
                    offs.append(offs[-1])
            amax = max(1, max(b - a for a, b in zip(offs[:-1], offs[1:])))
            cache[key] = (torch.from_numpy(np.concatenate(bps) if bps else np.zeros(1)).to(device),
                          torch.from_numpy(np.concatenate(orders) if orders else np.zeros(1, np.int32)).to(device),
                          torch.tensor(offs, dtype=torch.int32, device=device), amax)
        return cache[key]

    def call_alleles_batch(self, images, predictions, probs, rfu):
        """``call_alleles`` for a whole batch.  ``probs`` / ``rfu``: (N,1,H,W) fp32 device tensors (probabilities of this
        batch, the raw inputs they were computed from).  Profiles are grouped by panel; every group is one launch."""
        return self.finish_alleles_batch(images, predictions, self.launch_alleles_batch(images, probs, rfu))

    def launch_alleles_batch(self, images, probs, rfu) -> list:
        """Device half of ``call_alleles_batch``: queues the allele kernel(s) and the copy of their small tables into pinned
        host memory; waits for nothing.  Returns the handle ``finish_alleles_batch`` takes."""
        import torch
        from . import hip
        n, _, h, w = probs.shape
        groups: Dict[int, List[int]] = {}
        for i, im in enumerate(images):
            groups.setdefault(id(im._panel), []).append(i)
        handle = []
        for idx in groups.values():
            panel = images[idx[0]]._panel
            tb, to, toff, amax = self._device_tables(panel, h, probs.device)
            whole = len(idx) == n
            p_sel = probs if whole else probs[idx].contiguous()
            x_sel = rfu if whole else rfu[idx].contiguous()
            scalers = [images[i]._scaler for i in idx]
            shared = all(sc is scalers[0] for sc in scalers)
            sc_np = np.asarray(scalers[0], dtype=np.float64).reshape(1, -1) if shared else \
                np.stack([np.asarray(sc, dtype=np.float64).reshape(-1) for sc in scalers])
            if sc_np.shape[1] != w:
                raise ValueError(f"scaler has {sc_np.shape[1]} entries for {w} scan columns")
            sc_cache = self.__dict__.setdefault("_dev_scalers", {})
            sc_key = (id(scalers[0]), str(probs.device)) if shared else None
            if sc_key in sc_cache and sc_cache[sc_key][0] is scalers[0]:
                sc_dev = sc_cache[sc_key][1]                 # the base-pair axis every profile of a dataset shares: sent once
            else:
                sc_dev = torch.from_numpy(np.ascontiguousarray(sc_np)).to(probs.device)
                if shared:
                    if len(sc_cache) >= 16:
                        sc_cache.clear()
                    sc_cache[sc_key] = (scalers[0], sc_dev)
            out = torch.empty((2, len(idx), h, amax), dtype=torch.int32, device=probs.device)
            hip.check(hip.lib.dn_call_alleles(hip.ptr(p_sel), hip.ptr(x_sel), hip.ptr(sc_dev), 0 if shared else w, hip.ptr(tb),
                                              hip.ptr(to), hip.ptr(toff), len(idx), h, w, amax, hip.ptr(out[0]), hip.ptr(out[1]),
                                              hip.stream_ptr()))
            host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
            host.copy_(out, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            handle.append((idx, panel, h, host, event, (p_sel, x_sel, sc_dev, out)))     # device operands alive until the copy is done
        return handle

    def finish_alleles_batch(self, images, predictions, handle: list):
        """Host half: the per-(dye, allele) tables -> ``Marker`` / ``Allele`` objects in the reference's order."""
        for idx, panel, h, host, event, _keep in handle:
            event.synchronize()
            peak, first = host.numpy()
            tables = {dye: self._table(panel, dye) for dye in range(h) if dye in panel.dye_bp_to_allele_mapping}
            names = [tables[d].names if d in tables else [] for d in range(h)]
            # A (marker, allele) name that the panel lists once has exactly one (dye, table row): its height is that row's
            # peak and no second dictionary is needed (every kit table seen so far; the general path below keeps the
            # reference's max over equal names)
            all_names = [nm for t in tables.values() for nm in t.names]
            unique_names = len(set(all_names)) == len(all_names)
            for k, i in enumerate(idx):
                dyes, js = np.nonzero(first[k] != ABSENT)
                order = np.lexsort((first[k][dyes, js], dyes))       # dye rows in order, runs left to right
                dyes, js = dyes[order], js[order]
                if unique_names:
                    by_locus: Dict[Tuple[int, str], List[Allele]] = {}
                    for d, j, p in zip(dyes.tolist(), js.tolist(), peak[k][dyes, js].tolist()):
                        marker_name, allele_name = names[d][j]
                        slot = by_locus.get((d, marker_name))
                        if slot is None:
                            slot = by_locus[(d, marker_name)] = []
                        slot.append(Allele(name=allele_name, height=p if p > 0 else 0))
                    called = [Marker(dye, marker_name, alleles=alleles) for (dye, marker_name), alleles in by_locus.items()]
                else:
                    loci: Dict[Tuple[int, str], Dict[str, None]] = {}
                    rfus: Dict[Tuple[str, str], int] = {}
                    for d, j in zip(dyes.tolist(), js.tolist()):
                        marker_name, allele_name = names[d][j]
                        loci.setdefault((d, marker_name), {})[allele_name] = None
                        key = (marker_name, allele_name)
                        rfus[key] = max(rfus.get(key, 0), int(peak[k, d, j]))
                    called = [Marker(dye, marker_name, alleles=[Allele(name=a, height=rfus[(marker_name, a)]) for a in alleles])
                              for (dye, marker_name), alleles in loci.items()]
                if "called_alleles_manual" in images[i].meta:
                    called = [m for m in called if m.name not in NON_AUTOSOMAL_MARKERS]
                predictions[i].meta["called_alleles"] = called
        return predictions


ABSENT = 0x7F7F7F7F      # DN_ALLELE_ABSENT (include/dnanet_hip.h)
