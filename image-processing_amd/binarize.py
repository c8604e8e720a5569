"""Feature binarisation thresholds on the device, with the reference's
function names and semantics (src/models/train_bernoulli_nb.py:331-385).

fit_thresholds / apply_thresholds take and return what the reference's do
(DataFrame in, {feature: threshold} / uint8 matrix out); the order statistics,
histograms and comparisons run in libfx (fx_column_thresholds_f64,
fx_binarize_ge_f64).  *_device variants work on tensors already in HBM, e.g.
the gathered feature table of an extraction run.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import engine

logger = logging.getLogger(__name__)

_METHODS = ("median", "quantile", "otsu")


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("image-processing_amd needs a ROCm device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def class_codes(y: Sequence) -> tuple:
    """Classes in the order of sorted(set(y)) and the int32 code of every row."""
    y = np.asarray(y)
    classes = sorted(set(y.tolist()))
    lookup = {c: i for i, c in enumerate(classes)}
    return classes, np.fromiter((lookup[v] for v in y.tolist()), dtype=np.int32, count=len(y))


def fit_thresholds_device(x: torch.Tensor, labels: Optional[torch.Tensor] = None, n_classes: int = 0,
                          method: str = "median", quantile: float = 0.5, strategy: str = "global",
                          return_flags: bool = False):
    """Thresholds (float64 [cols], host) of a float64 device table [rows, cols];
    `labels` = int32 class codes 0..n_classes-1 for strategy "per_class".  With
    return_flags also the per-column flags (1 = constant column)."""
    assert strategy in {"global", "per_class"}
    if method not in _METHODS:
        raise ValueError("Unknown binarize method")
    if x.shape[0] == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")  # values[0] of an empty column
    res = engine.column_thresholds(x, method, quantile, constant_rule=True)
    thr = res["thr"].cpu().numpy()
    flags = res["flags"].cpu().numpy()
    if method == "otsu" and np.any((flags & 2) != 0):
        raise ValueError("autodetected range of [nan, nan] is not finite")  # np.histogram on a column with NaN
    if strategy == "global" or labels is None or n_classes == 0:
        return (thr, flags) if return_flags else thr
    per_class: List[np.ndarray] = []
    for cls in range(n_classes):
        r = engine.column_thresholds(x, method, quantile, labels=labels, label=cls, constant_rule=False)
        if int(r["count"].item()) == 0:
            continue
        f = r["flags"].cpu().numpy()
        if method == "otsu" and np.any((f & 2) != 0):
            raise ValueError("autodetected range of [nan, nan] is not finite")
        per_class.append(r["thr"].cpu().numpy())
    out = thr.copy()
    if per_class:
        stack = np.stack(per_class)  # [classes, cols]
        for c in range(out.shape[0]):
            if not (flags[c] & 1):   # a constant column keeps its value
                out[c] = float(np.mean(list(stack[:, c])))
    return (out, flags) if return_flags else out


def fit_thresholds(X_train, y_train, method: str = "median", quantile: float = 0.5,
                   strategy: str = "global") -> Dict[str, float]:
    """Reference fit_thresholds (train_bernoulli_nb.py:331-376): thresholds of
    the training features only; 'global' or the mean of the class-wise ones."""
    assert strategy in {"global", "per_class"}
    feat_cols = [c for c in X_train.columns if c != "id"]
    if method not in _METHODS:
        raise ValueError("Unknown binarize method")
    if not feat_cols:
        return {}
    dev = _device()
    table = np.ascontiguousarray(X_train[feat_cols].astype(float).values, dtype=np.float64)
    x = torch.from_numpy(table).to(dev)
    labels, n_classes = None, 0
    if strategy == "per_class":
        classes, codes = class_codes(y_train)
        labels, n_classes = torch.from_numpy(codes).to(dev), len(classes)
    thr, flags = fit_thresholds_device(x, labels, n_classes, method, quantile, strategy, return_flags=True)
    out: Dict[str, float] = {}
    for c, v, f in zip(feat_cols, thr, flags):
        out[c] = float(v)
        if f & 1:
            logger.warning(f"Feature '{c}' is constant in training. Threshold set to {out[c]:.6g}")
    return out


def apply_thresholds_device(x: torch.Tensor, thr: torch.Tensor) -> torch.Tensor:
    """uint8 [rows, cols] on the device."""
    return engine.binarize_ge(x, thr)


def apply_thresholds(X, thresholds: Dict[str, float]) -> np.ndarray:
    """Reference apply_thresholds (train_bernoulli_nb.py:379-385): {0,1} uint8
    matrix (n_samples, n_features)."""
    feat_cols = [c for c in X.columns if c != "id"]
    Xv = np.ascontiguousarray(X[feat_cols].astype(float).values, dtype=np.float64)
    thr = np.array([thresholds[c] for c in feat_cols], dtype=float)
    if Xv.shape[0] == 0 or Xv.shape[1] == 0:
        return np.zeros(Xv.shape, dtype=np.uint8)
    dev = _device()
    return engine.binarize_ge(torch.from_numpy(Xv).to(dev), torch.from_numpy(thr).to(dev)).cpu().numpy()
