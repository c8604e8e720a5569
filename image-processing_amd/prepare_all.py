"""Point-cloud normalisation and farthest point sampling on MI355X behind the
reference's function names.

Mirrors the two numeric functions of reference src/preprocessing/prepare_all.py
that produce the fixed-size sets the PCA extractor reads: `center_and_scale`
(:199-217) and `farthest_point_sampling` (:250-293).  Same arguments, same
return values (bit for bit: the device kernels evaluate in float64 in NumPy's
operation order); the arithmetic runs in csrc/fx_fps.hip and there is no CPU
fallback for it.  The branch of farthest_point_sampling that does not compute
(fewer points than samples: tiling + RandomState.shuffle) and the draw of the
start index from the caller's RandomState stay on the host, so that the
caller's random stream advances exactly as in the reference.

Added next to the reference API: `center_and_scale_batch`,
`farthest_point_sampling_batch` and `normalize_and_sample`, which put many
clouds through one call.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np


def _device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the point-cloud preparation runs on the GPU only")
    return torch.device("cuda", torch.cuda.current_device())


def _pack(clouds: Sequence[np.ndarray], dtype) -> Tuple[np.ndarray, np.ndarray]:
    sizes = np.array([int(c.shape[0]) for c in clouds], dtype=np.int64)
    off = np.zeros(len(clouds) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    flat = np.empty((int(off[-1]), 3), dtype=dtype)
    for c, a, b in zip(clouds, off[:-1], off[1:]):
        flat[a:b] = c
    return flat, off


def center_and_scale_batch(clouds: Sequence[np.ndarray]) -> List[Tuple[np.ndarray, np.ndarray]]:
    """center_and_scale of many clouds in one call: a list of
    (normalized float32 [M, 3], centroid float32 [3])."""
    import torch
    from . import engine

    clouds = [np.asarray(c) for c in clouds]
    for c in clouds:
        if c.ndim != 2 or c.shape[1] != 3:
            raise ValueError(f"invalid shape {c.shape} (expected Mx3)")
    out: List[Optional[Tuple[np.ndarray, np.ndarray]]] = [None] * len(clouds)
    live = [k for k, c in enumerate(clouds) if c.size]
    for k, c in enumerate(clouds):
        if not c.size:        # the reference's early return
            out[k] = (c.astype(np.float32), np.zeros(3, dtype=np.float32))
    if live:
        wide = any(clouds[k].dtype != np.float32 for k in live)
        flat, off = _pack([clouds[k] for k in live], np.float64 if wide else np.float32)
        res = engine.center_scale(torch.from_numpy(flat).to(_device()), off)
        pts, cen = res["points"].cpu().numpy(), res["centroid"].cpu().numpy()
        for j, k in enumerate(live):
            out[k] = (pts[off[j]:off[j + 1]].copy(), cen[j].copy())
    return out  # type: ignore[return-value]


def center_and_scale(points: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Centre to the centroid and scale to the unit sphere (reference :199-217):
    (normalized float32 [M, 3], centroid float32 [3]).  Evaluated in float64,
    which is what the reference's reader hands to this function (:108-134); a
    float32 array is widened first (NumPy would average it in float32)."""
    return center_and_scale_batch([points])[0]


def _host_branch(M: int, n_samples: int, rng) -> Optional[np.ndarray]:
    """The outcomes of farthest_point_sampling that involve no distance
    computation (reference :264-274); None when sampling has to run."""
    if M == 0 or n_samples <= 0:
        return np.array([], dtype=np.int64)
    if n_samples >= M:
        reps = int(math.ceil(n_samples / M))
        base = np.tile(np.arange(M, dtype=np.int64), reps)[:n_samples]
        if rng is not None:
            rng.shuffle(base)
        return base.astype(np.int64)
    return None


def farthest_point_sampling_batch(clouds: Sequence[np.ndarray], n_samples: int,
                                  rngs: Optional[Sequence[Optional[np.random.RandomState]]] = None) -> List[np.ndarray]:
    """farthest_point_sampling of many clouds in one call; rngs[k] plays the
    role of the reference's `rng` argument for cloud k (None: start at index 0)."""
    import torch
    from . import engine

    clouds = [np.asarray(c) for c in clouds]
    rngs = list(rngs) if rngs is not None else [None] * len(clouds)
    out: List[Optional[np.ndarray]] = [None] * len(clouds)
    live, starts = [], []
    for k, c in enumerate(clouds):
        if c.ndim != 2 or c.shape[1] != 3:
            raise ValueError(f"invalid shape {c.shape} (expected Mx3)")
        M = int(c.shape[0])
        done = _host_branch(M, int(n_samples), rngs[k])
        if done is not None:
            out[k] = done
            continue
        live.append(k)
        starts.append(int(rngs[k].randint(0, M)) if rngs[k] is not None else 0)
    if live:
        for k in live:
            # the reference widens whatever it gets to float64; float32 holds the
            # production inputs (it is called with normalized.astype(np.float32))
            if clouds[k].dtype != np.float32 and not np.array_equal(clouds[k].astype(np.float32), clouds[k]):
                raise ValueError("farthest_point_sampling on the device takes float32 coordinates "
                                 "(or values exactly representable in float32)")
        flat, off = _pack([clouds[k] for k in live], np.float32)
        res = engine.fps(torch.from_numpy(flat).to(_device()), off, int(n_samples), np.asarray(starts), gather=False)
        idx = res["idx"].cpu().numpy()
        for j, k in enumerate(live):
            out[k] = idx[j].copy()
    return out  # type: ignore[return-value]


def farthest_point_sampling(points: np.ndarray, n_samples: int,
                            rng: Optional[np.random.RandomState] = None) -> np.ndarray:
    """Indices (int64 [n_samples]) of a farthest point sampling of `points`
    (reference :250-293)."""
    return farthest_point_sampling_batch([points], n_samples, [rng])[0]


def normalize_and_sample(clouds: Sequence[np.ndarray], npoints: int,
                         rngs: Optional[Sequence[Optional[np.random.RandomState]]] = None):
    """The numeric middle of the reference's per-file pipeline (:395-415) for a
    batch: center_and_scale, farthest_point_sampling on the float32 result,
    gather.  Returns a list of (sampled float32 [npoints, 3], centroid float32
    [3], indices int64 [npoints]); the normalized clouds stay on the device
    between the two steps."""
    import torch
    from . import engine

    clouds = [np.asarray(c) for c in clouds]
    rngs = list(rngs) if rngs is not None else [None] * len(clouds)
    out = [None] * len(clouds)
    live = [k for k, c in enumerate(clouds) if c.ndim == 2 and c.shape[1] == 3 and c.shape[0] > 0]
    if len(live) != len(clouds):
        raise ValueError("every cloud must be a non-empty Mx3 array")
    wide = any(c.dtype != np.float32 for c in clouds)
    flat, off = _pack(clouds, np.float64 if wide else np.float32)
    dev = _device()
    norm = engine.center_scale(torch.from_numpy(flat).to(dev), off)
    idx_host = {}
    sample_ids, starts = [], []
    for k, c in enumerate(clouds):
        M = int(c.shape[0])
        done = _host_branch(M, int(npoints), rngs[k])
        if done is not None:
            idx_host[k] = done
        else:
            sample_ids.append(k)
            starts.append(int(rngs[k].randint(0, M)) if rngs[k] is not None else 0)
    cen = norm["centroid"].cpu().numpy()
    if sample_ids:
        if len(sample_ids) == len(clouds):
            pts, soff = norm["points"], off
        else:
            rows = np.concatenate([np.arange(off[k], off[k + 1]) for k in sample_ids])
            pts = norm["points"][torch.from_numpy(rows).to(dev)]
            soff = np.zeros(len(sample_ids) + 1, dtype=np.int64)
            np.cumsum([off[k + 1] - off[k] for k in sample_ids], out=soff[1:])
        res = engine.fps(pts, soff, int(npoints), np.asarray(starts), gather=True)
        idx, sampled = res["idx"].cpu().numpy(), res["sampled"].cpu().numpy()
        for j, k in enumerate(sample_ids):
            out[k] = (sampled[j].copy(), cen[k].copy(), idx[j].copy())
    if idx_host:
        host_pts = norm["points"].cpu().numpy()
        for k, idx in idx_host.items():
            out[k] = (host_pts[off[k]:off[k + 1]][idx].astype(np.float32), cen[k].copy(), idx)
    return out
