"""Device-level batch API: torch tensors carry the buffers, libfx does the work.

Every function here enqueues HIP kernels on torch's current stream of the
tensor's device and returns torch tensors that live on that device.  Nothing is
computed on the host.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib

HU_COLS = 7
MOMENT_COLS = 24


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor (got {t.device}); this engine has no CPU path")


_initialised = set()


def _enter(device: torch.device) -> int:
    """Makes `device` current for HIP and returns torch's current stream handle on it."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    torch.cuda.set_device(idx)
    if idx not in _initialised:
        _lib.call("fx_init", idx)
        _initialised.add(idx)
    return int(torch.cuda.current_stream(idx).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def hu_batch(images: torch.Tensor, thresh: int = -1, on_value: int = 255, *, moments: bool = False,
             hu: bool = False, contour_bins: Optional[int] = None, contour_eig: bool = False,
             out: Optional[dict] = None, noalloc: bool = False) -> dict:
    """Log-Hu vectors of a batch of uint8 images [B, H, W] (rows may be padded:
    any strides with a unit inner stride are accepted).

    Returns {"hulog7": f32 [B,7]} plus, on request, "moments24" f64 [B,24]
    (the cv2.moments record), "hu7" f64 [B,7], "cpca" f32 [B, 9+contour_bins]
    (PCA descriptor of the contour pixels) and "ceig" f64 [B,6].

    `out` supplies preallocated output tensors under those names (contiguous,
    right shape and dtype); with noalloc=True the library entry point that
    performs no device allocation is used (fx_hu_batch_u8_noalloc: scratch in a
    built-in block, at most 65536 images per call, calls on one device must not
    overlap).
    """
    _require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 3:
        raise ValueError(f"images must be uint8 [B, H, W], got {images.dtype} {tuple(images.shape)}")
    if images.stride(2) != 1 and images.shape[2] > 1:
        raise ValueError("images must have a unit stride along the row")
    stream = _enter(images.device)
    B, H, W = images.shape
    dev = images.device
    nb = int(contour_bins) if contour_bins is not None else 0
    want = {"hulog7": ((B, HU_COLS), torch.float32)}
    if moments:
        want["moments24"] = ((B, MOMENT_COLS), torch.float64)
    if hu:
        want["hu7"] = ((B, HU_COLS), torch.float64)
    if contour_bins is not None:
        want["cpca"] = ((B, 9 + nb), torch.float32)
        if contour_eig:
            want["ceig"] = ((B, 6), torch.float64)
    if out is None:
        out = {k: torch.empty(shape, dtype=dt, device=dev) for k, (shape, dt) in want.items()}
    else:
        for k, (shape, dt) in want.items():
            t = out.get(k)
            if t is None or tuple(t.shape) != shape or t.dtype != dt or t.device != dev or not t.is_contiguous():
                raise ValueError(f"out[{k!r}] must be a contiguous {dt} tensor of shape {shape} on {dev}")
        out = {k: out[k] for k in want}
    if B == 0:
        return out
    pitch = images.stride(1) if H > 1 else W
    stride = images.stride(0) if B > 1 else pitch * H
    _lib.call("fx_hu_batch_u8_noalloc" if noalloc else "fx_hu_batch_u8", images.data_ptr(), B, H, W, pitch, stride,
              int(thresh), int(on_value),
              _ptr(out.get("moments24")), _ptr(out.get("hu7")), out["hulog7"].data_ptr(), nb,
              _ptr(out.get("cpca")), _ptr(out.get("ceig")), stream)
    return out


def hu_partial(bands: torch.Tensor, y_offset: int, thresh: int = -1, on_value: int = 255) -> torch.Tensor:
    """Exact raw moments of row bands [B, H, W] that start at row y_offset of
    taller images, in the tall image's coordinates: int64 [B, 10, 3] (three
    48-bit limbs per moment).  Bands of one image are combined by adding these."""
    _require_cuda(bands, "bands")
    if bands.dtype != torch.uint8 or bands.dim() != 3:
        raise ValueError(f"bands must be uint8 [B, H, W], got {bands.dtype} {tuple(bands.shape)}")
    if bands.stride(2) != 1 and bands.shape[2] > 1:
        raise ValueError("bands must have a unit stride along the row")
    stream = _enter(bands.device)
    B, H, W = bands.shape
    out = torch.empty((B, 10, 3), dtype=torch.int64, device=bands.device)
    if B == 0:
        return out
    pitch = bands.stride(1) if H > 1 else W
    stride = bands.stride(0) if B > 1 else pitch * H
    _lib.call("fx_hu_partial_u8", bands.data_ptr(), B, H, W, pitch, stride, int(thresh), int(on_value), int(y_offset),
              out.data_ptr(), stream)
    return out


def hu_from_partial(partial: torch.Tensor, *, moments: bool = False, hu: bool = False) -> dict:
    """Finalises summed limb arrays [B, 10, 3] (see hu_partial): {"hulog7"} and on request "moments24", "hu7"."""
    _require_cuda(partial, "partial")
    if partial.dtype != torch.int64 or partial.dim() != 3 or tuple(partial.shape[1:]) != (10, 3) \
            or not partial.is_contiguous():
        raise ValueError("partial must be contiguous int64 [B, 10, 3]")
    stream = _enter(partial.device)
    B = partial.shape[0]
    dev = partial.device
    out = {"hulog7": torch.empty((B, HU_COLS), dtype=torch.float32, device=dev)}
    if moments:
        out["moments24"] = torch.empty((B, MOMENT_COLS), dtype=torch.float64, device=dev)
    if hu:
        out["hu7"] = torch.empty((B, HU_COLS), dtype=torch.float64, device=dev)
    if B:
        _lib.call("fx_hu_from_partial", partial.data_ptr(), B, _ptr(out.get("moments24")), _ptr(out.get("hu7")),
                  out["hulog7"].data_ptr(), stream)
    return out


SIL_BACKENDS = {"pil": 0, "opencv": 1}   # FX_SIL_PIL, FX_SIL_OPENCV


def silhouette_batch(images: torch.Tensor, resolution: int = 224, crop_margin: float = 0.05,
                     preprocess_crop: bool = True, min_size: int = -1, *, info: bool = False,
                     backend: str = "opencv") -> dict:
    """Cleaned 0/255 silhouettes [B, resolution, resolution] of a batch of gray
    uint8 views [B, H, W] (Gaussian blur, Otsu, morphology, crop, nearest
    resize, component filter).  With info=True also "info" int32 [B, 8]:
    Otsu level, inverted flag, crop x0 y0 x1 y1, resized width and height.
    `backend` names the branch of the reference that is reproduced: "opencv"
    (the one its production run took) or "pil" (the one without OpenCV)."""
    if backend not in SIL_BACKENDS:
        raise ValueError(f"backend must be 'opencv' or 'pil', got {backend!r}")
    _require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 3:
        raise ValueError(f"images must be uint8 [B, H, W], got {images.dtype} {tuple(images.shape)}")
    if images.stride(2) != 1 and images.shape[2] > 1:
        raise ValueError("images must have a unit stride along the row")
    stream = _enter(images.device)
    B, H, W = images.shape
    res = int(resolution)
    out = {"mask": torch.empty((B, res, res), dtype=torch.uint8, device=images.device)}
    if info:
        out["info"] = torch.empty((B, 8), dtype=torch.int32, device=images.device)
    if B == 0:
        return out
    pitch = images.stride(1) if H > 1 else W
    stride = images.stride(0) if B > 1 else pitch * H
    _lib.call("fx_silhouette_batch_u8", images.data_ptr(), B, H, W, pitch, stride, res, float(crop_margin),
              int(bool(preprocess_crop)), int(min_size), SIL_BACKENDS[backend], out["mask"].data_ptr(),
              _ptr(out.get("info")), stream)
    return out


def silhouette_hu_batch(images: torch.Tensor, resolution: int = 224, crop_margin: float = 0.05,
                        preprocess_crop: bool = True, min_size: int = -1, *, mask: bool = False, info: bool = False,
                        backend: str = "opencv", moments: bool = False, hu: bool = False) -> dict:
    """silhouette_batch followed by hu_batch on the silhouettes, in one pass: the raw moments are summed from the
    bit mask while it is on chip, so the silhouettes are neither re-read nor (unless mask=True) written.
    {"hulog7"} [B, 7] float32 -- bit for bit hu_batch(silhouette_batch(...)["mask"])["hulog7"] -- and on request
    "moments24", "hu7", "mask", "info"."""
    if backend not in SIL_BACKENDS:
        raise ValueError(f"backend must be 'opencv' or 'pil', got {backend!r}")
    _require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 3:
        raise ValueError(f"images must be uint8 [B, H, W], got {images.dtype} {tuple(images.shape)}")
    if images.stride(2) != 1 and images.shape[2] > 1:
        raise ValueError("images must have a unit stride along the row")
    stream = _enter(images.device)
    B, H, W = images.shape
    res = int(resolution)
    extra = {}
    if mask:
        extra["mask"] = torch.empty((B, res, res), dtype=torch.uint8, device=images.device)
    if info:
        extra["info"] = torch.empty((B, 8), dtype=torch.int32, device=images.device)
    partial = torch.empty((B, 10, 3), dtype=torch.int64, device=images.device)
    if B:
        pitch = images.stride(1) if H > 1 else W
        stride = images.stride(0) if B > 1 else pitch * H
        _lib.call("fx_silhouette_moments_batch_u8", images.data_ptr(), B, H, W, pitch, stride, res, float(crop_margin),
                  int(bool(preprocess_crop)), int(min_size), SIL_BACKENDS[backend], _ptr(extra.get("mask")),
                  _ptr(extra.get("info")), partial.data_ptr(), stream)
    out = hu_from_partial(partial, moments=moments, hu=hu)
    out.update(extra)
    return out


def gray_fit_batch(images: torch.Tensor, info: Optional[torch.Tensor] = None, resolution: int = 224,
                   bg_value: int = 255) -> torch.Tensor:
    """Bicubic aspect-preserving fit of each view's crop rectangle (columns 2..5
    of the silhouette call's info block, or the whole view) into a
    resolution x resolution canvas of bg_value: uint8 [B, resolution, resolution]."""
    _require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 3:
        raise ValueError(f"images must be uint8 [B, H, W], got {images.dtype} {tuple(images.shape)}")
    if images.stride(2) != 1 and images.shape[2] > 1:
        raise ValueError("images must have a unit stride along the row")
    B, H, W = images.shape
    if info is not None:
        _require_cuda(info, "info")
        if info.dtype != torch.int32 or tuple(info.shape) != (B, 8) or not info.is_contiguous():
            raise ValueError("info must be contiguous int32 [B, 8]")
    stream = _enter(images.device)
    res = int(resolution)
    out = torch.empty((B, res, res), dtype=torch.uint8, device=images.device)
    if B == 0:
        return out
    pitch = images.stride(1) if H > 1 else W
    stride = images.stride(0) if B > 1 else pitch * H
    _lib.call("fx_gray_fit_batch_u8", images.data_ptr(), B, H, W, pitch, stride, _ptr(info), res, int(bg_value),
              out.data_ptr(), stream)
    return out


def hu_groupstat(hulog7: torch.Tensor, group_offsets: torch.Tensor) -> torch.Tensor:
    """mean || std (ddof=0, float32) of the rows of each group -> f32 [G, 14]."""
    _require_cuda(hulog7, "hulog7")
    _require_cuda(group_offsets, "group_offsets")
    if hulog7.dtype != torch.float32 or hulog7.dim() != 2 or hulog7.shape[1] != HU_COLS or not hulog7.is_contiguous():
        raise ValueError("hulog7 must be contiguous float32 [rows, 7]")
    if group_offsets.dtype != torch.int64 or group_offsets.dim() != 1 or group_offsets.numel() < 1:
        raise ValueError("group_offsets must be int64 [G+1]")
    stream = _enter(hulog7.device)
    G = group_offsets.numel() - 1
    out = torch.empty((G, 2 * HU_COLS), dtype=torch.float32, device=hulog7.device)
    _lib.call("fx_hu_groupstat", hulog7.data_ptr(), group_offsets.contiguous().data_ptr(), G, out.data_ptr(), stream)
    return out


def pca_batch(points: torch.Tensor, offsets: Optional[torch.Tensor] = None, n_bins: int = 8,
              normalize: bool = False, *, eig: bool = False) -> dict:
    """PCA descriptors of point sets.

    points: float32 or float64 [B, N, D] (equal-size sets) or, with `offsets`
    (int64 [B+1] point offsets on the device), [total_points, D]; D is 2 or 3.
    float64 points are evaluated as they are (the chunked path for every set:
    slower for small sets; meant for coordinates float32 does not hold exactly).
    Returns {"out": f32 [B, 9+n_bins]} and, with eig=True, "eig" f64 [B,6]
    (lambda1..3, v1).  Sets with fewer than 3 points yield NaN rows.
    """
    _require_cuda(points, "points")
    if points.dtype not in (torch.float32, torch.float64) or not points.is_contiguous():
        raise ValueError("points must be contiguous float32 or float64")
    stream = _enter(points.device)
    if offsets is None:
        if points.dim() != 3:
            raise ValueError("points must be [B, N, D] when offsets is None")
        B, N, D = points.shape
        off_ptr = None
    else:
        _require_cuda(offsets, "offsets")
        if points.dim() != 2 or offsets.dtype != torch.int64 or offsets.dim() != 1:
            raise ValueError("ragged input needs points [total, D] and int64 offsets [B+1]")
        offsets = offsets.contiguous()
        B, N, D = offsets.numel() - 1, int(points.shape[0]), points.shape[1]   # N = points behind the pointer
        off_ptr = offsets.data_ptr()
    res = {"out": torch.empty((B, 9 + int(n_bins)), dtype=torch.float32, device=points.device)}
    if eig:
        res["eig"] = torch.empty((B, 6), dtype=torch.float64, device=points.device)
    if B == 0:
        return res
    _lib.call("fx_pca_batch_f64" if points.dtype == torch.float64 else "fx_pca_batch_f32", points.data_ptr(), off_ptr,
              B, N, D, int(n_bins), int(bool(normalize)), res["out"].data_ptr(), _ptr(res.get("eig")), stream)
    return res


def center_scale(points: torch.Tensor, offsets, *, maxdist: bool = False) -> dict:
    """Centred, unit-sphere-scaled float32 copy of B clouds stored back to back
    ([total, 3] float32 or float64 on the device; `offsets` = B + 1 row
    offsets on the host).  Returns {"points": f32 [total, 3], "centroid": f32
    [B, 3]} and on request "maxdist" f64 [B]."""
    import numpy as np
    _require_cuda(points, "points")
    if points.dim() != 2 or points.shape[1] != 3 or points.dtype not in (torch.float32, torch.float64) \
            or not points.is_contiguous():
        raise ValueError("points must be contiguous float32/float64 [total, 3]")
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    if off.ndim != 1 or off.size < 1 or off[0] != 0 or off[-1] != points.shape[0]:
        raise ValueError("offsets must start at 0 and end at the number of points")
    B = off.size - 1
    stream = _enter(points.device)
    out = {"points": torch.empty((points.shape[0], 3), dtype=torch.float32, device=points.device),
           "centroid": torch.empty((B, 3), dtype=torch.float32, device=points.device)}
    if maxdist:
        out["maxdist"] = torch.empty((B,), dtype=torch.float64, device=points.device)
    if B:
        _lib.call("fx_center_scale", points.data_ptr(), int(points.dtype == torch.float64), off.ctypes.data, B,
                  out["points"].data_ptr(), out["centroid"].data_ptr(), _ptr(out.get("maxdist")), stream)
    return out


def fps(points: torch.Tensor, offsets, n_samples: int, start, *, gather: bool = True) -> dict:
    """Farthest point sampling of B clouds stored back to back (float32
    [total, 3] on the device, `offsets` and the per-cloud first index `start`
    on the host; every cloud needs more than n_samples points).  Returns
    {"idx": int64 [B, n_samples]} and, with gather, "sampled" f32 [B, n_samples, 3]."""
    import numpy as np
    _require_cuda(points, "points")
    if points.dim() != 2 or points.shape[1] != 3 or points.dtype != torch.float32 or not points.is_contiguous():
        raise ValueError("points must be contiguous float32 [total, 3]")
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    st = np.ascontiguousarray(start, dtype=np.int64)
    if off.ndim != 1 or off.size < 1 or off[0] != 0 or off[-1] != points.shape[0]:
        raise ValueError("offsets must start at 0 and end at the number of points")
    B = off.size - 1
    if st.shape != (B,):
        raise ValueError("start must hold one index per cloud")
    stream = _enter(points.device)
    n = int(n_samples)
    out = {"idx": torch.empty((B, n), dtype=torch.int64, device=points.device)}
    if gather:
        out["sampled"] = torch.empty((B, n, 3), dtype=torch.float32, device=points.device)
    if B:
        _lib.call("fx_fps_f32", points.data_ptr(), off.ctypes.data, B, n, st.ctypes.data, out["idx"].data_ptr(),
                  _ptr(out.get("sampled")), stream)
    return out


def column_thresholds(x: torch.Tensor, method: str = "median", quantile: float = 0.5, *, labels: Optional[torch.Tensor] = None,
                      label: int = 0, constant_rule: bool = True) -> dict:
    """Per-column thresholds of a float64 table [rows, cols] on the device
    (reference fit_thresholds' col_thresh, src/models/train_bernoulli_nb.py:348-364).
    `labels` (int32 [rows]) with `label` restricts the rows to one class.
    Returns {"thr": f64 [cols], "flags": int32 [cols] (1 constant, 2 NaN), "count": rows used}."""
    _require_cuda(x, "x")
    if x.dim() != 2 or x.dtype != torch.float64 or x.shape[1] < 1 or (x.shape[0] > 1 and x.stride(1) != 1):
        raise ValueError("x must be float64 [rows, cols] with unit column stride")
    code = {"median": 0, "quantile": 1, "otsu": 2}.get(method)
    if code is None:
        raise ValueError("Unknown binarize method")
    if labels is not None:
        _require_cuda(labels, "labels")
        if labels.dtype != torch.int32 or labels.shape != (x.shape[0],) or not labels.is_contiguous():
            raise ValueError("labels must be contiguous int32 [rows]")
    stream = _enter(x.device)
    rows, cols = int(x.shape[0]), int(x.shape[1])
    thr = torch.empty(cols, dtype=torch.float64, device=x.device)
    flags = torch.empty(cols, dtype=torch.int32, device=x.device)
    count = torch.zeros(1, dtype=torch.int64, device=x.device)
    _lib.call("fx_column_thresholds_f64", x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols, _ptr(labels),
              int(label), code, float(quantile), 1 if constant_rule else 0, thr.data_ptr(), flags.data_ptr(),
              count.data_ptr(), stream)
    return {"thr": thr, "flags": flags, "count": count}


def binarize_ge(x: torch.Tensor, thr: torch.Tensor) -> torch.Tensor:
    """uint8 [rows, cols] = x >= thr per column (reference apply_thresholds,
    src/models/train_bernoulli_nb.py:379-385)."""
    _require_cuda(x, "x")
    _require_cuda(thr, "thr")
    if x.dim() != 2 or x.dtype != torch.float64 or (x.shape[0] > 1 and x.stride(1) != 1):
        raise ValueError("x must be float64 [rows, cols] with unit column stride")
    if thr.dtype != torch.float64 or thr.shape != (x.shape[1],) or not thr.is_contiguous():
        raise ValueError("thr must be contiguous float64 [cols]")
    stream = _enter(x.device)
    rows, cols = int(x.shape[0]), int(x.shape[1])
    out = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    if rows and cols:
        _lib.call("fx_binarize_ge_f64", x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols, thr.data_ptr(),
                  out.data_ptr(), stream)
    return out


def cloud_bounds(points: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    """float64 [B, 9] = min xyz, max xyz, mean xyz of B clouds stored back to
    back (float32 [total, 3], int64 offsets [B + 1], both on the device), as
    the reference's bbox_and_center (src/preprocessing/render_views.py:77-87)."""
    _require_cuda(points, "points")
    _require_cuda(offsets, "offsets")
    if points.dim() != 2 or points.shape[1] != 3 or points.dtype != torch.float32 or not points.is_contiguous():
        raise ValueError("points must be contiguous float32 [total, 3]")
    if offsets.dtype != torch.int64 or offsets.dim() != 1 or offsets.numel() < 1 or not offsets.is_contiguous():
        raise ValueError("offsets must be contiguous int64 [B + 1]")
    stream = _enter(points.device)
    B = offsets.numel() - 1
    out = torch.empty((B, 9), dtype=torch.float64, device=points.device)
    if B:
        _lib.call("fx_cloud_bounds_f32", points.data_ptr(), offsets.data_ptr(), B, out.data_ptr(), stream)
    return out


def camera_table(bounds: torch.Tensor, trig, radius_factor: float, resolution: int):
    """Camera table of B clouds on the device from their bounds (float64
    [B, 9], cloud_bounds) and the per-view cos/sin table (host, [V, 4]):
    (cam float64 [B, V, 9], scale float64 [B]).  Reference:
    compute_camera_poses, src/preprocessing/render_views.py:93-164, and the
    camera basis of render_fallback_matplotlib :306-320."""
    import numpy as np
    _require_cuda(bounds, "bounds")
    if bounds.dtype != torch.float64 or bounds.dim() != 2 or bounds.shape[1] != 9 or not bounds.is_contiguous():
        raise ValueError("bounds must be contiguous float64 [B, 9]")
    tr = np.ascontiguousarray(trig, dtype=np.float64)
    if tr.ndim != 2 or tr.shape[1] != 4:
        raise ValueError("trig must be [V, 4]")
    stream = _enter(bounds.device)
    B, V = int(bounds.shape[0]), int(tr.shape[0])
    cam = torch.empty((B, V, 9), dtype=torch.float64, device=bounds.device)
    scale = torch.empty((B,), dtype=torch.float64, device=bounds.device)
    _lib.call("fx_camera_table_f64", bounds.data_ptr(), B, V, tr.ctypes.data, float(radius_factor), int(resolution),
              cam.data_ptr(), scale.data_ptr(), stream)
    return cam, scale


def render_views(points: torch.Tensor, offsets: torch.Tensor, cam: torch.Tensor, scale: torch.Tensor, resolution: int,
                 stamp, marker_value: int = 0) -> torch.Tensor:
    """uint8 [B, V, res, res] views of B clouds (layout as cloud_bounds) from
    the camera table cam (float64 [B, V, 9]: right, true_up, look-at) and the
    per-cloud scale (float64 [B]); `stamp` is the K x K uint8 coverage stamp of
    the marker (host array).  Reference: render_fallback_matplotlib,
    src/preprocessing/render_views.py:294-366."""
    import numpy as np
    for name, t in (("points", points), ("offsets", offsets), ("cam", cam), ("scale", scale)):
        _require_cuda(t, name)
    if points.dim() != 2 or points.shape[1] != 3 or points.dtype != torch.float32 or not points.is_contiguous():
        raise ValueError("points must be contiguous float32 [total, 3]")
    B = offsets.numel() - 1
    if offsets.dtype != torch.int64 or not offsets.is_contiguous():
        raise ValueError("offsets must be contiguous int64 [B + 1]")
    if cam.dtype != torch.float64 or cam.dim() != 3 or cam.shape[0] != B or cam.shape[2] != 9 or not cam.is_contiguous():
        raise ValueError("cam must be contiguous float64 [B, V, 9]")
    if scale.dtype != torch.float64 or scale.shape != (B,) or not scale.is_contiguous():
        raise ValueError("scale must be contiguous float64 [B]")
    st = np.ascontiguousarray(stamp, dtype=np.uint8)
    if st.ndim != 2 or st.shape[0] != st.shape[1]:
        raise ValueError("stamp must be a square uint8 array")
    V, res = int(cam.shape[1]), int(resolution)
    stream = _enter(points.device)
    out = torch.empty((B, V, res, res), dtype=torch.uint8, device=points.device)
    if B and V:
        _lib.call("fx_render_views_u8", points.data_ptr(), offsets.data_ptr(), B, V, cam.data_ptr(), scale.data_ptr(), res,
                  st.ctypes.data, int(st.shape[0]), int(marker_value), out.data_ptr(), stream)
    return out


def synth_shapes(B: int, H: int, W: int, seed: int = 42, first_index: int = 0, device="cuda",
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """B synthetic {0,255} shape images generated on the device."""
    dev = torch.device(device)
    if out is None:
        out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    _require_cuda(out, "out")
    stream = _enter(out.device)
    if B:
        _lib.call("fx_synth_shapes_u8", out.data_ptr(), B, H, W, out.stride(1), out.stride(0), int(seed),
                  int(first_index), stream)
    return out


def synth_points(B: int, N: int, seed: int = 42, first_index: int = 0, device="cuda") -> torch.Tensor:
    """B synthetic Gaussian clouds [B, N, 3] float32 generated on the device."""
    out = torch.empty((B, N, 3), dtype=torch.float32, device=torch.device(device))
    _require_cuda(out, "out")
    stream = _enter(out.device)
    if B and N:
        _lib.call("fx_synth_points_f32", out.data_ptr(), B, N, int(seed), int(first_index), stream)
    return out


def profile(enable: bool) -> None:
    """Switches HIP-event timing of the library's launches on or off."""
    _lib.call("fx_profile", int(bool(enable)))


def profile_last_ms() -> float:
    """Milliseconds the kernels of the last hu_batch / pca_batch call took
    (waits for them); negative when profiling was off."""
    return float(_lib.load().fx_profile_last_ms())


def png_inflate_batch(staged: dict, H: int, W: int, device) -> dict:
    """Second half of the device decode of PNG files (ingest.stage_png_batch is the first): copies the staged zlib
    streams to the device and inflates, unfilters and converts them there.  Only the files whose stage status is 0
    take part.  {"gray": uint8 [m, H, W] on the device, "rows": int64 [m] positions of those files in the staged
    batch, "status": int32 [m] on the device (0 = decoded; the others are for the host decoders)} -- byte for byte
    what PIL's Image.open(path).convert("L") gives for the decoded ones."""
    device = torch.device(device)
    if device.type != "cuda":
        raise ValueError("png_inflate_batch needs a HIP device")
    stream = _enter(device)
    rows = np.nonzero(staged["status"] == 0)[0].astype(np.int64)
    m = len(rows)
    gray = torch.empty((m, int(H), int(W)), dtype=torch.uint8, device=device)
    status = torch.zeros(m, dtype=torch.int32, device=device)
    if m:
        d_z = staged["staging"][: staged["used"]].to(device, non_blocking=True)
        meta = torch.from_numpy(np.concatenate([staged["offset"][rows], staged["length"][rows].astype(np.int64),
                                                staged["channels"][rows].astype(np.int64)]))
        meta = (meta.pin_memory() if torch.cuda.is_available() else meta).to(device, non_blocking=True)
        d_off = meta[:m].contiguous()
        d_len = meta[m:2 * m].to(torch.int32)
        d_ch = meta[2 * m:].to(torch.int32)
        _lib.call("fx_png_inflate_gray_u8", d_z.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), d_ch.data_ptr(), m,
                  int(H), int(W), gray.data_ptr(), status.data_ptr(), stream)
    if staged.get("block") is not None:
        from . import ingest
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(device))
        ingest.release_staging(staged["block"], done)   # (the block may be staged into again once the copy is out)
        staged["block"] = None
    return {"gray": gray, "rows": rows, "status": status}


def profile_ms_back(back: int) -> float:
    """profile_last_ms of the `back`-th timed call before the most recent one (the library keeps the last 32):
    read after the following calls have been enqueued, it does not hold the host up."""
    return float(_lib.load().fx_profile_ms_back(int(back)))


def profile_last_units() -> int:
    """Units the launch timed by profile_last_ms processed, when the call split its units over several launches of
    that kernel; 0 when the launch processed all of the call's units."""
    return int(_lib.load().fx_profile_last_units())
