"""CPU restatement of the reference's Hu / PCA feature arithmetic (NumPy).

TEST INFRASTRUCTURE ONLY: this module is the checker.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` import
it; the product package never does.

* ``compute_pca_features`` follows reference
  ``src/features/extract_3d_pca_features.py:59-145`` step for step with the same
  NumPy/LAPACK calls (``np.cov`` ddof=1, ``np.linalg.eigh``, ``np.linspace``,
  ``np.histogram``), so on the same input it returns the same float32 vector as
  the reference; ``tests/golden/pca_golden.npz`` (made by importing the
  reference, see ``tests/golden/make_golden.py``) pins that.
* ``moments_u8`` / ``hu_from_moments`` / ``hu_log`` restate
  ``extract_2d_hu.py:443-478``; the arithmetic inside ``cv2.moments`` /
  ``cv2.HuMoments`` is OpenCV's (absent from this image, unpinned version) and
  is restated from its published algorithm.  PARITY PIN for Hu: committed
  silhouette panels + feature table only; per-view last digits against a given
  OpenCV build are "parity unpinned".
* ``hu_groupstat`` restates ``extract_2d_hu.py:641-644`` (float32 mean/std).
* ``contour_points`` defines the build's 2-D contour point set (no reference
  counterpart); the descriptor of it is ``compute_pca_features`` on (x, y, 0).
* ``libfx_oracle.so`` (``oracle/fx_oracle.c``) holds the same Hu path in C for
  speed; ``load_c_oracle`` binds it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

MOMENT_KEYS = (
    "m00", "m10", "m01", "m20", "m11", "m02", "m30", "m21", "m12", "m03",
    "mu20", "mu11", "mu02", "mu30", "mu21", "mu12", "mu03",
    "nu20", "nu11", "nu02", "nu30", "nu21", "nu12", "nu03",
)


# --------------------------------------------------------------------------
# Hu path
# --------------------------------------------------------------------------
def binarise(img: np.ndarray, thresh: int, on_value: int = 255) -> np.ndarray:
    """``(img > thresh) * on_value`` as uint8; ``thresh < 0`` returns img
    unchanged (reference extract_2d_hu.py:163,258)."""
    img = np.asarray(img, dtype=np.uint8)
    if thresh < 0:
        return img
    return ((img > thresh).astype(np.uint8) * np.uint8(on_value)).astype(np.uint8)


def raw_moments_exact(img: np.ndarray) -> list[int]:
    """The ten raw moments m_pq = sum x^p y^q I(x, y) as exact Python ints
    (x = column, y = row, origin at the top-left pixel)."""
    a = np.asarray(img, dtype=np.uint8).astype(object)
    h, w = a.shape
    x = np.arange(w, dtype=object)
    y = np.arange(h, dtype=object)
    r0 = a.sum(axis=1)
    r1 = (a * x).sum(axis=1)
    r2 = (a * x * x).sum(axis=1)
    r3 = (a * x * x * x).sum(axis=1)
    return [
        int(r0.sum()), int(r1.sum()), int((y * r0).sum()),
        int(r2.sum()), int((y * r1).sum()), int((y * y * r0).sum()),
        int(r3.sum()), int((y * r2).sum()), int((y * y * r1).sum()),
        int((y * y * y * r0).sum()),
    ]


def complete_moments(raw10) -> np.ndarray:
    """OpenCV completeMomentState in IEEE double, operation order kept."""
    m = np.zeros(24, dtype=np.float64)
    m[:10] = [float(v) for v in raw10]
    m00, m10, m01, m20, m11, m02, m30, m21, m12, m03 = (np.float64(v) for v in m[:10])
    cx = cy = inv = np.float64(0.0)
    if abs(m00) > np.finfo(np.float64).eps:
        inv = np.float64(1.0) / m00
        cx = m10 * inv
        cy = m01 * inv
    mu20 = m20 - m10 * cx
    mu11 = m11 - m10 * cy
    mu02 = m02 - m01 * cy
    m[10], m[11], m[12] = mu20, mu11, mu02
    m[13] = m30 - cx * (3 * mu20 + cx * m10)
    mu11 = mu11 + mu11
    m[14] = m21 - cx * (mu11 + cx * m01) - cy * mu20
    m[15] = m12 - cy * (mu11 + cy * m10) - cx * mu02
    m[16] = m03 - cy * (3 * mu02 + cy * m01)
    s = np.sqrt(abs(inv))
    s2 = inv * inv
    s3 = s2 * s
    m[17:20] = m[10:13] * s2
    m[20:24] = m[13:17] * s3
    return m


def moments_u8(img: np.ndarray, thresh: int = -1, on_value: int = 255) -> np.ndarray:
    """cv2.moments of a uint8 image as a 24-vector ordered like MOMENT_KEYS."""
    return complete_moments(raw_moments_exact(binarise(img, thresh, on_value)))


def hu_from_moments(m: np.ndarray) -> np.ndarray:
    """cv2.HuMoments, operation order kept."""
    nu20, nu11, nu02, nu30, nu21, nu12, nu03 = (np.float64(v) for v in m[17:24])
    t0 = nu30 + nu12
    t1 = nu21 + nu03
    q0, q1 = t0 * t0, t1 * t1
    n4 = 4 * nu11
    s = nu20 + nu02
    d = nu20 - nu02
    hu = np.zeros(7, dtype=np.float64)
    hu[0] = s
    hu[1] = d * d + n4 * nu11
    hu[3] = q0 + q1
    hu[5] = d * (q0 - q1) + n4 * t0 * t1
    t0 = t0 * (q0 - 3 * q1)
    t1 = t1 * (3 * q0 - q1)
    q0 = nu30 - 3 * nu12
    q1 = 3 * nu21 - nu03
    hu[2] = q0 * q0 + q1 * q1
    hu[4] = q0 * t0 + q1 * t1
    hu[6] = q1 * t0 - q0 * t1
    return hu


def hu_log(hu: np.ndarray) -> np.ndarray:
    """-sign(h) * log10(|h| + 1e-300) in float64; inf/nan -> 0
    (extract_2d_hu.py:471-478, before the float32 cast)."""
    out = np.zeros(7, dtype=np.float64)
    for i, h in enumerate(np.asarray(hu, dtype=np.float64)):
        v = -np.sign(h) * np.log10(abs(float(h)) + 1e-300)
        if np.isinf(v) or np.isnan(v):
            v = 0.0
        out[i] = v + 0.0
    return out


def compute_hu_from_image(img: np.ndarray, thresh: int = -1, on_value: int = 255) -> np.ndarray:
    """float32[7] log-Hu vector of a uint8 image (extract_2d_hu.py:443-478)."""
    return hu_log(hu_from_moments(moments_u8(img, thresh, on_value))).astype(np.float32)


def hu_textbook(img: np.ndarray) -> np.ndarray:
    """Hu invariants from the definition: central moments summed directly as
    sum (x-xc)^p (y-yc)^q I in float64, eta = mu / m00^(1+(p+q)/2), then the
    seven classical polynomials.  Independent of OpenCV's evaluation order."""
    a = np.asarray(img, dtype=np.float64)
    h, w = a.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    m00 = a.sum()
    if m00 == 0:
        return np.zeros(7)
    xc, yc = (xx * a).sum() / m00, (yy * a).sum() / m00
    dx, dy = xx - xc, yy - yc

    def eta(p, q):
        return ((dx ** p) * (dy ** q) * a).sum() / m00 ** (1 + (p + q) / 2.0)

    n20, n11, n02 = eta(2, 0), eta(1, 1), eta(0, 2)
    n30, n21, n12, n03 = eta(3, 0), eta(2, 1), eta(1, 2), eta(0, 3)
    a0, a1 = n30 + n12, n21 + n03
    b0, b1 = n30 - 3 * n12, 3 * n21 - n03
    return np.array([
        n20 + n02,
        (n20 - n02) ** 2 + 4 * n11 ** 2,
        b0 ** 2 + b1 ** 2,
        a0 ** 2 + a1 ** 2,
        b0 * a0 * (a0 ** 2 - 3 * a1 ** 2) + b1 * a1 * (3 * a0 ** 2 - a1 ** 2),
        (n20 - n02) * (a0 ** 2 - a1 ** 2) + 4 * n11 * a0 * a1,
        b1 * a0 * (a0 ** 2 - 3 * a1 ** 2) - b0 * a1 * (3 * a0 ** 2 - a1 ** 2),
    ])


def hu_groupstat(hulog7: np.ndarray, offsets=None) -> np.ndarray:
    """mean || std(ddof=0) per group of view rows in float32
    (extract_2d_hu.py:641-644).  offsets=None treats all rows as one group."""
    rows = np.asarray(hulog7, dtype=np.float32)
    if offsets is None:
        offsets = [0, rows.shape[0]]
    out = np.zeros((len(offsets) - 1, 14), dtype=np.float32)
    for g in range(len(offsets) - 1):
        arr = rows[offsets[g]:offsets[g + 1]]
        with np.errstate(all="ignore"):
            out[g, :7] = np.mean(arr, axis=0).astype(np.float32)
            out[g, 7:] = np.std(arr, axis=0, ddof=0).astype(np.float32)
    return out


# --------------------------------------------------------------------------
# PCA path
# --------------------------------------------------------------------------
def compute_pca_features(points: np.ndarray, n_bins: int = 8, normalize: bool = False,
                         return_eig: bool = False):
    """float32[9 + n_bins] descriptor of an N x 3 float64 point set
    (extract_3d_pca_features.py:59-145).  With return_eig also returns the
    float64 (lambda1..3, v1) used, for eigenvalue / eigenvector tolerances."""
    pts = np.asarray(points)
    if pts.shape[0] < 3:
        raise ValueError("too few points (<3)")
    centred = pts - pts.mean(axis=0)
    if normalize:
        radius = np.linalg.norm(centred, axis=1)
        rmax = float(np.max(radius)) if radius.size > 0 else 0.0
        if rmax > 0:
            centred = centred / rmax
    cov = np.cov(centred, rowvar=False, bias=False)
    cov = (cov + cov.T) / 2.0
    w, v = np.linalg.eigh(cov)
    w, v = w[::-1], v[:, ::-1]
    w = np.where(w < 0.0, np.clip(w, a_min=0.0, a_max=None), w).astype(float)
    l1, l2, l3 = float(w[0]), float(w[1]), float(w[2])
    trace = l1 + l2 + l3
    denom = trace if trace > 1e-12 else 1e-12
    head = [l1, l2, l3, l1 / denom, l2 / denom, l3 / denom,
            l1 / (l2 + 1e-12), l2 / (l3 + 1e-12), trace]
    axis1 = v[:, 0].astype(np.float64)
    proj = np.abs(centred.dot(axis1))
    top = float(np.max(proj)) if proj.size > 0 else 0.0
    if top <= 0:
        hist = np.zeros(n_bins, dtype=float)
        hist[0] = 1.0
    else:
        edges = np.linspace(0.0, top, n_bins + 1, endpoint=True)
        hist, _ = np.histogram(proj, bins=edges)
        hist = hist.astype(float)
        tot = hist.sum()
        hist = hist / tot if tot > 0 else np.zeros_like(hist, dtype=float)
    vec = np.asarray(head + [float(c) for c in hist.tolist()], dtype=np.float32)
    if return_eig:
        return vec, np.array([l1, l2, l3, axis1[0], axis1[1], axis1[2]], dtype=np.float64)
    return vec


def contour_points(img: np.ndarray, thresh: int = 127) -> np.ndarray:
    """Foreground pixels (v > thresh; 127 when thresh < 0) with a 4-neighbour
    that is background or outside the image, as float64 rows (x, y, 0) in
    row-major pixel order."""
    t = 127 if thresh < 0 else thresh
    fg = np.asarray(img, dtype=np.uint8) > t
    pad = np.pad(fg, 1, mode="constant", constant_values=False)
    interior = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    ys, xs = np.nonzero(fg & ~interior)
    out = np.zeros((ys.size, 3), dtype=np.float64)
    out[:, 0] = xs
    out[:, 1] = ys
    return out


def contour_pca_features(img: np.ndarray, thresh: int = 127, n_bins: int = 8,
                         return_eig: bool = False):
    """PCA descriptor of the contour point set; NaN vector when it has < 3
    points (the reference function raises ValueError there)."""
    pts = contour_points(img, thresh)
    if pts.shape[0] < 3:
        vec = np.full(9 + n_bins, np.nan, dtype=np.float32)
        return (vec, np.full(6, np.nan)) if return_eig else vec
    return compute_pca_features(pts, n_bins=n_bins, normalize=False, return_eig=return_eig)


# --------------------------------------------------------------------------
# C oracle binding
# --------------------------------------------------------------------------
# --------------------------------------------------------------------------
# Silhouette preprocessing: the reference's branch without OpenCV
# (src/features/extract_2d_hu.py:101-440), restated with the same PIL and
# scipy.ndimage calls.  Pinned by tests/golden/preprocess_golden.npz, which
# holds the reference's own outputs.
# --------------------------------------------------------------------------
def otsu_threshold(img: np.ndarray) -> int:
    """Otsu level in [0, 255]; ties resolved towards the higher level
    (reference :101-126 uses `>=`)."""
    hist, _ = np.histogram(img.ravel(), bins=256, range=(0, 256))
    total = img.size
    weighted_total = np.dot(np.arange(256), hist)
    w_back = 0.0
    sum_back = 0.0
    best, level = 0.0, 0
    for i in range(256):
        w_back += hist[i]
        if w_back == 0:
            continue
        w_fore = total - w_back
        if w_fore == 0:
            break
        sum_back += i * hist[i]
        m_back = sum_back / w_back
        m_fore = (weighted_total - sum_back) / w_fore
        between = w_back * w_fore * (m_back - m_fore) * (m_back - m_fore)
        if between >= best:
            level, best = i, between
    return int(level)


def pil_gaussian_blur_r1(img: np.ndarray) -> np.ndarray:
    """What PIL's GaussianBlur(radius=1) computes on an 8-bit image, written
    out: three box passes of radius 0.25 along x, then three along y, each
    (c*ww + (l + r)*fw + 2^23) >> 24 with replicated edges."""
    ww = int(np.float32(1 << 24) / np.float32(1.5))
    fw = ((1 << 24) - ww) // 2

    def passes(a):
        a = a.astype(np.uint32)
        for _ in range(3):
            left = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
            right = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
            a = (a * ww + (left + right) * fw + (1 << 23)) >> 24
        return a

    return passes(passes(img).T).T.astype(np.uint8)


def compute_otsu(img: np.ndarray):
    """(threshold, 0/255 mask): radius-1 Gaussian blur, Otsu, inversion when the
    foreground exceeds half of the image, 3x3 closing then opening (reference
    :129-188)."""
    from PIL import Image, ImageFilter
    from scipy import ndimage as ndi
    blur = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=1)))
    thresh = otsu_threshold(blur)
    mask = (blur > thresh).astype(np.uint8) * 255
    if float(np.count_nonzero(mask) / mask.size) > 0.5:
        mask = (255 - mask).astype(np.uint8)
    b = mask > 127
    b = ndi.binary_closing(b, structure=np.ones((3, 3)))
    b = ndi.binary_opening(b, structure=np.ones((3, 3)))
    return int(thresh), b.astype(np.uint8) * 255


def find_largest_bbox(mask: np.ndarray):
    """x, y, w, h of the non-zero pixels; the whole image when there are none
    (reference :191-221)."""
    h, w = mask.shape[:2]
    coords = np.argwhere(mask > 0)
    if coords.size == 0:
        return 0, 0, w, h
    y0, x0 = coords.min(axis=0)
    y1, x1 = coords.max(axis=0)
    return int(x0), int(y0), int(x1 - x0 + 1), int(y1 - y0 + 1)


def pad_and_resize(img: np.ndarray, target: int, bg_color: str) -> np.ndarray:
    """Aspect-preserving fit into target x target, centred on the background colour (reference :224-240)."""
    from PIL import Image, ImageOps
    fitted = ImageOps.contain(Image.fromarray(img), (target, target)).convert("L")
    canvas = Image.new("L", (target, target), color=255 if bg_color == "white" else 0)
    canvas.paste(fitted, ((target - fitted.width) // 2, (target - fitted.height) // 2))
    return np.asarray(canvas, dtype=np.uint8)


def clean_mask_np(mask_np: np.ndarray, resolution: int, min_size=None) -> np.ndarray:
    """3x3 opening, then components smaller than min_size dropped (largest kept
    when none reaches it); default min_size = max(8, 0.1 % of resolution^2)
    (reference :246-346)."""
    from scipy import ndimage as ndi
    if min_size is None:
        min_size = max(8, int(0.001 * resolution * resolution))
    b = (mask_np > 127).astype(np.uint8)
    b = ndi.binary_opening(b, structure=np.ones((3, 3))).astype(np.uint8)
    labels, count = ndi.label(b)
    if count == 0:
        return b.astype(np.uint8) * 255
    sizes = np.array(ndi.sum(b, labels, range(1, count + 1)))
    keep = sizes >= min_size
    if not keep.any():
        keep[np.argmax(sizes)] = True
    return np.isin(labels, 1 + np.nonzero(keep)[0]).astype(np.uint8) * 255


def mask_to_canvas(mask: np.ndarray, resolution: int, min_size=None):
    """Nearest resize to fit, cleaning, centred paste (reference :395-440).
    Returns (canvas, resized width, resized height)."""
    from PIL import Image
    pil = Image.fromarray(mask).convert("L")
    scale = min(resolution / float(pil.width), resolution / float(pil.height))
    new_w = max(1, int(round(pil.width * scale)))
    new_h = max(1, int(round(pil.height * scale)))
    small = np.asarray(pil.resize((new_w, new_h), resample=Image.NEAREST), dtype=np.uint8)
    small = clean_mask_np((small > 128).astype(np.uint8) * 255, resolution, min_size)
    canvas = Image.new("L", (resolution, resolution), color=0)
    canvas.paste(Image.fromarray(small).convert("L"), ((resolution - new_w) // 2, (resolution - new_h) // 2))
    return np.asarray(canvas, dtype=np.uint8), new_w, new_h


def silhouette(img: np.ndarray, resolution: int, crop_margin: float, preprocess_crop: bool, min_size=None):
    """(cleaned 0/255 silhouette, info) of one gray view; info = [Otsu level,
    inverted, crop x0, y0, x1, y1, resized width, height] (reference :352-440)."""
    from PIL import Image, ImageFilter
    h, w = img.shape[:2]
    thresh, mask = compute_otsu(img)
    blur = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=1)))
    inverted = int(np.count_nonzero(blur > thresh) / blur.size > 0.5)
    x0, y0, x1, y1 = 0, 0, w, h
    if preprocess_crop:
        x, y, ww, hh = find_largest_bbox(mask)
        side = max(ww, hh)
        pad = int(round(crop_margin * side)) if crop_margin <= 1.0 else int(round(crop_margin))
        x0, y0 = max(0, x - pad), max(0, y - pad)
        x1, y1 = min(w, x + ww + pad), min(h, y + hh + pad)
        if x1 <= x0 or y1 <= y0:
            x0, y0, x1, y1 = 0, 0, w, h
    canvas, new_w, new_h = mask_to_canvas(mask[y0:y1, x0:x1], resolution, min_size)
    return canvas, np.array([thresh, inverted, x0, y0, x1, y1, new_w, new_h], dtype=np.int32)


def preprocess_image(img: np.ndarray, resolution: int, crop_margin: float, bg_color: str, preprocess_crop: bool):
    """(preprocessed gray image, cleaned 0/255 silhouette), both
    resolution x resolution uint8 (reference :352-440)."""
    mask, info = silhouette(img, resolution, crop_margin, preprocess_crop)
    x0, y0, x1, y1 = (int(v) for v in info[2:6])
    return pad_and_resize(img[y0:y1, x0:x1], resolution, bg_color), mask


# --------------------------------------------------------------------------
# Silhouette preprocessing: the reference's branch WITH OpenCV
# (src/features/extract_2d_hu.py:137-148 GaussianBlur + THRESH_OTSU, :171-177
# 5x5 MORPH_ELLIPSE close x2 / open x1, :198-221 findContours EXTERNAL ->
# largest contourArea -> boundingRect, :261-264 3x3 ellipse open, :284-297
# connectedComponentsWithStats(connectivity=8)), the branch the production run
# took.  OpenCV is a third-party dependency of the reference, version not
# pinned (requirements.txt lists only tabulate) and not importable here; its
# published algorithms (modules/imgproc: smooth.dispatch.cpp fixed-point
# Gaussian, thresh.cpp getThreshVal_Otsu_8u, morph getStructuringElement /
# morphOp, contours.cpp border following, shapeapprox contourArea,
# connectedcomponents.cpp) are restated below.  PINNED by the reference's own
# artefacts: on all 21 (raw view_01 -> gray crop, cleaned mask) panel pairs of
# reports/sample_images/*_hu_preview.png (tests/golden/hu_panels.npz) this
# restatement reproduces both panels with zero differing pixels.  The panels
# discriminate the blur kernel / rounding / border, the erosion border, both
# structuring elements and the largest-contour box; they do not discriminate
# Otsu tie handling, 8- vs 4-connectivity of the final filter, contour area vs
# pixel count, or the order among equal candidates: those follow OpenCV's
# algorithm as published.
# --------------------------------------------------------------------------
CV_ELLIPSE5 = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], dtype=bool)
CV_ELLIPSE3 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)   # what getStructuringElement gives for 3 x 3


def cv_gaussian_blur5(img: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(img, (5, 5), 0) on an 8-bit image: sigma <= 0 with
    ksize 5 selects the table kernel [1, 4, 6, 4, 1] / 16, applied in 8.8
    fixed point along x (exact), then along y with one rounding,
    (sum + 128) >> 8; BORDER_REFLECT_101."""
    a = img.astype(np.int64)
    h, w = a.shape
    p = np.pad(a, 2, mode="reflect")
    k = (1, 4, 6, 4, 1)
    rows = sum(k[j] * p[:, j:j + w] for j in range(5))
    tot = sum(k[j] * rows[j:j + h, :] for j in range(5))
    return ((tot + 128) >> 8).astype(np.uint8)


def cv_otsu_threshold(img: np.ndarray) -> int:
    """getThreshVal_Otsu_8u: running class mean kept recursively in double,
    classes lighter than FLT_EPSILON skipped, first strict maximum."""
    h = np.bincount(img.ravel(), minlength=256).astype(np.float64)
    scale = 1.0 / img.size
    mu = 0.0
    for i in range(256):
        mu += i * h[i]
    mu *= scale
    eps = float(np.finfo(np.float32).eps)
    mu1 = q1 = 0.0
    best, level = 0.0, 0
    for i in range(256):
        p_i = h[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > best:
            best, level = sigma, i
    return int(level)


def cv_erode(b: np.ndarray, se: np.ndarray) -> np.ndarray:
    """cv2.erode with the default border: pixels outside never lower the minimum."""
    from scipy import ndimage as ndi
    return ndi.binary_erosion(b, structure=se, border_value=1)


def cv_dilate(b: np.ndarray, se: np.ndarray) -> np.ndarray:
    """cv2.dilate with the default border: pixels outside never raise the maximum."""
    from scipy import ndimage as ndi
    return ndi.binary_dilation(b, structure=se, border_value=0)


def cv_compute_otsu(img: np.ndarray):
    """(threshold, inverted, boolean mask): reference :137-177 with cv2."""
    blur = cv_gaussian_blur5(img)
    thresh = cv_otsu_threshold(blur)
    b = blur > thresh
    inverted = bool(np.count_nonzero(b) / b.size > 0.5)
    if inverted:
        b = ~b
    b = cv_dilate(cv_dilate(b, CV_ELLIPSE5), CV_ELLIPSE5)      # MORPH_CLOSE, iterations=2: dilate twice,
    b = cv_erode(cv_erode(b, CV_ELLIPSE5), CV_ELLIPSE5)        # then erode twice
    b = cv_dilate(cv_erode(b, CV_ELLIPSE5), CV_ELLIPSE5)       # MORPH_OPEN
    return thresh, inverted, b


def cv_fill_holes(b: np.ndarray) -> np.ndarray:
    """Foreground plus every background pixel that is not 4-connected to the
    image frame.  Its 8-connected components are the components that
    findContours(RETR_EXTERNAL) reports (components nested in a hole of another
    one disappear), each with its outer border unchanged."""
    from scipy import ndimage as ndi
    bg, _ = ndi.label(np.pad(~b, 1, constant_values=True))
    return (bg != bg[0, 0])[1:-1, 1:-1]


def cv_external_contours(b: np.ndarray):
    """[(contourArea, (x, y, w, h), raster rank of the first pixel)] of the
    external contours.  The border follower walks the boundary pixels of a
    component with the background on one side; the shoelace sum of that closed
    walk is a sum of local terms over the corners of the pixel grid: a step
    between the two foreground pixels of a straight boundary corner, a diagonal
    step between the two neighbours of the background pixel at a concave
    corner (saddle corners contribute two opposite steps that cancel)."""
    from scipy import ndimage as ndi
    h, w = b.shape
    filled = cv_fill_holes(b)
    lab, n = ndi.label(filled, structure=np.ones((3, 3)))
    if n == 0:
        return []
    L = np.pad(lab, 1)
    A, B, C, D = L[:-1, :-1], L[:-1, 1:], L[1:, :-1], L[1:, 1:]   # pixels around corner (vy, vx): TL, TR, BL, BR
    fa, fb, fc, fd = A > 0, B > 0, C > 0, D > 0
    vy, vx = np.mgrid[0:h + 1, 0:w + 1]
    ax, ay, bx, by, cx, cy, dx, dy = vx - 1, vy - 1, vx, vy - 1, vx - 1, vy, vx, vy
    steps = [   # (corner pattern, from pixel, to pixel, label carrier): background kept on the same hand throughout
        (fa & fb & ~fc & ~fd, (ax, ay), (bx, by), A), (fc & fd & ~fa & ~fb, (dx, dy), (cx, cy), C),
        (fa & fc & ~fb & ~fd, (cx, cy), (ax, ay), A), (fb & fd & ~fa & ~fc, (bx, by), (dx, dy), B),
        (fa & fb & fc & ~fd, (cx, cy), (bx, by), A), (fa & fb & fd & ~fc, (ax, ay), (dx, dy), A),
        (fa & fc & fd & ~fb, (dx, dy), (ax, ay), A), (fb & fc & fd & ~fa, (bx, by), (cx, cy), B),
    ]
    twice = np.zeros(n + 1, dtype=np.int64)
    for sel, (px, py), (qx, qy), carrier in steps:
        np.add.at(twice, carrier[sel], (px * qy - py * qx)[sel])
    first = ndi.minimum_position(np.arange(h * w).reshape(h, w), lab, index=range(1, n + 1))
    boxes = ndi.find_objects(lab)
    return [(abs(int(twice[k])) / 2.0,
             (boxes[k - 1][1].start, boxes[k - 1][0].start, boxes[k - 1][1].stop - boxes[k - 1][1].start,
              boxes[k - 1][0].stop - boxes[k - 1][0].start), first[k - 1][0] * w + first[k - 1][1])
            for k in range(1, n + 1)]


def cv_trace_outer_areas(b: np.ndarray) -> dict:
    """Slow cross-check of cv_external_contours: follows the outer border of
    every 8-connected component pixel by pixel (Suzuki-Abe border following as
    findContours does) and returns {raster rank of first pixel: shoelace area}."""
    from scipy import ndimage as ndi
    h, w = b.shape
    lab, n = ndi.label(b, structure=np.ones((3, 3)))
    p = np.pad(b, 1)
    ring = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]  # clockwise from the left neighbour
    out = {}
    for k in range(1, n + 1):
        ys, xs = np.nonzero(lab == k)
        j = np.lexsort((xs, ys))[0]
        sx, sy = int(xs[j]) + 1, int(ys[j]) + 1
        key = (sy - 1) * w + (sx - 1)
        start = None
        for d in range(8):
            if p[sy + ring[d][1], sx + ring[d][0]]:
                start = d
                break
        if start is None:
            out[key] = 0.0
            continue
        first_next = (sx + ring[start][0], sy + ring[start][1])
        pts = [(sx, sy)]
        prev, cur = first_next, (sx, sy)
        while True:
            d0 = ring.index((prev[0] - cur[0], prev[1] - cur[1]))
            for step in range(1, 9):
                d = (d0 - step) % 8
                if p[cur[1] + ring[d][1], cur[0] + ring[d][0]]:
                    break
            nxt = (cur[0] + ring[d][0], cur[1] + ring[d][1])
            if nxt == (sx, sy) and cur == first_next:
                break
            pts.append(nxt)
            prev, cur = cur, nxt
        twice = sum(pts[t - 1][0] * pts[t][1] - pts[t - 1][1] * pts[t][0] for t in range(len(pts)))
        out[key] = abs(twice) / 2.0
    return out


def cv_find_largest_bbox(b: np.ndarray):
    """x, y, w, h of the external contour with the largest contourArea
    (reference :198-221).  cv2.findContours lists the contours in reverse order
    of discovery and Python's max() keeps the first maximum, so among equal
    areas the component found last in raster order wins."""
    h, w = b.shape
    best = None
    for area, box, rank in cv_external_contours(b):
        if best is None or area > best[0] or (area == best[0] and rank > best[2]):
            best = (area, box, rank)
    return (0, 0, w, h) if best is None else tuple(int(v) for v in best[1])


def cv_clean_mask_np(mask_np: np.ndarray, resolution: int, min_size=None) -> np.ndarray:
    """Reference :246-297 with cv2: 3x3 ellipse (= cross) opening, then
    8-connected components below min_size dropped; when none reaches it the
    largest is kept, the lowest label among equals.  OpenCV's labelling scans
    2x2 blocks, so labels are ordered by the block-raster position of a
    component's first block."""
    from scipy import ndimage as ndi
    if min_size is None:
        min_size = max(8, int(0.001 * resolution * resolution))
    b = mask_np > 127
    b = cv_dilate(cv_erode(b, CV_ELLIPSE3), CV_ELLIPSE3)
    lab, n = ndi.label(b, structure=np.ones((3, 3)))
    if n == 0:
        return b.astype(np.uint8) * 255
    sizes = np.asarray(ndi.sum(b, lab, range(1, n + 1))).astype(np.int64)
    keep = sizes >= min_size
    if not keep.any():
        ys, xs = np.nonzero(b)
        block_rank = (ys >> 1) * (b.shape[1] + 1) + (xs >> 1)
        first_block = np.asarray(ndi.minimum(block_rank, lab[ys, xs], range(1, n + 1)))
        order = np.lexsort((first_block, -sizes))
        keep[order[0]] = True
    return np.isin(lab, 1 + np.nonzero(keep)[0]).astype(np.uint8) * 255


def silhouette_cv(img: np.ndarray, resolution: int, crop_margin: float, preprocess_crop: bool, min_size=None):
    """silhouette() on the reference's OpenCV branch: (cleaned 0/255
    silhouette, info) with info = [Otsu level, inverted, crop x0, y0, x1, y1,
    resized width, height] (reference :352-440)."""
    from PIL import Image
    h, w = img.shape[:2]
    thresh, inverted, b = cv_compute_otsu(img)
    mask = b.astype(np.uint8) * 255
    x0, y0, x1, y1 = 0, 0, w, h
    if preprocess_crop:
        x, y, ww, hh = cv_find_largest_bbox(b)
        side = max(ww, hh)
        pad = int(round(crop_margin * side)) if crop_margin <= 1.0 else int(round(crop_margin))
        x0, y0 = max(0, x - pad), max(0, y - pad)
        x1, y1 = min(w, x + ww + pad), min(h, y + hh + pad)
        if x1 <= x0 or y1 <= y0:
            x0, y0, x1, y1 = 0, 0, w, h
    pil = Image.fromarray(mask[y0:y1, x0:x1]).convert("L")
    scale = min(resolution / float(pil.width), resolution / float(pil.height))
    new_w = max(1, int(round(pil.width * scale)))
    new_h = max(1, int(round(pil.height * scale)))
    small = np.asarray(pil.resize((new_w, new_h), resample=Image.NEAREST), dtype=np.uint8)
    small = cv_clean_mask_np((small > 128).astype(np.uint8) * 255, resolution, min_size)
    canvas = Image.new("L", (resolution, resolution), color=0)
    canvas.paste(Image.fromarray(small).convert("L"), ((resolution - new_w) // 2, (resolution - new_h) // 2))
    return np.asarray(canvas, dtype=np.uint8), np.array([thresh, int(inverted), x0, y0, x1, y1, new_w, new_h], dtype=np.int32)


def preprocess_image_cv(img: np.ndarray, resolution: int, crop_margin: float, bg_color: str, preprocess_crop: bool):
    """preprocess_image() on the reference's OpenCV branch."""
    mask, info = silhouette_cv(img, resolution, crop_margin, preprocess_crop)
    x0, y0, x1, y1 = (int(v) for v in info[2:6])
    return pad_and_resize(img[y0:y1, x0:x1], resolution, bg_color), mask


# --------------------------------------------------------------------------
# Point-cloud normalisation and farthest point sampling
# (reference src/preprocessing/prepare_all.py:199-217, :250-293), restated.
# Pinned by tests/golden/fps_golden.npz, which holds the reference's outputs.
# --------------------------------------------------------------------------
def center_and_scale(points: np.ndarray):
    """(normalized float32 [M, 3], centroid float32 [3])."""
    if points.size == 0:
        return points.astype(np.float32), np.zeros(3, dtype=np.float32)
    centroid = np.mean(points, axis=0)
    centered = points - centroid
    dists = np.linalg.norm(centered, axis=1)
    max_dist = float(np.max(dists)) if dists.size > 0 else 0.0
    scaled = centered if max_dist <= 0 else centered / max_dist
    return scaled.astype(np.float32), centroid.astype(np.float32)


def farthest_point_sampling(points: np.ndarray, n_samples: int, rng=None) -> np.ndarray:
    """int64 [n_samples] indices; rng (a RandomState) draws the start index, or
    shuffles the tiled indices when there are fewer points than samples."""
    import math
    M = points.shape[0]
    if M == 0 or n_samples <= 0:
        return np.array([], dtype=np.int64)
    if n_samples >= M:
        base = np.tile(np.arange(M, dtype=np.int64), int(math.ceil(n_samples / M)))[:n_samples]
        if rng is not None:
            rng.shuffle(base)
        return base.astype(np.int64)
    pts = points.astype(np.float64)
    indices = np.zeros(n_samples, dtype=np.int64)
    start = int(rng.randint(0, M)) if rng is not None else 0
    indices[0] = start
    dists = np.linalg.norm(pts - pts[start], axis=1)
    for i in range(1, n_samples):
        far_idx = int(np.argmax(dists))
        indices[i] = far_idx
        dists = np.minimum(dists, np.linalg.norm(pts - pts[far_idx], axis=1))
    return indices


def build_c_oracle(force: bool = False) -> Path:
    so = _HERE / "libfx_oracle.so"
    src = _HERE / "fx_oracle.c"
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s", "libfx_oracle.so"],
                       check=True, stdout=subprocess.DEVNULL)
    return so


# ---- view renderer without Open3D / pyrender (reference src/preprocessing/render_views.py) ----

def bbox_and_center(points: np.ndarray):
    """bbox_and_center :77-87: min, max, mean, diagonal."""
    mins = points.min(axis=0)
    maxs = points.max(axis=0)
    return mins, maxs, points.mean(axis=0), float(np.linalg.norm(maxs - mins))


def compute_camera_poses(n_views: int, bbox_min, bbox_max, centroid, radius_factor: float = 2.0, seed: int = 42,
                         elevation_levels=None):
    """compute_camera_poses :93-164: rings of cameras at the given elevations
    (default -30, 0, 30 degrees), azimuths evenly spaced per ring plus a jitter
    of +-1e-6 degrees from RandomState(seed), radius = radius_factor * bbox
    diagonal, all looking at the centroid with z up."""
    import math
    if elevation_levels is None:
        elevation_levels = [-30.0, 0.0, 30.0]
    n_elev = len(elevation_levels)
    per_level = max(1, n_views // n_elev)
    counts = [per_level] * n_elev
    for i in range(n_views - per_level * n_elev):
        counts[i] += 1
    diag = float(np.linalg.norm(bbox_max - bbox_min))
    radius = diag * radius_factor if diag > 1e-9 else radius_factor
    rng = np.random.RandomState(seed)
    poses = []
    for elev, cnt in zip(elevation_levels, counts):
        for k in range(cnt):
            az = float((360.0 * k) / max(1, cnt)) + float(rng.uniform(-1e-6, 1e-6))
            el_rad, az_rad = math.radians(elev), math.radians(az)
            offset = np.array([radius * math.cos(el_rad) * math.cos(az_rad),
                               radius * math.cos(el_rad) * math.sin(az_rad),
                               radius * math.sin(el_rad)], dtype=np.float64)
            position = offset + centroid
            up = np.array([0.0, 0.0, 1.0])
            fwd = centroid - position
            if np.linalg.norm(fwd) < 1e-9:
                fwd = np.array([0.0, 0.0, 1.0])
            fwd = fwd / np.linalg.norm(fwd)
            if abs(np.dot(fwd, up)) > 0.999:
                up = np.array([0.0, 1.0, 0.0])
            poses.append({"position": position, "lookat": centroid.copy(), "up": up, "elevation": elev, "azimuth": az})
    if len(poses) > n_views:
        poses = poses[:n_views]
    while len(poses) < n_views:
        poses.append(poses[len(poses) % len(poses)])
    return poses


def camera_plane(pose):
    """Camera basis of render_fallback_matplotlib :306-320: right, true_up."""
    eye, center = pose["position"], pose["lookat"]
    up = pose["up"].astype(np.float64)
    forward = center - eye
    if np.linalg.norm(forward) < 1e-12:
        forward = np.array([0.0, 0.0, 1.0])
    forward = forward / np.linalg.norm(forward)
    right = np.cross(forward, up)
    if np.linalg.norm(right) < 1e-9:
        right = np.cross(forward, np.array([0.0, 1.0, 0.0]))
    right = right / (np.linalg.norm(right) + 1e-12)
    true_up = np.cross(right, forward)
    true_up = true_up / (np.linalg.norm(true_up) + 1e-12)
    return right, true_up


def agg_blend(p: int, c: int, alpha: int) -> int:
    """One channel of Agg's fixed_blender_rgba_plain on an opaque pixel
    (matplotlib src/agg_workaround.h): value p, source c, alpha 0..255."""
    if alpha == 0:
        return p
    if alpha == 255:
        return c
    r = p * 255
    a = ((alpha + 255) << 8) - alpha * 255
    return (((c << 8) - r) * alpha + (r << 8)) // a


def render_view(points: np.ndarray, pose, resolution: int, stamp, bg_color: str = "white") -> np.ndarray:
    """render_fallback_matplotlib :294-366 as matplotlib's Agg backend executes
    it, uint8 [res, res] (R = G = B): projection and scale :322-335; axis limits
    min-1 .. max+1 with equal aspect on the square, decoration-free figure
    (:337-355; the data box is scaled by min(res/W, res/H) and centred); every
    marker snapped to the pixel (floor(px+0.5), floor(res-py+0.5)); the marker's
    coverage stamp blended in point order onto the white figure (the axes patch
    is off, so the background stays white for either bg_color)."""
    res = int(resolution)
    right, true_up = camera_plane(pose)
    rel = points - pose["lookat"]
    x, y = rel.dot(right), rel.dot(true_up)
    diag = float(np.linalg.norm(points.max(axis=0) - points.min(axis=0)))
    if diag < 1e-9:
        diag = 1.0
    scale = (res * 0.35) / diag
    xs, ys = x * scale, y * scale
    x0, x1 = np.min(xs) - 1, np.max(xs) + 1
    y0, y1 = np.min(ys) - 1, np.max(ys) + 1
    W, H = x1 - x0, y1 - y0
    s = min(res / W, res / H)
    offx, offy = (res - W * s) / 2.0, (res - H * s) / 2.0
    px = offx + (xs - x0) * s
    py = offy + (ys - y0) * s
    cols = np.floor(px + 0.5).astype(np.int64)
    rows = np.floor((res - py) + 0.5).astype(np.int64)
    stamp = np.asarray(stamp, dtype=np.int64)
    K = stamp.shape[0]
    half = K // 2
    marker = 0 if bg_color == "white" else 255
    img = np.full((res, res), 255, dtype=np.int64)
    cells = [(j - half, i - half, int(stamp[j, i])) for j in range(K) for i in range(K) if stamp[j, i]]
    for r0, c0 in zip(rows.tolist(), cols.tolist()):
        for dj, di, a in cells:
            r, c = r0 + dj, c0 + di
            if 0 <= r < res and 0 <= c < res:
                img[r, c] = agg_blend(int(img[r, c]), marker, a)
    return img.astype(np.uint8)


# ---- feature binarisation thresholds (reference src/models/train_bernoulli_nb.py:307-385) ----

def otsu_threshold_1d(values: np.ndarray) -> float:
    """_otsu_threshold_1d :307-328, the NumPy branch (scikit-image is absent
    where the reference ran here): 256-bin histogram, cumulative class weight
    and mean, first maximum of the between-class variance, centre of that bin."""
    if values.size == 0:
        return 0.0
    vals = values.astype(float)
    hist, edges = np.histogram(vals, bins=256)
    hist = hist.astype(float)
    p = hist / hist.sum() if hist.sum() > 0 else hist
    omega = np.cumsum(p)
    mu = np.cumsum(p * ((edges[:-1] + edges[1:]) / 2.0))
    mu_t = mu[-1] if len(mu) else 0.0
    sigma = (mu_t * omega - mu) ** 2 / (omega * (1 - omega) + 1e-12)
    idx = int(np.nanargmax(sigma)) if sigma.size else 0
    return float((edges[idx] + edges[idx + 1]) / 2.0 if idx + 1 < len(edges) else edges[idx])


def column_threshold(values: np.ndarray, method: str, quantile: float) -> float:
    """col_thresh :348-356."""
    if method == "median":
        return float(np.median(values))
    if method == "quantile":
        return float(np.quantile(values, quantile))
    if method == "otsu":
        return otsu_threshold_1d(values)
    raise ValueError("Unknown binarize method")


def fit_thresholds(table: np.ndarray, y, method: str = "median", quantile: float = 0.5,
                   strategy: str = "global") -> np.ndarray:
    """fit_thresholds :331-376 on a float64 table [rows, cols] (the feature
    columns of X_train in order) and the labels y: float64 [cols]."""
    assert strategy in {"global", "per_class"}
    table = np.asarray(table, dtype=float)
    y = np.asarray(y)
    out = np.empty(table.shape[1], dtype=np.float64)
    for c in range(table.shape[1]):
        values = table[:, c]
        if np.all(values == values[0]):
            out[c] = float(values[0])
            continue
        if strategy == "global":
            out[c] = column_threshold(values, method, quantile)
            continue
        per_class = []
        for cls in sorted(set(y.tolist())):
            cls_vals = values[y == cls]
            if cls_vals.size == 0:
                continue
            per_class.append(column_threshold(cls_vals, method, quantile))
        out[c] = float(np.mean(per_class)) if per_class else column_threshold(values, method, quantile)
    return out


def apply_thresholds(table: np.ndarray, thr: np.ndarray) -> np.ndarray:
    """apply_thresholds :379-385."""
    return (np.asarray(table, dtype=float) >= np.asarray(thr, dtype=float)).astype(np.uint8)


class COracle:
    """ctypes view of libfx_oracle.so."""

    def __init__(self, path: os.PathLike | None = None):
        self.lib = ctypes.CDLL(str(path or build_c_oracle()))
        c_u8p = ctypes.POINTER(ctypes.c_uint8)
        c_dp = ctypes.POINTER(ctypes.c_double)
        c_fp = ctypes.POINTER(ctypes.c_float)
        c_i64p = ctypes.POINTER(ctypes.c_int64)
        self.lib.fxo_hu_batch_u8.argtypes = [
            c_u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp, c_dp, c_dp, c_fp]
        self.lib.fxo_hu_batch_u8.restype = None
        self.lib.fxo_hu_groupstat.argtypes = [c_fp, c_i64p, ctypes.c_int64, c_fp]
        self.lib.fxo_hu_groupstat.restype = None
        self.lib.fxo_contour_points_u8.argtypes = [
            c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int32), ctypes.c_int64]
        self.lib.fxo_contour_points_u8.restype = ctypes.c_int64

    def hu_batch(self, imgs: np.ndarray, thresh: int = -1, on_value: int = 255, exact: bool = False):
        """imgs: uint8 [B, H, W] C-contiguous.  Returns dict of moments24 [B,24] f64,
        hu7 [B,7] f64, hulog7_f64 [B,7] f64, hulog7 [B,7] f32."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        if imgs.ndim == 2:
            imgs = imgs[None]
        b, h, w = imgs.shape
        m = np.zeros((b, 24), dtype=np.float64)
        hu = np.zeros((b, 7), dtype=np.float64)
        lg = np.zeros((b, 7), dtype=np.float64)
        lf = np.zeros((b, 7), dtype=np.float32)
        as_p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
        self.lib.fxo_hu_batch_u8(as_p(imgs, ctypes.c_uint8), b, h, w, w, h * w, int(thresh),
                                 int(on_value), int(bool(exact)), as_p(m, ctypes.c_double),
                                 as_p(hu, ctypes.c_double), as_p(lg, ctypes.c_double),
                                 as_p(lf, ctypes.c_float))
        return {"moments24": m, "hu7": hu, "hulog7_f64": lg, "hulog7": lf}

    def hu_groupstat(self, hulog7: np.ndarray, offsets) -> np.ndarray:
        rows = np.ascontiguousarray(hulog7, dtype=np.float32)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        out = np.zeros((off.size - 1, 14), dtype=np.float32)
        self.lib.fxo_hu_groupstat(rows.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                  off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                  off.size - 1,
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        return out

    def contour_points(self, img: np.ndarray, thresh: int = 127) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = h * w
        xy = np.zeros((cap, 2), dtype=np.int32)
        n = self.lib.fxo_contour_points_u8(img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                           h, w, w, int(thresh),
                                           xy.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), cap)
        out = np.zeros((n, 3), dtype=np.float64)
        out[:, :2] = xy[:n]
        return out


_C = None


def load_c_oracle() -> COracle:
    global _C
    if _C is None:
        _C = COracle()
    return _C
