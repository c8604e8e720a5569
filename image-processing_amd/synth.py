"""Host twin of the device shape generator (csrc/fx_synth.hip).

`shapes_host` returns byte-identical images to `engine.synth_shapes` for the
same (seed, index, H, W): the rasteriser uses integer arithmetic only.  It
exists so that CPU-only tests and the CPU baseline can work on the very images
the GPU benchmarks use.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1
_DIR_X = np.array([1024, 946, 724, 392, 0, -392, -724, -946, -1024, -946, -724, -392, 0, 392, 724, 946], dtype=np.int64)
_DIR_Y = np.array([0, 392, 724, 946, 1024, 946, 724, 392, 0, -392, -724, -946, -1024, -946, -724, -392], dtype=np.int64)


def _mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _draw(seed: int, index: int, slot: int) -> int:
    return _mix64((_mix64((seed ^ ((index * 0xD1342543DE82EF95) & _M64)) & _M64) + slot * 0x9E3779B97F4A7C15) & _M64)


def _rng(h: int, lo: int, span: int) -> int:
    return lo + ((h >> 20) % span)


def shape_params(seed: int, index: int, H: int, W: int) -> dict:
    """Ellipses and polygon of image `index` (all integers)."""
    m = min(W, H)
    n_ell = 3 + ((_draw(seed, index, 0) >> 20) % 4)
    ell = []
    for e in range(6):
        base = 16 + 8 * e
        cx = _rng(_draw(seed, index, base + 0), W // 4, max(W // 2, 1))
        cy = _rng(_draw(seed, index, base + 1), H // 4, max(H // 2, 1))
        lo, span = max(m // 25, 1), max(m // 8, 1)
        a = _rng(_draw(seed, index, base + 2), lo, span)
        b = _rng(_draw(seed, index, base + 3), lo, span)
        ux = _rng(_draw(seed, index, base + 4), 1, 16)
        uy = _rng(_draw(seed, index, base + 5), -16, 33)
        ell.append((cx, cy, a * a, b * b, ux, uy, max(a, b)))
    n_vert = 5 + ((_draw(seed, index, 1) >> 20) % 4)
    pcx = _rng(_draw(seed, index, 2), (3 * W) // 10, max((2 * W) // 5, 1))
    pcy = _rng(_draw(seed, index, 3), (3 * H) // 10, max((2 * H) // 5, 1))
    verts = []
    for k in range(n_vert):
        d = (k * 16) // n_vert
        lo, span = max(m // 20, 1), max(m // 6, 1)
        r = _rng(_draw(seed, index, 80 + k), lo, span)
        verts.append((pcx + ((int(_DIR_X[d]) * r) >> 10), pcy + ((int(_DIR_Y[d]) * r) >> 10)))
    return {"ellipses": ell[:n_ell], "polygon": verts}


def shape_image(seed: int, index: int, H: int, W: int) -> np.ndarray:
    """uint8 [H, W] image `index`, values {0, 255}."""
    p = shape_params(seed, index, H, W)
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    fg = np.zeros((H, W), dtype=bool)
    for (cx, cy, a2, b2, ux, uy, rad) in p["ellipses"]:
        dx, dy = x - cx, y - cy
        box = (np.abs(dx) <= rad) & (np.abs(dy) <= rad)
        du = ux * dx + uy * dy
        dv = -uy * dx + ux * dy
        n2 = ux * ux + uy * uy
        # outside the box the products may be large but cannot overflow int64 for sizes <= 16384
        fg |= box & (du * du * b2 + dv * dv * a2 <= a2 * b2 * n2)
    v = p["polygon"]
    xs = [q[0] for q in v]
    ys = [q[1] for q in v]
    box = (x >= min(xs)) & (x <= max(xs)) & (y >= min(ys)) & (y <= max(ys))
    inside = np.zeros((H, W), dtype=bool)
    for k in range(len(v)):
        x1, y1 = v[k]
        x2, y2 = v[(k + 1) % len(v)]
        straddle = (y1 > y) != (y2 > y)
        val = (y - y1) * (x2 - x1) - (x - x1) * (y2 - y1)
        hit = straddle & (val != 0) & ((val > 0) == (y2 > y1))
        inside ^= hit
    fg |= box & inside
    return (fg.astype(np.uint8) * np.uint8(255)).astype(np.uint8)


def shapes_host(B: int, H: int, W: int, seed: int = 42, first_index: int = 0) -> np.ndarray:
    """uint8 [B, H, W]; image b is image `first_index + b` of the data set."""
    return np.stack([shape_image(seed, first_index + b, H, W) for b in range(B)]) if B else np.zeros((0, H, W), np.uint8)


def points_host(B: int, N: int, seed: int = 42, first_index: int = 0) -> np.ndarray:
    """float32 [B, N, 3] anisotropic Gaussian clouds with the same law as the
    device generator (per-axis scale log-uniform in [0.05, 1], random rotation).
    Not byte-identical to the device generator (which uses float32 device
    math); parity tests copy device data back instead."""
    out = np.empty((B, N, 3), dtype=np.float32)
    for b in range(B):
        rng = np.random.default_rng([seed, first_index + b])
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        s = 0.05 * np.exp(rng.uniform(size=3) * np.log(20.0))
        out[b] = ((rng.normal(size=(N, 3)) * s) @ q.T).astype(np.float32)
    return out
