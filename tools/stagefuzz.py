"""Random geometry for the stages in front of the feature path against the CPU oracle: the view renderer (batches of
clouds of 1 .. 6000 points, any resolution from 16 to 700 px, every marker size of the table, 1 .. 12 views: zero
differing pixels), centre/scale (float32 and float64 clouds of 1 .. 300 000 points: bit-equal) and farthest point
sampling (clouds of 2 .. 40 000 points with duplicates, lattices and clusters, more or fewer samples than points,
with and without a random start: the reference's indices).
usage (GPU box, repository root): python3 tools/stagefuzz.py [cases] [seed]"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
rv = importlib.import_module("image-processing_amd.render_views")
prep = importlib.import_module("image-processing_amd.prepare_all")
stamps = importlib.import_module("image-processing_amd.marker_stamps")
from oracle import cpu_ref  # noqa: E402


def cloud(rng, n, dtype=np.float32):
    kind = rng.integers(0, 7)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    p = (rng.normal(size=(n, 3)) * np.exp(rng.uniform(-2, 1, size=3))) @ q.T + rng.normal(size=3) * 3
    if kind == 1:
        p = np.tile(rng.normal(size=(1, 3)), (n, 1))                                      # all points equal
    elif kind == 2:
        p = np.outer(np.linspace(-1, 1, n), rng.normal(size=3))                          # a line
    elif kind == 3:
        p = np.round(p * 2) / 2                                                            # a lattice: ties, duplicates
    elif kind == 4:
        p = p[rng.integers(0, max(1, n // 5), n)]                                         # repeated points
    elif kind == 5:
        c = rng.normal(size=(3, 3)) * 4
        p = c[rng.integers(0, 3, n)] + rng.normal(size=(n, 3)) * 1e-3                    # far-apart clusters
    return np.ascontiguousarray(p.astype(dtype))


def near_a_rounding_tie(p64, pose, res, eps=1e-9):
    """Does a marker of this view sit within eps of the half-pixel boundary where its snapped position changes?  The
    projection is a float64 dot product; NumPy's and the kernel's may differ in the last bit (the order of the three
    products), which decides such a marker's pixel.  An odd resolution puts the view's centre on such a boundary."""
    right, true_up = cpu_ref.camera_plane(pose)
    rel = p64 - pose["lookat"]
    x, y = rel.dot(right), rel.dot(true_up)
    diag = float(np.linalg.norm(p64.max(axis=0) - p64.min(axis=0)))
    scale = (res * 0.35) / (diag if diag >= 1e-9 else 1.0)
    xs, ys = x * scale, y * scale
    x0, x1, y0, y1 = xs.min() - 1, xs.max() + 1, ys.min() - 1, ys.max() + 1
    s = min(res / (x1 - x0), res / (y1 - y0))
    px = (res - (x1 - x0) * s) / 2.0 + (xs - x0) * s
    py = (res - (y1 - y0) * s) / 2.0 + (ys - y0) * s
    fx = np.abs((px + 0.5) - np.rint(px + 0.5))
    fy = np.abs((res - py + 0.5) - np.rint(res - py + 0.5))
    return bool(min(fx.min(), fy.min()) < eps)


def render_case(rng):
    B = int(rng.integers(1, 6))
    ps = float(rng.choice(sorted(stamps.STAMPS.keys())))
    stamp = rv.marker_stamp(ps)
    if stamp.shape[0] > 7:
        ps, stamp = 2.0, rv.marker_stamp(2.0)
    cells = int((np.asarray(stamp) > 0).sum())
    res = int(rng.choice([int(rng.integers(16, 129)), int(rng.integers(129, 400)), int(rng.integers(400, 701)), 224]))
    views = int(rng.integers(1, 13))
    sizes = [int(rng.choice([1, 2, int(rng.integers(3, 200)), int(rng.integers(200, 3000)), int(rng.integers(3000, 6000))]))
             for _ in range(B)]
    clouds = [cloud(rng, n) for n in sizes]
    got = rv.render_views_batch(clouds, views, res, ps).cpu().numpy()
    tag = f"render B={B} sizes={sizes} res={res} size={ps} views={views}"
    assert got.shape == (B, views, res, res), tag
    budget = 400000   # blends the oracle's Python loop is given per case
    for b, c in enumerate(clouds):
        p64 = c.astype(np.float64)
        mins, maxs, cen, _ = cpu_ref.bbox_and_center(p64)
        poses = cpu_ref.compute_camera_poses(views, mins, maxs, cen, 2.0, 42)
        for v in sorted({0, int(rng.integers(0, views))}):
            if budget < 0:
                break
            if near_a_rounding_tie(p64, poses[v], res):
                continue
            budget -= c.shape[0] * cells
            want = cpu_ref.render_view(p64, poses[v], res, stamp)
            diff = int((got[b, v] != want).sum())
            assert diff == 0, tag + f": {diff} pixels differ in view {v} of cloud {b}"
    return tag


def center_case(rng):
    B = int(rng.integers(1, 6))
    wide = bool(rng.integers(0, 2))
    sizes = [int(rng.choice([1, 2, int(rng.integers(3, 100)), int(rng.integers(100, 5000)), int(rng.integers(5000, 300000))]))
             for _ in range(B)]
    clouds = [cloud(rng, n, np.float64 if wide else np.float32) for n in sizes]
    got = prep.center_and_scale_batch(clouds)
    tag = f"center_scale B={B} sizes={sizes} {'f64' if wide else 'f32'}"
    for c, (pts, cen) in zip(clouds, got):
        want_pts, want_cen = cpu_ref.center_and_scale(c.astype(np.float64))
        assert np.array_equal(cen, want_cen), tag + ": centroid"
        assert np.array_equal(pts, want_pts), tag + f": {int((pts != want_pts).sum())} coordinates differ"
    return tag


def fps_case(rng):
    B = int(rng.integers(1, 5))
    n_samples = int(rng.choice([1, 2, 16, 64, 257, 1024]))
    sizes = [int(rng.choice([2, int(rng.integers(3, 200)), int(rng.integers(200, 4097)), int(rng.integers(4097, 40000))]))
             for _ in range(B)]
    clouds = [cloud(rng, n) for n in sizes]
    seeds = [None if rng.integers(0, 3) == 0 else int(rng.integers(0, 1 << 30)) for _ in range(B)]
    got = prep.farthest_point_sampling_batch(clouds, n_samples, [None if s is None else np.random.RandomState(s) for s in seeds])
    tag = f"fps B={B} sizes={sizes} samples={n_samples}"
    for c, s, idx in zip(clouds, seeds, got):
        want = cpu_ref.farthest_point_sampling(c, n_samples, None if s is None else np.random.RandomState(s))
        assert np.array_equal(idx, want), tag + f": first difference at sample {int(np.argmax(idx != want))} of a cloud of {c.shape[0]}"
    return tag


def run_case(rng, verbose=False):
    tag = [render_case, center_case, fps_case][int(rng.integers(0, 3))](rng)
    if verbose:
        print(tag + ": ok", flush=True)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 90
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    for _ in range(cases):
        run_case(rng, verbose=True)
    print(f"stagefuzz: {cases} cases, seed {seed}: ok")


if __name__ == "__main__":
    main()
