"""Random batch geometry for fx_pca_batch_f32 / fx_pca_batch_f64 against the CPU oracle.  tools/pcasoak.py varies the
sets of one kind of batch; this varies the batch: equal-size batches [B, N, D] of any N and CSR batches whose sets run
from none or two points (NaN rows) over the register and LDS kernels' size classes to clouds of tens of thousands of
points in one call, 2-D and 3-D, float32 and float64, 1 .. 64 bins, both `normalize` settings, identical and collinear
points.  usage (GPU box, repository root): python3 tools/pcafuzz.py [cases] [seed]"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
eng = importlib.import_module("image-processing_amd.engine")
from oracle import cpu_ref  # noqa: E402
from pcasoak import make_set  # noqa: E402


def draw_size(rng):
    kind = rng.integers(0, 8)
    if kind == 0:
        return int(rng.integers(0, 3))            # no descriptor: NaN row
    if kind == 1:
        return int(rng.integers(3, 17))
    if kind == 2:
        return int(rng.integers(17, 257))
    if kind == 3:
        return int(rng.integers(257, 1025))
    if kind == 4:
        return int(rng.integers(1025, 2049))
    if kind == 5:   # next to the limits of the size classes
        return max(3, int(rng.choice([16, 64, 128, 256, 512, 1024, 2048, 8192]) + rng.integers(-1, 2)))
    if kind == 6:
        return int(rng.integers(2049, 60000))
    return int(rng.integers(3, 300))


def one_set(rng, n, dims, wide):
    if n == 0:
        return np.zeros((0, dims), np.float64 if wide else np.float32)
    kind = rng.integers(0, 10) if n >= 3 else 9
    if n < 3:
        p = rng.normal(size=(n, dims))
    elif kind == 0:
        p = np.repeat(rng.normal(size=(1, dims)), n, axis=0)                      # identical points
    elif kind == 1:
        p = np.outer(np.arange(n, dtype=np.float64), rng.normal(size=dims))        # exactly collinear, equally spaced
    else:
        p = make_set(rng, n, dims).astype(np.float64)
    if wide:
        return p + rng.normal(size=p.shape) * 1e-9 if kind > 1 else p              # values float32 does not hold
    return p.astype(np.float32)


def check_set(tag, pts, n_bins, normalize, out, eig):
    n = pts.shape[0]
    if n < 3:
        assert np.isnan(out).all(), tag + ": a set of fewer than three points must give a NaN row"
        return 0
    if np.all(pts == pts[0]):
        # identical points: the reference's row is nine zeros and all mass in the first bin as long as np.mean
        # returns the point itself; for thousands of points its pairwise sums round, and what the reference then
        # computes is rounding noise divided by rounding noise.  The kernels centre on the first point and give
        # the exact row at any size.
        want = np.zeros(9 + n_bins, np.float32)
        want[9] = 1.0
        assert np.array_equal(out, want) and np.all(eig[:3] == 0.0), tag + f": identical points give {out}"
        return 0
    p3 = np.zeros((n, 3))
    p3[:, :pts.shape[1]] = pts.astype(np.float64)
    ro, re = cpu_ref.compute_pca_features(p3, n_bins, normalize, return_eig=True)
    l1 = max(re[0], 1e-300)
    # 1e-9 of the largest eigenvalue, plus what the rounding of the coordinates themselves leaves open when a set is
    # a tight cluster far from the origin (the reference subtracts a rounded mean: spread / distance of its digits
    # are gone; a cluster of 1e-9 around 0.7 carries seven digits)
    c0 = p3 - p3.mean(axis=0)
    spread = float(np.linalg.norm(c0, axis=1).max())
    tol = 1e-9 + 64.0 * 2.2e-16 * float(np.abs(p3).max()) / max(spread, 1e-300)
    assert np.all(np.abs(eig[:3] - re[:3]) <= tol * l1), tag + f": eigenvalues {eig[:3]} against {re[:3]} (tolerance {tol:.1e})"
    if tol > 1e-7:
        return 0   # (ratios and bins of such a set are decided by those lost digits)
    # the nine scalars: against the oracle where the small eigenvalues are more than rounding noise of the large one
    # (their ratios are not defined otherwise), and always against the formulas applied to the kernel's own float64
    # eigenvalues (reference extract_3d_pca_features.py:112-125)
    if re[2] > 1e-6 * l1:
        assert np.allclose(out[:9], ro[:9], rtol=1e-6, atol=1e-30), tag + f": head {out[:9]} against {ro[:9]}"
    lam = np.clip(eig[:3], 0.0, None)
    tr = lam.sum()
    safe = tr if tr > 1e-12 else 1e-12
    own = np.array([lam[0], lam[1], lam[2], lam[0] / safe, lam[1] / safe, lam[2] / safe, lam[0] / (lam[1] + 1e-12),
                    lam[1] / (lam[2] + 1e-12), tr], dtype=np.float64).astype(np.float32)
    assert np.allclose(out[:9], own, rtol=1e-6, atol=1e-30), tag + f": head {out[:9]} against {own} from its own eigenvalues"
    assert abs(float(out[9:].astype(np.float64).sum()) - 1.0) <= 1e-5, tag + ": histogram does not sum to 1"
    gap = (re[0] - re[1]) / l1
    diff = np.abs(out[9:].astype(np.float64) - ro[9:]) * n
    moved = diff.sum() / 2
    if gap > 1e-6 and diff.max() > 2.5:
        c = p3 - p3.mean(axis=0)
        if normalize:
            rm = np.linalg.norm(c, axis=1).max()
            c = c / rm if rm > 0 else c
        pr = np.abs(c @ re[3:])
        q = pr / max(pr.max(), 1e-300) * n_bins
        on_edge = int(np.sum(np.abs(q - np.rint(q)) < 1e-9 * max(1.0, n_bins)))
        assert moved <= on_edge + 0.5, tag + f": {moved:.0f} counts moved, {on_edge} points on bin edges"
    return int(round(moved)) if gap > 1e-6 else 0


def run_case(rng, dev, verbose=False):
    dims = int(rng.choice([2, 3, 3]))
    wide = bool(rng.integers(0, 4) == 0)
    normalize = bool(rng.integers(0, 2))
    n_bins = int(rng.choice([1, 2, 8, 8, 10, 33, 64]))
    uniform = bool(rng.integers(0, 3) == 0)
    dt = np.float64 if wide else np.float32
    if uniform:
        N = max(3, draw_size(rng))
        B = int(rng.integers(1, 40))
        while B > 1 and B * N > 400000:
            B -= 1
        sets = [one_set(rng, N, dims, wide) for _ in range(B)]
        pts = torch.from_numpy(np.stack(sets).astype(dt)).to(dev)
        res = eng.pca_batch(pts, None, n_bins, normalize, eig=True)
        tag = f"uniform B={B} N={N} D={dims} {'f64' if wide else 'f32'} bins={n_bins} normalize={normalize}"
    else:
        B = int(rng.choice([1, 2, 7, 60, 300]))
        sizes, total = [], 0
        for _ in range(B):
            n = draw_size(rng)
            if total + n > 500000:
                n = int(rng.integers(3, 100))
            sizes.append(n)
            total += n
        sets = [one_set(rng, n, dims, wide) for n in sizes]
        off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(sizes, out=off[1:])
        flat = np.concatenate(sets).astype(dt) if total else np.zeros((0, dims), dt)
        slack = int(rng.integers(0, 5))            # points behind the last set that belong to nobody
        flat = np.concatenate([flat, np.full((slack, dims), 7.0, dt)])
        res = eng.pca_batch(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), n_bins, normalize, eig=True)
        tag = (f"csr B={B} sizes {min(sizes)}..{max(sizes)} total={total} D={dims} {'f64' if wide else 'f32'} bins={n_bins} "
               f"normalize={normalize}")
    torch.cuda.synchronize()
    out, eig = res["out"].cpu().numpy(), res["eig"].cpu().numpy()
    moved = 0
    for b in range(len(sets)):
        moved += check_set(tag + f" set {b} (n={sets[b].shape[0]})", sets[b], n_bins, normalize, out[b], eig[b])
    if verbose:
        print(tag + f": ok ({moved} counts in a neighbouring bin)", flush=True)
    return moved


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    moved = 0
    for _ in range(cases):
        moved += run_case(rng, dev, verbose=True)
    print(f"pcafuzz: {cases} cases, seed {seed}: ok ({moved} counts moved in all)")


if __name__ == "__main__":
    main()
