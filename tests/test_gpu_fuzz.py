"""Seeded random sweeps over shapes, strides and settings for every device
entry point, each case checked against the oracle to the bar of its path
(bit-exact integer work; the PCA tolerances of test_gpu_pca)."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def blobs(rng, h, w, gray=False):
    yy, xx = np.mgrid[0:h, 0:w]
    f = np.zeros((h, w), dtype=bool)
    for _ in range(int(rng.integers(1, 5))):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        ry, rx = rng.uniform(1, h / 2 + 1), rng.uniform(1, w / 2 + 1)
        f |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    if gray:
        return np.where(f, rng.integers(128, 256, (h, w)), rng.integers(0, 128, (h, w))).astype(np.uint8)
    return f.astype(np.uint8) * 255


def test_hu_random_shapes_strides_thresholds(gpu, c_oracle):
    eng = pkg("engine")
    rng = np.random.default_rng(20251111)
    for case in range(60):
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        if case % 7 == 0:
            w = int(rng.integers(1, 20)) * 16            # the aligned vector path
        b = int(rng.integers(1, 6))
        gray = bool(case % 3 == 0)
        imgs = np.stack([blobs(rng, h, w, gray) for _ in range(b)])
        thresh = int(rng.choice([-1, 0, 100, 127, 200, 254]))
        on_value = int(rng.integers(1, 256))
        pad = int(rng.choice([0, 0, 3, 16]))
        d = torch.zeros((b, h, w + pad), dtype=torch.uint8, device=gpu)
        d[:, :, :w] = torch.from_numpy(imgs).to(gpu)
        res = eng.hu_batch(d[:, :, :w], thresh=thresh, on_value=on_value, moments=True, hu=True)
        ref = c_oracle.hu_batch(imgs, thresh=thresh, on_value=on_value)
        tag = f"case {case}: {b}x{h}x{w} thresh {thresh} on {on_value} pad {pad}"
        np.testing.assert_array_equal(res["moments24"].cpu().numpy(), ref["moments24"], err_msg=tag)
        np.testing.assert_array_equal(res["hu7"].cpu().numpy(), ref["hu7"], err_msg=tag)
        np.testing.assert_allclose(res["hulog7"].cpu().numpy(), ref["hulog7"], rtol=1e-5, atol=0, err_msg=tag)


def test_contour_descriptor_random_shapes(gpu, oracle):
    eng = pkg("engine")
    rng = np.random.default_rng(7)
    for case in range(25):
        h, w = int(rng.integers(8, 400)), int(rng.integers(8, 400))
        imgs = np.stack([blobs(rng, h, w) for _ in range(3)])
        nb = int(rng.choice([4, 8, 16]))
        res = eng.hu_batch(torch.from_numpy(imgs).to(gpu), thresh=127, contour_bins=nb, contour_eig=True)
        got = res["cpca"].cpu().numpy()
        for k in range(3):
            pts = oracle.contour_points(imgs[k], 127)
            if pts.shape[0] < 3:
                assert np.isnan(got[k]).all(), f"case {case}/{k}: fewer than 3 contour pixels must give NaN"
                continue
            want = oracle.contour_pca_features(imgs[k], 127, nb)
            np.testing.assert_allclose(got[k, :9], want[:9], rtol=2e-6, atol=1e-9, err_msg=f"case {case}/{k}")
            assert np.abs(got[k, 9:] - want[9:]).max() * pts.shape[0] <= 2.5, f"case {case}/{k}: histogram"


def test_pca_random_sets(gpu, oracle):
    pca = pkg("extract_3d_pca_features")
    rng = np.random.default_rng(99)
    for case in range(12):
        equal = bool(case % 2)
        n0 = int(rng.choice([3, 4, 64, 255, 256, 1000, 1024, 1500, 2048, 3000]))
        sets = []
        for _ in range(int(rng.integers(1, 20))):
            n = n0 if equal else int(rng.integers(3, 2500))
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            p = (rng.normal(size=(n, 3)) * np.exp(rng.uniform(-3, 1, size=3))) @ q.T + rng.normal(size=3) * 5
            sets.append(p.astype(np.float32))
        nb = int(rng.choice([5, 8, 16]))
        norm = bool(case % 3 == 0)
        got = pca.compute_pca_features_batch(sets, n_bins=nb, normalize=norm)
        for k, p in enumerate(sets):
            want = oracle.compute_pca_features(p.astype(np.float64), nb, norm)
            tag = f"case {case} set {k} n={len(p)} bins={nb} normalize={norm}"
            lam1 = max(float(want[0]), 1e-30)
            np.testing.assert_allclose(got[k, :3], want[:3], rtol=1e-6, atol=1e-9 * lam1, err_msg=tag)
            np.testing.assert_allclose(got[k, 3:6], want[3:6], rtol=1e-6, atol=1e-9, err_msg=tag)
            np.testing.assert_allclose(got[k, 8], want[8], rtol=1e-6, err_msg=tag)
            assert np.abs(got[k, 9:] - want[9:]).max() * len(p) <= 2.5, tag


def test_silhouette_and_gray_fit_random_views(gpu, oracle):
    eng = pkg("engine")
    rng = np.random.default_rng(31)
    for case in range(20):
        h, w = int(rng.integers(4, 400)), int(rng.integers(4, 400))
        res = int(rng.choice([32, 64, 100, 224, 300]))
        margin = float(rng.choice([0.0, 0.05, 0.3, 5.0]))
        crop = bool(case % 2)
        views = []
        for k in range(3):
            base = blobs(rng, h, w).astype(np.float64)
            lo, hi = sorted(rng.uniform(0, 255, 2))
            img = lo + (hi - lo) * base / 255.0 + rng.normal(0, rng.uniform(0, 20), (h, w))
            views.append(np.clip(np.rint(img if k % 2 else 255 - img), 0, 255).astype(np.uint8))
        views = np.stack(views)
        d = torch.from_numpy(views).to(gpu)
        for backend, ref in (("pil", oracle.silhouette), ("opencv", oracle.silhouette_cv)):
            out = eng.silhouette_batch(d, res, margin, crop, info=True, backend=backend)
            masks, info = out["mask"].cpu().numpy(), out["info"].cpu().numpy()
            fit = eng.gray_fit_batch(d, out["info"], res, 255).cpu().numpy()
            for k in range(3):
                tag = f"case {case}/{k} ({backend}): {h}x{w} -> {res}, margin {margin}, crop {crop}"
                want, winfo = ref(views[k], res, margin, crop)
                np.testing.assert_array_equal(info[k], winfo, err_msg=tag)
                np.testing.assert_array_equal(masks[k], want, err_msg=tag)
                x0, y0, x1, y1 = (int(v) for v in winfo[2:6])
                cw, ch = x1 - x0, y1 - y0
                if min(cw / ch, ch / cw) * res >= 0.5:      # PIL refuses a zero-sized target
                    np.testing.assert_array_equal(fit[k], oracle.pad_and_resize(views[k][y0:y1, x0:x1], res, "white"),
                                                  err_msg=tag)


def test_center_scale_and_fps_random_clouds(gpu, oracle):
    prep = pkg("prepare_all")
    rng = np.random.default_rng(57)
    clouds, seeds = [], []
    for case in range(14):
        m = int(rng.choice([5, 33, 1023, 1024, 1025, 2049, 5000, 16384, 16385, 30000]))
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        p = (rng.normal(size=(m, 3)) * np.exp(rng.uniform(-2, 2, size=3))) @ q.T + rng.normal(size=3) * 100
        clouds.append(p if case % 2 else p.astype(np.float32).astype(np.float64))
        seeds.append(None if case % 3 == 0 else int(rng.integers(0, 1 << 30)))
    n = 512
    got = prep.normalize_and_sample(clouds, n, [np.random.RandomState(s) if s is not None else None for s in seeds])
    for k, c in enumerate(clouds):
        norm, cen = oracle.center_and_scale(c)
        idx = oracle.farthest_point_sampling(norm.astype(np.float32), n,
                                             np.random.RandomState(seeds[k]) if seeds[k] is not None else None)
        np.testing.assert_array_equal(got[k][2], idx, err_msg=f"cloud {k} ({len(c)} points)")
        np.testing.assert_array_equal(got[k][0], norm[idx].astype(np.float32))
        np.testing.assert_array_equal(got[k][1], cen)
