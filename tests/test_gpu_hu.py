"""Parity of the HIP Hu path (through the C ABI) with the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu

LOG_RTOL = 1e-5  # relative tolerance on log-scaled Hu in float64 before the float32 cast


def run_gpu(gpu, imgs, thresh=-1, on_value=255, **kw):
    import torch
    eng = pkg("engine")
    t = torch.from_numpy(np.ascontiguousarray(imgs)).to(gpu)
    res = eng.hu_batch(t, thresh=thresh, on_value=on_value, moments=True, hu=True, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in res.items()}


def check_against_oracle(got, ref):
    # raw moments are integers: bit-exact; central/normalised moments and the
    # invariants follow the same IEEE operation order: bit-exact too
    np.testing.assert_array_equal(got["moments24"][:, :10], ref["moments24"][:, :10])
    np.testing.assert_array_equal(got["moments24"], ref["moments24"])
    np.testing.assert_array_equal(got["hu7"], ref["hu7"])
    # log10 comes from different math libraries: relative 1e-5 in float64 terms,
    # and at most 1 float32 ulp after the cast
    a, b = got["hulog7"].astype(np.float64), ref["hulog7_f64"]
    assert np.all(np.abs(a - b) <= LOG_RTOL * np.abs(b) + 1e-30)
    ulp = np.spacing(np.abs(ref["hulog7"]).astype(np.float32))
    assert np.all(np.abs(got["hulog7"] - ref["hulog7"]) <= ulp)


@pytest.mark.parametrize("size", [(224, 224), (256, 256), (512, 512), (64, 1024), (300, 200), (17, 33), (1, 1), (5, 2048), (2048, 16)])
def test_shapes_binary_default(gpu, c_oracle, size):
    H, W = size
    B = 6
    synth = pkg("synth")
    if H >= 8 and W >= 8:
        imgs = synth.shapes_host(B, H, W, seed=7)
    else:
        imgs = (np.random.default_rng(1).integers(0, 2, size=(B, H, W)) * 255).astype(np.uint8)
    check_against_oracle(run_gpu(gpu, imgs), c_oracle.hu_batch(imgs))
    check_against_oracle(run_gpu(gpu, imgs, thresh=127), c_oracle.hu_batch(imgs, thresh=127))


@pytest.mark.parametrize("thresh,on", [(-1, 255), (127, 255), (128, 255), (0, 1), (200, 77), (254, 255)])
def test_gray_images_and_thresholds(gpu, c_oracle, thresh, on):
    rng = np.random.default_rng(3)
    imgs = rng.integers(0, 256, size=(5, 96, 144), dtype=np.uint8)
    imgs[0] = 0          # empty image: m00 = 0 -> all outputs 0
    imgs[1] = 255        # full image
    imgs[2, :, :] = 0
    imgs[2, 40, 70] = 255  # single pixel
    check_against_oracle(run_gpu(gpu, imgs, thresh, on), c_oracle.hu_batch(imgs, thresh, on))


@pytest.mark.parametrize("shape,thresh", [((37, 53), 127), ((64, 256), -1), ((224, 224), 127), ((130, 512), 100)])
def test_many_small_images(gpu, c_oracle, shape, thresh):
    """Batches of more than 8192 small images (several rounds of blocks per CU)."""
    import torch
    eng = pkg("engine")
    H, W = shape
    B = 8192 + 3
    if H >= 8 and W >= 8 and thresh >= 0:
        d = eng.synth_shapes(B, H, W, seed=11, device=gpu)
    else:
        g = torch.Generator(device=gpu)
        g.manual_seed(5)
        d = torch.randint(0, 256, (B, H, W), generator=g, device=gpu, dtype=torch.uint8)
    res = eng.hu_batch(d, thresh=thresh, moments=True, hu=True)
    got = {k: res[k].cpu().numpy() for k in ("moments24", "hu7", "hulog7")}
    check_against_oracle(got, c_oracle.hu_batch(d.cpu().numpy(), thresh=thresh))


def test_reference_silhouette_panels(gpu, c_oracle, hu_panels):
    """The 21 cleaned silhouettes the reference's production run computed Hu on."""
    check_against_oracle(run_gpu(gpu, hu_panels["masks"]), c_oracle.hu_batch(hu_panels["masks"]))
    check_against_oracle(run_gpu(gpu, hu_panels["preproc"]), c_oracle.hu_batch(hu_panels["preproc"]))


def test_padded_rows_and_unaligned_base(gpu, c_oracle):
    import torch
    eng = pkg("engine")
    imgs = pkg("synth").shapes_host(4, 100, 100, seed=11)
    ref = c_oracle.hu_batch(imgs)
    big = torch.zeros((4, 104, 131), dtype=torch.uint8, device=gpu)
    view = big[:, 3:103, 5:105]
    view.copy_(torch.from_numpy(imgs).to(gpu))
    res = eng.hu_batch(view, moments=True, hu=True)
    got = {k: v.cpu().numpy() for k, v in res.items()}
    check_against_oracle(got, ref)


def test_compute_hu_from_image_signature(gpu, c_oracle, hu_panels):
    mod = pkg("extract_2d_hu")
    out = mod.compute_hu_from_image(hu_panels["masks"][0])
    assert out.dtype == np.float32 and out.shape == (7,)
    ref = c_oracle.hu_batch(hu_panels["masks"][:1])["hulog7"][0]
    assert np.all(np.abs(out - ref) <= np.spacing(np.abs(ref)))


def test_groupstat(gpu, oracle, c_oracle):
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(5)
    rows = (rng.normal(size=(57, 7)) * np.array([0.1, 0.3, 0.5, 1, 10, 5, 20]) + np.array([3, 6, 9, 10, 15, 10, 0])).astype(np.float32)
    off = np.array([0, 12, 24, 25, 25, 45, 57], dtype=np.int64)
    got = eng.hu_groupstat(torch.from_numpy(rows).to(gpu), torch.from_numpy(off).to(gpu)).cpu().numpy()
    ref_np = oracle.hu_groupstat(rows, off.tolist())
    ref_c = c_oracle.hu_groupstat(rows, off)
    np.testing.assert_array_equal(ref_c, ref_np)  # the two restatements agree bit for bit
    np.testing.assert_array_equal(got, ref_np)    # so does the kernel (NaN rows for the empty group included)


def test_device_generator_matches_host_twin(gpu):
    eng, synth = pkg("engine"), pkg("synth")
    for (B, H, W, first) in [(5, 256, 256, 0), (3, 512, 512, 1000), (2, 224, 224, 7), (2, 100, 260, 3)]:
        dev = eng.synth_shapes(B, H, W, seed=42, first_index=first, device=gpu).cpu().numpy()
        np.testing.assert_array_equal(dev, synth.shapes_host(B, H, W, seed=42, first_index=first))


@pytest.mark.parametrize("size,nb", [((256, 256), 8), ((512, 512), 8), ((224, 224), 5), ((100, 260), 16), ((40, 1000), 8)])
def test_contour_descriptor(gpu, oracle, size, nb):
    H, W = size
    imgs = pkg("synth").shapes_host(6, H, W, seed=21)
    imgs[5] = 0
    imgs[5, 10, 10:12] = 255  # two contour pixels only -> NaN row
    got = run_gpu(gpu, imgs, contour_bins=nb, contour_eig=True)
    ties = 0
    for b in range(imgs.shape[0]):
        ref, eig = oracle.contour_pca_features(imgs[b], 127, nb, return_eig=True)
        if np.isnan(ref).all():
            assert np.isnan(got["cpca"][b]).all()
            continue
        lam, lam_ref = got["ceig"][b, :3], eig[:3]
        assert np.all(np.abs(lam - lam_ref) <= 1e-9 * lam_ref[0])          # eigenvalues, float64
        assert abs(abs(np.dot(got["ceig"][b, 3:], eig[3:])) - 1.0) <= 1e-9  # v1 up to sign
        np.testing.assert_allclose(got["cpca"][b, :9], ref[:9], rtol=1e-6, atol=1e-30)
        n = oracle.contour_points(imgs[b]).shape[0]
        dcount = np.abs(got["cpca"][b, 9:].astype(np.float64) - ref[9:]) * n
        ties += int(np.round(dcount.sum() / 2))
        assert dcount.max() <= 2.5, "histogram differs by more than edge ties"
    assert ties <= 2


@pytest.mark.parametrize("size,nb", [((224, 224), 8), ((256, 256), 13), ((96, 200), 4), ((512, 512), 32)])
def test_contour_histogram_of_symmetric_shapes(gpu, oracle, size, nb):
    """Centred rectangles, ellipses and diamonds: the centroid is a half pixel, the axis lies along the grid,
    rows of contour pixels cross bin edges and pixels sit exactly on them.  The kernel splits such rows at the
    edge in closed form and takes the centroid from its own division sequence; both must agree with NumPy's
    float64 result."""
    H, W = size
    yy, xx = np.mgrid[0:H, 0:W]
    cy, cx = (H - 1) / 2, (W - 1) / 2
    imgs = np.zeros((9, H, W), np.uint8)
    for k in range(9):
        ry, rx = H * (0.15 + 0.03 * k), W * (0.2 + 0.02 * k)
        if k % 3 == 0:
            imgs[k][(np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)] = 255
        elif k % 3 == 1:
            imgs[k][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
        else:
            imgs[k][np.abs(yy - cy) / ry + np.abs(xx - cx) / rx <= 1.0] = 255
    got = run_gpu(gpu, imgs, contour_bins=nb)
    ties = 0
    for b in range(imgs.shape[0]):
        ref = oracle.contour_pca_features(imgs[b], 127, nb)
        n = oracle.contour_points(imgs[b]).shape[0]
        np.testing.assert_allclose(got["cpca"][b, :9], ref[:9], rtol=1e-6, atol=1e-30)
        dcount = np.abs(got["cpca"][b, 9:].astype(np.float64) - ref[9:]) * n
        ties += int(np.round(dcount.sum() / 2))
        assert dcount.max() <= 2.5, "histogram differs by more than edge ties"
    assert ties <= 2


@pytest.mark.parametrize("size,B", [((2048, 2048), 2), ((1100, 1030), 3), ((3000, 4100), 2), ((8192, 8192), 1), ((1025, 16), 2)])
def test_large_images_tiled_path(gpu, c_oracle, size, B):
    """Images beyond 1024 x 1024 are cut into tiles whose integer sums are
    combined in 128-bit arithmetic: the raw moments are the exact integers,
    rounded once to double (OpenCV's own double accumulation deviates from them
    by rounding only, see test_oracle.py)."""
    H, W = size
    eng = pkg("engine")
    dev = eng.synth_shapes(B, H, W, seed=13, device=gpu)
    imgs = dev.cpu().numpy()
    res = eng.hu_batch(dev, thresh=127, moments=True, hu=True)
    got = {k: v.cpu().numpy() for k, v in res.items()}
    ref = c_oracle.hu_batch(imgs, thresh=127, exact=True)
    check_against_oracle(got, ref)
    near = c_oracle.hu_batch(imgs, thresh=127, exact=False)  # OpenCV's tile order
    np.testing.assert_allclose(got["moments24"][:, :10], near["moments24"][:, :10], rtol=1e-12)
    np.testing.assert_allclose(got["hulog7"][:, :4], near["hulog7"][:, :4], rtol=1e-5)
    # gray weights through the same path
    g = eng.hu_batch(dev // 3 + 7, thresh=-1, moments=True, hu=True)
    check_against_oracle({k: v.cpu().numpy() for k, v in g.items()}, c_oracle.hu_batch(imgs // 3 + 7, exact=True))


def test_contour_on_oversized_image_is_rejected(gpu):
    import torch
    eng, lib = pkg("engine"), pkg("_lib")
    big = torch.zeros((1, 4096, 4096), dtype=torch.uint8, device=gpu)
    with pytest.raises(lib.FxError, match="contour descriptor supports images up to"):
        eng.hu_batch(big, thresh=127, contour_bins=8)


def test_ragged_masks_overflow_path(gpu, oracle, c_oracle):
    """Noise images have a contour segment in almost every 16-pixel chunk, far
    beyond the on-chip segment queue: such images are redone by the mask kernel."""
    rng = np.random.default_rng(17)
    imgs = pkg("synth").shapes_host(4, 512, 512, seed=33)
    imgs[1] = (rng.integers(0, 2, size=(512, 512)) * 255).astype(np.uint8)
    imgs[3] = (rng.integers(0, 4, size=(512, 512)) == 0).astype(np.uint8) * 255
    got = run_gpu(gpu, imgs, thresh=127, contour_bins=8, contour_eig=True)
    check_against_oracle(got, c_oracle.hu_batch(imgs, thresh=127))
    for b in range(4):
        ref, eig = oracle.contour_pca_features(imgs[b], 127, 8, return_eig=True)
        assert np.all(np.abs(got["ceig"][b, :3] - eig[:3]) <= 1e-9 * eig[0])
        np.testing.assert_allclose(got["cpca"][b, :9], ref[:9], rtol=1e-6, atol=1e-30)
        n = oracle.contour_points(imgs[b]).shape[0]
        assert (np.abs(got["cpca"][b, 9:].astype(np.float64) - ref[9:]) * n).max() <= 2.5


@pytest.mark.parametrize("size", [(224, 16), (224, 22), (224, 32), (512, 20), (1024, 8), (1000, 47)])
def test_gray_weights_on_narrow_images_with_contours(gpu, c_oracle, size):
    """Gray weights, images of one or two 16-pixel chunks per row, all pixels 255: the largest partial sums the
    sweep forms.  With the contour descriptor switched on the rows of a lane lay 32 or 64 apart and the 32-bit sum
    of (row offset)^3 * weights overflowed; the moments must be the oracle's with and without it."""
    H, W = size
    rng = np.random.default_rng(5)
    imgs = np.concatenate([np.full((2, H, W), 255, np.uint8), rng.integers(0, 256, (2, H, W), dtype=np.uint8)])
    ref = c_oracle.hu_batch(imgs, thresh=-1, exact=True)
    for nb in (None, 8):
        check_against_oracle(run_gpu(gpu, imgs, thresh=-1, contour_bins=nb), ref)


def test_random_geometry_against_oracle(gpu, c_oracle):
    """tools/hufuzz.py's cases (sizes from 1 x 1 to beyond the tile size, widths next to multiples of 16, padded
    rows, unaligned bases, the three weight modes, mixed content, with and without contours), a fixed seed."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import hufuzz
    rng = np.random.default_rng(20261112)
    for _ in range(120):
        hufuzz.run_case(rng, gpu, c_oracle)


def test_full_size_properties(gpu):
    """C2/C3-sized batches: properties that need no CPU pass over the data.
    Raw moments are additive over disjoint masks and the Hu vector is invariant
    under translation of the shape."""
    import torch
    eng = pkg("engine")
    B, H, W = 2048, 512, 512
    imgs = eng.synth_shapes(B, H, W, seed=42, device=gpu)
    res = eng.hu_batch(imgs, moments=True, hu=True)
    left, right = imgs.clone(), imgs.clone()
    left[:, :, W // 2:] = 0
    right[:, :, : W // 2] = 0
    ml = eng.hu_batch(left, moments=True)["moments24"][:, :10]
    mr = eng.hu_batch(right, moments=True)["moments24"][:, :10]
    assert torch.equal(ml + mr, res["moments24"][:, :10])
    # m00 = 255 * foreground pixel count
    assert torch.equal(res["moments24"][:, 0], 255.0 * (imgs > 127).sum(dim=(1, 2)).double())
    # translation by (dx, dy) inside the canvas: shapes live in the central half
    small = eng.synth_shapes(64, 256, 256, seed=5, device=gpu)
    canvas_a = torch.zeros((64, 512, 512), dtype=torch.uint8, device=gpu)
    canvas_b = torch.zeros_like(canvas_a)
    canvas_a[:, 10:266, 20:276] = small
    canvas_b[:, 200:456, 150:406] = small
    ha = eng.hu_batch(canvas_a, hu=True)["hu7"]
    hb = eng.hu_batch(canvas_b, hu=True)["hu7"]
    scale = ha.abs().max(dim=0).values
    assert torch.all((ha - hb).abs() <= 1e-6 * ha.abs() + 1e-7 * scale)


def test_row_bands_equal_whole_image(gpu, oracle):
    """An image cut into row bands: the summed limb arrays finalise to exactly
    the unsplit image's moments (the multi-GPU path for very tall images)."""
    eng, fxd = pkg("engine"), pkg("dist")
    rng = np.random.default_rng(12)
    for (H, W, thresh, on_value, world) in ((4096, 3000, 127, 255, 3), (8192, 8192, -1, 255, 8), (700, 333, 90, 7, 4),
                                            (300, 64, 127, 255, 2)):
        img = torch.from_numpy(rng.integers(0, 256, (2, H, W), dtype=np.uint8)).to(gpu)
        whole = eng.hu_batch(img, thresh=thresh, on_value=on_value, moments=True, hu=True)
        total = torch.zeros((2, 10, 3), dtype=torch.int64, device=gpu)
        for r in range(world):
            y0, y1 = fxd.band_range(H, r, world)
            if y1 > y0:
                total += eng.hu_partial(img[:, y0:y1], y0, thresh, on_value)
        got = eng.hu_from_partial(total, moments=True, hu=True)
        for k in ("moments24", "hu7", "hulog7"):
            assert torch.equal(got[k], whole[k]), f"{k} differs for {H}x{W} in {world} bands"
        # the limbs stand for the exact integers
        if H <= 700:
            raw = oracle.raw_moments_exact(oracle.binarise(img[0].cpu().numpy(), thresh, on_value))
            host = total[0].cpu().numpy()
            assert [fxd.from_limbs(host[k]) for k in range(10)] == [int(v) for v in raw]
    out = fxd.hu_rowbands(img, 0, thresh, on_value, moments=True)       # single process: no group, one band
    assert torch.equal(out["moments24"], whole["moments24"])


def test_two_streams_overlap(gpu, c_oracle):
    """Scratch belongs to (device, stream): calls enqueued on two streams with
    different batch sizes run concurrently and neither disturbs the other."""
    eng = pkg("engine")
    a = eng.synth_shapes(6000, 512, 512, seed=31, device=gpu)
    b = eng.synth_shapes(777, 256, 256, seed=32, device=gpu)
    c = eng.synth_shapes(9, 224, 224, seed=33, device=gpu)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
    res = {}
    for rnd in range(3):   # the buffers grow on the first round and are reused afterwards
        with torch.cuda.stream(s1):
            res["a"] = eng.hu_batch(a, thresh=127, moments=True, hu=True, contour_bins=8)
        with torch.cuda.stream(s2):
            res["b"] = eng.hu_batch(b, thresh=127, moments=True, hu=True, contour_bins=8)
            res["c"] = eng.hu_batch(c, thresh=127, moments=True, hu=True)
        with torch.cuda.stream(s1):
            res["c1"] = eng.hu_batch(c, thresh=127, moments=True, hu=True)   # smaller batch after the big one
    torch.cuda.synchronize()
    single = {"a": eng.hu_batch(a, thresh=127, moments=True, hu=True, contour_bins=8),
              "b": eng.hu_batch(b, thresh=127, moments=True, hu=True, contour_bins=8)}
    torch.cuda.synchronize()
    for k in ("a", "b"):
        for f in ("moments24", "hu7", "hulog7", "cpca"):
            np.testing.assert_array_equal(res[k][f].cpu().numpy(), single[k][f].cpu().numpy(), err_msg=f"{k}.{f}")
    for k, src in (("a", a[-200:]), ("b", b[:200]), ("c", c), ("c1", c)):
        got = {f: res[k][f][-200:].cpu().numpy() if k == "a" else res[k][f][:200].cpu().numpy()
               for f in ("moments24", "hu7", "hulog7")}
        check_against_oracle(got, c_oracle.hu_batch(src.cpu().numpy(), thresh=127))
    pkg("_lib").call("fx_release_stream", int(s1.cuda_stream))
    pkg("_lib").call("fx_release_stream", int(s2.cuda_stream))
    torch.cuda.synchronize()


def test_full_shard_125k_images(gpu, c_oracle):
    """The benchmark's own launch: 125 000 x 512 x 512 (32.8 GB, 125 000
    blocks) with the contour descriptor.  m00 against a torch count over every
    image, additivity of the raw moments over a split of every image, finite
    descriptors, and the oracle on rows from both ends of the grid."""
    eng = pkg("engine")
    B, H, W = 125_000, 512, 512
    imgs = torch.empty((B, H, W), dtype=torch.uint8, device=gpu)
    for a in range(0, B, 8192):
        b = min(B, a + 8192)
        eng.synth_shapes(b - a, H, W, seed=42, first_index=a, device=gpu, out=imgs[a:b])
    res = eng.hu_batch(imgs, thresh=127, moments=True, hu=True, contour_bins=8)
    counts = torch.empty(B, dtype=torch.float64, device=gpu)
    for a in range(0, B, 4096):
        counts[a:a + 4096] = (imgs[a:a + 4096] > 127).sum(dim=(1, 2)).double()
    assert torch.equal(res["moments24"][:, 0], 255.0 * counts)
    assert torch.all(torch.isfinite(res["hulog7"]))
    few = counts < 3
    assert torch.all(torch.isfinite(res["cpca"][~few]))
    assert torch.all((res["cpca"][~few][:, 9:].sum(dim=1) - 1).abs() < 1e-5)
    rows = list(range(4)) + [62_499, 62_500] + list(range(B - 4, B))
    sel = torch.tensor(rows, device=gpu)
    got = {k: res[k][sel].cpu().numpy() for k in ("moments24", "hu7", "hulog7")}
    check_against_oracle(got, c_oracle.hu_batch(imgs[sel].cpu().numpy(), thresh=127))
    # additivity: left half + right half of every image = the image (integers, so exactly)
    whole = res["moments24"][:, :10].clone()
    del res
    half = imgs.clone()
    half[:, :, W // 2:] = 0
    ml = eng.hu_batch(half, thresh=127, moments=True)["moments24"][:, :10].clone()
    half.copy_(imgs)
    half[:, :, : W // 2] = 0
    mr = eng.hu_batch(half, thresh=127, moments=True)["moments24"][:, :10]
    assert torch.equal(ml + mr, whole)
