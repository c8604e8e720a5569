"""Silhouette preprocessing on the device, both branches of the reference
(with and without OpenCV), against the reference's own outputs (golden
fixtures: the committed production panels for the OpenCV branch, outputs of
the imported reference for the other) and against the oracle: bit-exact."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, pkg

pytestmark = pytest.mark.gpu


BACKEND = "pil"   # set per test by the `backend` fixture


@pytest.fixture(params=["pil", "opencv"])
def backend(request):
    global BACKEND
    BACKEND = request.param
    yield request.param
    BACKEND = "pil"


def oracle_silhouette(oracle, *a):
    return (oracle.silhouette_cv if BACKEND == "opencv" else oracle.silhouette)(*a)


def run(gpu, imgs, res=224, margin=0.05, crop=True, min_size=-1):
    eng = pkg("engine")
    out = eng.silhouette_batch(torch.from_numpy(np.ascontiguousarray(imgs)).to(gpu), res, margin, crop, min_size,
                               info=True, backend=BACKEND)
    torch.cuda.synchronize()
    return out["mask"].cpu().numpy(), out["info"].cpu().numpy()


def check(gpu, oracle, imgs, res=224, margin=0.05, crop=True, min_size=-1):
    masks, info = run(gpu, imgs, res, margin, crop, min_size)
    assert masks.shape == (len(imgs), res, res) and masks.dtype == np.uint8
    for k, img in enumerate(imgs):
        want, winfo = oracle_silhouette(oracle, img, res, margin, crop, None if min_size < 0 else min_size)
        np.testing.assert_array_equal(info[k], winfo, err_msg=f"info of image {k}")
        bad = np.argwhere(masks[k] != want)
        assert bad.size == 0, f"image {k}: {len(bad)} pixels differ, first at {bad[0]}"


def gray_scene(rng, h, w, dark_object=True, noise=12.0, blobs=4):
    """A rendered-view look-alike: smooth blobs on a flat background + noise."""
    yy, xx = np.mgrid[0:h, 0:w]
    f = np.zeros((h, w))
    for _ in range(blobs):
        cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w
        ry, rx = rng.uniform(0.05, 0.3) * h + 1, rng.uniform(0.05, 0.3) * w + 1
        f = np.maximum(f, np.clip(1.5 - ((yy - cy) / ry) ** 2 - ((xx - cx) / rx) ** 2, 0, 1))
    bg, fg = (235.0, 40.0) if dark_object else (20.0, 210.0)
    img = bg + (fg - bg) * f + rng.normal(0, noise, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_golden_masks_of_the_reference(gpu):
    g = np.load(GOLDEN / "preprocess_golden.npz")
    imgs = g["inputs"]
    for crop, tag in ((False, "nocrop"), (True, "crop")):
        masks, info = run(gpu, imgs, 224, 0.05, crop)
        np.testing.assert_array_equal(info[:, 0], g["otsu_thresh"])
        np.testing.assert_array_equal(np.packbits(masks > 127, axis=2), g[f"mask_{tag}"])
        assert set(np.unique(masks)) <= {0, 255}


def test_production_panels_of_the_reference(gpu, hu_panels):
    """OpenCV branch against the reference's own artefacts: for the 21 ids whose
    preview collage is committed, the device's silhouette of raw view_01 equals
    the mask panel and its gray crop equals the crop panel, pixel for pixel."""
    global BACKEND
    BACKEND = "opencv"
    try:
        raw = np.ascontiguousarray(hu_panels["raw_views"][:, 0])
        masks, info = run(gpu, raw, 224, 0.05, True)
        gray = fit(gpu, raw, info, 224, 255)
    finally:
        BACKEND = "pil"
    for i, sid in enumerate(hu_panels["ids"]):
        assert int((masks[i] != hu_panels["masks"][i]).sum()) == 0, f"mask of {sid}"
        assert int((gray[i] != hu_panels["preproc"][i]).sum()) == 0, f"gray crop of {sid}"
    assert np.all(info[:, 1] == 1)   # dark trees on white: the class above the level is the background


def test_view_panels_against_oracle(gpu, oracle, hu_panels, backend):
    raw = hu_panels["raw_views"].reshape(-1, 224, 224)
    check(gpu, oracle, raw, 224, 0.05, True)
    check(gpu, oracle, raw[:12], 224, 0.05, False)
    check(gpu, oracle, raw[:12], 128, 0.2, True)
    check(gpu, oracle, raw[:12], 256, 7.0, True)       # margin in pixels, upscaling


@pytest.mark.parametrize("h,w,res", [(224, 224, 224), (97, 131, 100), (131, 97, 75), (512, 512, 512), (300, 480, 224),
                                     (64, 64, 224), (33, 500, 96), (16, 16, 16)])
def test_synthetic_scenes(gpu, oracle, h, w, res, backend):
    rng = np.random.default_rng(h * 1000 + w)
    imgs = np.stack([gray_scene(rng, h, w, dark_object=bool(k & 1), noise=4.0 + 6 * k, blobs=1 + k) for k in range(6)])
    check(gpu, oracle, imgs, res, 0.05, True)
    check(gpu, oracle, imgs, res, 0.0, False)


def test_noise_images_many_components(gpu, oracle, backend):
    """Pure noise: thousands of runs (the run table leaves LDS), many components
    below min_size, the keep-largest rule when nothing reaches min_size."""
    rng = np.random.default_rng(77)
    imgs = np.stack([rng.integers(0, 256, (224, 224), dtype=np.uint8) for _ in range(4)]
                    + [(rng.random((224, 224)) < p).astype(np.uint8) * 200 for p in (0.3, 0.45, 0.55, 0.7)])
    check(gpu, oracle, imgs, 224, 0.05, True)
    check(gpu, oracle, imgs, 224, 0.05, False, min_size=1)
    check(gpu, oracle, imgs, 224, 0.05, True, min_size=100000)   # nothing reaches it: largest only
    big = np.stack([(rng.random((512, 512)) < 0.5).astype(np.uint8) * 255 for _ in range(2)])
    check(gpu, oracle, big, 512, 0.05, False)


def test_degenerate_views(gpu, oracle, backend):
    flat = np.full((3, 224, 224), 255, dtype=np.uint8)
    flat[1] = 0
    flat[2, 100:103, 50:53] = 7             # 3x3 speck: survives the openings exactly
    check(gpu, oracle, flat, 224, 0.05, True)
    check(gpu, oracle, flat, 224, 0.05, False)
    tiny = np.random.default_rng(3).integers(0, 256, (5, 1, 1), dtype=np.uint8)
    check(gpu, oracle, tiny, 16, 0.05, True)
    thin = np.random.default_rng(4).integers(0, 256, (4, 3, 200), dtype=np.uint8)
    check(gpu, oracle, thin, 64, 0.05, True)
    # object touching every border, and two equally large components (first in raster order wins)
    frame = np.full((2, 96, 96), 230, dtype=np.uint8)
    frame[0, :, :] = 30
    frame[0, 20:70, 20:70] = 230
    frame[1, 10:14, 10:14] = 20
    frame[1, 60:64, 70:74] = 20
    check(gpu, oracle, frame, 96, 0.05, False, min_size=1000)
    check(gpu, oracle, frame, 96, 0.1, True)
    # nested shapes: a blob inside a ring inside a ring (only the outermost border is an external contour),
    # shapes that touch diagonally, and equal areas
    nest = np.full((3, 128, 128), 240, dtype=np.uint8)
    nest[0, 10:118, 10:118] = 20
    nest[0, 30:98, 30:98] = 240
    nest[0, 50:78, 50:78] = 20
    nest[0, 2:8, 120:126] = 20
    nest[1, 20:60, 20:60] = 20
    nest[1, 60:100, 60:100] = 20
    nest[1, 100:120, 10:30] = 20
    nest[2, 10:40, 10:40] = 20
    nest[2, 80:110, 70:100] = 20
    check(gpu, oracle, nest, 128, 0.05, True)
    check(gpu, oracle, nest, 64, 0.0, True, min_size=5000)
    # a thin ring around a solid square: the ring's pixels are fewer than the square's, but the ring is the only
    # external contour (and the largest one); a second, separate square of the same size next to a ring; a ring
    # whose hole holds another ring
    thin = np.full((3, 160, 160), 240, dtype=np.uint8)
    thin[0, 8:150, 8:150] = 20
    thin[0, 16:142, 16:142] = 240
    thin[0, 50:110, 50:110] = 20
    thin[1, 5:75, 5:75] = 20
    thin[1, 13:67, 13:67] = 240
    thin[1, 90:150, 90:150] = 20
    thin[2, 4:156, 4:156] = 20
    thin[2, 12:148, 12:148] = 240
    thin[2, 30:130, 30:130] = 20
    thin[2, 40:120, 40:120] = 240
    check(gpu, oracle, thin, 160, 0.05, True)
    check(gpu, oracle, thin, 96, 0.0, True, min_size=1)


def test_padded_rows_and_empty_batch(gpu, oracle, backend):
    eng = pkg("engine")
    rng = np.random.default_rng(9)
    imgs = np.stack([gray_scene(rng, 120, 150) for _ in range(3)])
    wide = torch.zeros((3, 120, 176), dtype=torch.uint8, device=gpu)
    wide[:, :, :150] = torch.from_numpy(imgs).to(gpu)
    out = eng.silhouette_batch(wide[:, :, :150], 112, 0.05, True, backend=backend)["mask"].cpu().numpy()
    for k in range(3):
        np.testing.assert_array_equal(out[k], oracle_silhouette(oracle, imgs[k], 112, 0.05, True)[0])
    empty = eng.silhouette_batch(torch.zeros((0, 64, 64), dtype=torch.uint8, device=gpu), 32)
    assert tuple(empty["mask"].shape) == (0, 32, 32)


def test_rejects_unsupported_sizes(gpu):
    eng, lib = pkg("engine"), pkg("_lib")
    with pytest.raises(lib.FxError, match="supported up to 512x512"):
        eng.silhouette_batch(torch.zeros((1, 513, 64), dtype=torch.uint8, device=gpu), 224)
    with pytest.raises(lib.FxError, match="resolution"):
        eng.silhouette_batch(torch.zeros((1, 64, 64), dtype=torch.uint8, device=gpu), 1024)
    with pytest.raises(ValueError):
        eng.silhouette_batch(torch.zeros((1, 64, 64), dtype=torch.float32, device=gpu), 64)


def test_large_batch_is_deterministic_and_chunked(gpu, oracle, backend):
    """More views than two internal chunks of blurred scratch hold at 512x512 (a batch this large runs as two
    halves, each in chunks of its own)."""
    rng = np.random.default_rng(21)
    base = np.stack([gray_scene(rng, 512, 512, dark_object=bool(k & 1)) for k in range(8)])
    reps = 2100                                           # 16800 views x 256 KiB > 2 x 2 GiB
    d = torch.from_numpy(base).to(gpu).repeat(reps, 1, 1)
    eng = pkg("engine")
    out = eng.silhouette_batch(d, 224, 0.05, True, backend=backend)["mask"]
    torch.cuda.synchronize()
    first = out[:8].cpu().numpy()
    for k in range(8):
        np.testing.assert_array_equal(first[k], oracle_silhouette(oracle, base[k], 224, 0.05, True)[0])
    assert torch.equal(out.view(reps, 8, 224, 224), out[:8].unsqueeze(0).expand(reps, -1, -1, -1))


@pytest.mark.parametrize("shape", [(512, 512, 224, 3072), (200, 152, 160, 6000), (224, 224, 224, 9000)])
def test_later_rounds_of_a_persistent_block(gpu, oracle, backend, shape):
    """A mask block works through many views in turn and keeps what it derives from its thread index across them.
    Every view of a batch of several rounds per block must equal the result its image gets alone, and those the
    oracle's: the case that exposed a register clobbered between a block's first and second view (the compiler
    had placed a copy ahead of the lane-mask restore in front of a call; the build now checks the assembly for it)."""
    H, W, res, B = shape
    rng = np.random.default_rng(77)
    nbase = 12
    base = np.stack([gray_scene(rng, H, W, dark_object=bool(k & 1), blobs=2 + k % 4) for k in range(nbase)])
    base[3] = rng.integers(0, 256, (H, W), dtype=np.uint8)          # noise: thousands of runs, no dominant component
    base[7, H // 4:H // 2, W // 4:W // 2] = 128                      # a flat patch
    want = [oracle_silhouette(oracle, img, res, 0.05, True, None) for img in base]
    eng = pkg("engine")
    order = torch.from_numpy(rng.integers(0, nbase, B)).to(gpu)      # a different image per round of a block
    d = torch.from_numpy(base).to(gpu)[order].contiguous()
    for _ in range(2):
        r = eng.silhouette_batch(d, res, 0.05, True, info=True, backend=backend)
        torch.cuda.synchronize()
        masks, info = r["mask"], r["info"]
        for k in range(nbase):
            sel = (order == k).nonzero().flatten()
            assert len(sel) > 0
            wm = torch.from_numpy(np.array(want[k][0])).to(gpu)
            wi = torch.from_numpy(np.asarray(want[k][1], dtype=np.int32)).to(gpu)
            bad_i = (info[sel] != wi).any(1).nonzero().flatten()
            assert bad_i.numel() == 0, f"image {k}: info differs in views {sel[bad_i][:8].tolist()} (Otsu level, flags, crop)"
            bad_m = (masks[sel] != wm).flatten(1).any(1).nonzero().flatten()
            assert bad_m.numel() == 0, f"image {k}: mask differs in views {sel[bad_m][:8].tolist()}"


def test_two_halves_on_two_streams_equal_small_batches(gpu, backend):
    """A batch large enough to run as two halves (one on the caller's stream, one on a second stream the library
    keeps for it) gives every view the mask and the moments it gets in a small batch: on the default stream, and on
    a stream of the caller's own that is released afterwards.  The odd count leaves the halves unequal."""
    rng = np.random.default_rng(33)
    base = np.stack([gray_scene(rng, 224, 224, dark_object=bool(k & 1)) for k in range(16)])
    eng = pkg("engine")
    small = eng.silhouette_hu_batch(torch.from_numpy(base).to(gpu), 224, 0.05, True, mask=True, info=True,
                                    backend=backend)
    reps, B = 1251, 16 * 1251 - 5
    d = torch.from_numpy(base).to(gpu).repeat(reps, 1, 1)[:B].contiguous()
    torch.cuda.synchronize()
    own = torch.cuda.Stream(device=gpu)
    for stream in (None, own):
        with torch.cuda.stream(stream):
            big = eng.silhouette_hu_batch(d, 224, 0.05, True, mask=True, info=True, backend=backend)
            # enqueued behind the call on the same stream: must see the second half's results too
            tail = big["hulog7"][-11:].clone()
        torch.cuda.synchronize()
        for key in ("mask", "info", "hulog7"):
            want = small[key].repeat(reps, *([1] * (small[key].dim() - 1)))[:B]
            assert torch.equal(big[key], want), key
        assert torch.equal(tail, big["hulog7"][-11:])
    pkg("_lib").call("fx_release_stream", int(own.cuda_stream))
    again = eng.silhouette_hu_batch(d, 224, 0.05, True, backend=backend)["hulog7"]   # the default stream's own still works
    torch.cuda.synchronize()
    assert torch.equal(again, big["hulog7"])


# ------------------------------------------------------------ gray fit (pad_and_resize)
def fit(gpu, imgs, info, res, bg):
    eng = pkg("engine")
    d_info = None if info is None else torch.from_numpy(np.ascontiguousarray(info, dtype=np.int32)).to(gpu)
    out = eng.gray_fit_batch(torch.from_numpy(np.ascontiguousarray(imgs)).to(gpu), d_info, res, bg)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_gray_fit_golden_of_the_reference(gpu):
    g = np.load(GOLDEN / "preprocess_golden.npz")
    imgs = g["inputs"]
    for crop, tag in ((False, "nocrop"), (True, "crop")):
        _, info = run(gpu, imgs, 224, 0.05, crop)
        np.testing.assert_array_equal(fit(gpu, imgs, info, 224, 255), g[f"preproc_{tag}"])


@pytest.mark.parametrize("h,w,res", [(224, 224, 224), (224, 224, 128), (97, 131, 100), (131, 97, 300), (512, 512, 224),
                                     (512, 300, 512), (40, 500, 64), (17, 9, 224), (1, 1, 8), (3, 200, 64)])
def test_gray_fit_equals_pil(gpu, oracle, h, w, res):
    rng = np.random.default_rng(h * 977 + w)
    imgs = np.stack([rng.integers(0, 256, (h, w), dtype=np.uint8), gray_scene(rng, h, w),
                     np.full((h, w), 255, dtype=np.uint8), (np.indices((h, w)).sum(0) % 2 * 255).astype(np.uint8)])
    for bg, name in ((255, "white"), (0, "black")):
        got = fit(gpu, imgs, None, res, bg)
        for k in range(len(imgs)):
            np.testing.assert_array_equal(got[k], oracle.pad_and_resize(imgs[k], res, name), err_msg=f"image {k}")
    # crop rectangles of every shape, including single rows and columns
    info = np.zeros((len(imgs), 8), dtype=np.int32)
    rects = [(0, 0, w, h), (w // 3, h // 4, w, h), (0, 0, max(1, w // 2), max(1, h // 3)), (w - 1, 0, w, h)]
    for k, (x0, y0, x1, y1) in enumerate(rects):
        info[k, 2:6] = (x0, y0, x1, y1)
    got = fit(gpu, imgs, info, res, 255)
    for k, (x0, y0, x1, y1) in enumerate(rects):
        if (y1 - y0) / (x1 - x0) * res < 0.5 or (x1 - x0) / (y1 - y0) * res < 0.5:
            continue   # PIL refuses a zero-sized target; the device clamps it to one pixel
        np.testing.assert_array_equal(got[k], oracle.pad_and_resize(imgs[k][y0:y1, x0:x1], res, "white"),
                                      err_msg=f"rect {k}")


@pytest.mark.parametrize("h,w,res", [(482, 3, 189), (482, 4, 189), (301, 3, 224), (300, 3, 224), (512, 5, 100), (512, 2, 500)])
def test_gray_fit_of_very_tall_crops(gpu, oracle, h, w, res):
    """Pillow's Image.resize reduces an image more than a hundred times as tall as wide along y first and along x
    afterwards, any other along x first; the uint8 rounding between the passes shows which (300 x 3 is the last size
    that goes along x first)."""
    rng = np.random.default_rng(h + w)
    imgs = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    got = fit(gpu, imgs, None, res, 255)
    for k in range(len(imgs)):
        np.testing.assert_array_equal(got[k], oracle.pad_and_resize(imgs[k], res, "white"), err_msg=f"image {k}")


def test_preprocess_image_signature(gpu, oracle, hu_panels):
    hu = pkg("extract_2d_hu")
    img = hu_panels["raw_views"][3, 1]
    for crop in (False, True):
        pre, mask = hu.preprocess_image(img, 224, 0.05, "white", crop, backend="pil")
        wpre, wmask = oracle.preprocess_image(img, 224, 0.05, "white", crop)
        np.testing.assert_array_equal(pre, wpre)
        np.testing.assert_array_equal(mask, wmask)
        pre, mask = hu.preprocess_image(img, 224, 0.05, "white", crop)     # default: the OpenCV branch
        wpre, wmask = oracle.preprocess_image_cv(img, 224, 0.05, "white", crop)
        np.testing.assert_array_equal(pre, wpre)
        np.testing.assert_array_equal(mask, wmask)
    np.testing.assert_array_equal(hu.pad_and_resize(img[10:200, 30:180], 96, "black"),
                                  oracle.pad_and_resize(img[10:200, 30:180], 96, "black"))


@pytest.mark.parametrize("h,w,res,crop", [(224, 224, 224, True), (200, 152, 160, True), (97, 130, 100, False),
                                          (512, 512, 224, True), (64, 64, 512, True), (224, 224, 37, True),
                                          (512, 512, 512, False), (300, 500, 512, True)])
def test_moments_summed_on_chip_equal_hu_of_the_masks(gpu, backend, h, w, res, crop):
    """The fused call (moments from the bit mask's row runs) gives bit for bit what the two calls give, with and
    without the silhouettes written; empty, full and single-pixel silhouettes included."""
    eng = pkg("engine")
    rng = np.random.default_rng(h * 1000 + w + res)
    imgs = np.stack([gray_scene(rng, h, w, dark_object=bool(k % 2), blobs=1 + k % 5) for k in range(14)])
    imgs[0] = 200                      # flat view
    imgs[1] = 0
    imgs[2] = 255
    imgs[2, h // 2, w // 2] = 0        # a single dark pixel
    imgs[3] = 30                       # an object that fills the view up to a thin light frame: the largest sums
    imgs[3, :3, :] = imgs[3, -3:, :] = 230
    imgs[3, :, :3] = imgs[3, :, -3:] = 230
    d = torch.from_numpy(imgs).to(gpu)
    two = eng.silhouette_batch(d, res, 0.05, crop, info=True, backend=backend)
    want = eng.hu_batch(two["mask"], moments=True, hu=True)
    fused = eng.silhouette_hu_batch(d, res, 0.05, crop, mask=True, info=True, backend=backend, moments=True, hu=True)
    lean = eng.silhouette_hu_batch(d, res, 0.05, crop, backend=backend)
    torch.cuda.synchronize()
    assert torch.equal(fused["mask"], two["mask"]) and torch.equal(fused["info"], two["info"])
    for key in ("moments24", "hu7", "hulog7"):
        a, b = fused[key].cpu().numpy(), want[key].cpu().numpy()
        assert np.array_equal(a, b, equal_nan=True), key
    assert np.array_equal(lean["hulog7"].cpu().numpy(), want["hulog7"].cpu().numpy(), equal_nan=True)
    assert set(lean) == {"hulog7"}
    # an empty batch
    none = eng.silhouette_hu_batch(d[:0], res, 0.05, crop, backend=backend)
    assert tuple(none["hulog7"].shape) == (0, 7)
