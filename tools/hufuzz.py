"""Random geometry for fx_hu_batch_u8 against the CPU oracle: image sizes from 1 x 1 to beyond the tile size (widths
next to multiples of 16, narrow and flat images, the band / mask / tiled organisations), padded rows and unaligned
bases, the three weight modes, batches of mixed content (shapes, noise of several densities, empty, full, single
pixels, frames, stripes), with and without the contour descriptor.  Raw moments, central and normalised moments and
the invariants must be bit-equal, log-Hu within a float32 ulp; contour descriptors as in tests/test_gpu_hu.py.
usage (GPU box, repository root): python3 tools/hufuzz.py [cases] [seed]"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
eng = importlib.import_module("image-processing_amd.engine")
synth = importlib.import_module("image-processing_amd.synth")
fxlib = importlib.import_module("image-processing_amd._lib")
from oracle import cpu_ref  # noqa: E402


def draw_side(rng):
    kind = rng.integers(0, 6)
    if kind == 0:
        return int(rng.integers(1, 41))
    if kind == 1:
        return int(rng.integers(41, 301))
    if kind == 2:
        return int(rng.integers(301, 1101))
    if kind == 3:   # next to a multiple of 16
        return max(1, int(16 * rng.integers(1, 66) + rng.integers(-1, 2)))
    if kind == 4:   # the sizes the path is used at
        return int(rng.choice([224, 256, 512, 1024]))
    return int(rng.integers(1025, 2300))


def fill(rng, img, k):
    H, W = img.shape
    kind = (k + int(rng.integers(0, 3))) % 9
    if kind == 0:
        img[:] = synth.shapes_host(1, H, W, seed=int(rng.integers(0, 1 << 30)))[0]
    elif kind == 1:
        img[:] = (rng.random((H, W)) < rng.choice([0.02, 0.5, 0.97])).astype(np.uint8) * 255
    elif kind == 2:
        img[:] = 0
    elif kind == 3:
        img[:] = 255
    elif kind == 4:
        img[:] = 0
        img[rng.integers(0, H), rng.integers(0, W)] = 255
    elif kind == 5:   # a frame along the border
        img[:] = 0
        img[0, :] = img[-1, :] = 255
        img[:, 0] = img[:, -1] = 255
    elif kind == 6:   # stripes: many short contour runs
        img[:] = 0
        p = int(rng.integers(2, 9))
        if rng.integers(0, 2):
            img[:, ::p] = 255
        else:
            img[::p, :] = 255
    elif kind == 7:   # gray ramp with noise
        yy, xx = np.mgrid[0:H, 0:W]
        img[:] = np.clip((xx * 255.0 / max(1, W - 1) + yy * 40.0 / max(1, H - 1)) % 256 + rng.normal(0, 6, (H, W)), 0, 255).astype(np.uint8)
    else:
        img[:] = rng.integers(0, 256, (H, W), dtype=np.uint8)


def contour_fits(H, W):
    if W <= 1024 and H <= 1024:
        return True
    return W <= 2048 and H <= 2048 and (H + 2) * ((W + 31) // 32) * 4 <= 128 * 1024


def run_case(rng, dev, c_oracle, verbose=False):
    H, W = draw_side(rng), draw_side(rng)
    if H * W > 1500 * 1500:
        W = max(1, (1500 * 1500) // H)
    contour = bool(rng.integers(0, 2)) and contour_fits(H, W)
    B = int(rng.integers(1, 8))
    while B > 1 and B * H * W > (1 << 21 if contour else 1 << 23):
        B -= 1
    thresh = int(rng.choice([-1, 127, int(rng.integers(0, 255))]))
    on = int(rng.choice([255, 1, 7]))
    nb = int(rng.choice([1, 4, 8, 13, 64])) if contour else None
    imgs = np.empty((B, H, W), dtype=np.uint8)
    for b in range(B):
        fill(rng, imgs[b], b)
    # device layout: rows padded by 0..40 bytes, images by a few rows, the first pixel at any byte offset
    pad, off, extra = int(rng.choice([0, 0, 1, 16, 40])), int(rng.choice([0, 0, 3, 16])), int(rng.integers(0, 3))
    big = torch.zeros((B, H + extra, W + pad + off), dtype=torch.uint8, device=dev)
    view = big[:, :H, off:off + W]
    view.copy_(torch.from_numpy(imgs).to(dev))
    noalloc = bool(rng.integers(0, 4) == 0)   # the entry point that allocates nothing: same kernels, scratch in a built-in block
    tag = f"B={B} {H}x{W} pad={pad} off={off} extra={extra} thresh={thresh} on={on} bins={nb}" + (" noalloc" if noalloc else "")
    res = eng.hu_batch(view, thresh=thresh, on_value=on, moments=True, hu=True, contour_bins=nb, contour_eig=contour, noalloc=noalloc)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in res.items()}
    ref = c_oracle.hu_batch(imgs, thresh=thresh, on_value=on, exact=max(H, W) > 512)
    assert np.array_equal(got["moments24"], ref["moments24"]), tag + ": moments"
    assert np.array_equal(got["hu7"], ref["hu7"]), tag + ": invariants"
    ulp = np.spacing(np.abs(ref["hulog7"]).astype(np.float32))
    assert np.all(np.abs(got["hulog7"] - ref["hulog7"]) <= ulp), tag + ": log-Hu"
    ties = 0
    if contour:
        t = 127 if thresh < 0 else thresh
        for b in range(B):
            want, eig = cpu_ref.contour_pca_features(imgs[b], t, nb, return_eig=True)
            if np.isnan(want).all():
                assert np.isnan(got["cpca"][b]).all(), tag + f": image {b} should be a NaN row"
                continue
            assert np.all(np.abs(got["ceig"][b, :3] - eig[:3]) <= 1e-9 * max(eig[0], 1e-300)), tag + f": eigenvalues of image {b}"
            assert np.allclose(got["cpca"][b, :9], want[:9], rtol=1e-6, atol=1e-30), tag + f": head of image {b}"
            n = cpu_ref.contour_points(imgs[b], t).shape[0]
            d = np.abs(got["cpca"][b, 9:].astype(np.float64) - want[9:]) * n
            if eig[0] - eig[1] > 1e-6 * eig[0]:   # (equal eigenvalues -- a square frame -- leave the axis open)
                assert d.max() <= 2.5, tag + f": histogram of image {b} differs by more than edge ties ({d.max():.1f} points)"
                ties += int(np.round(d.sum() / 2))
            assert abs(float(np.sum(got["cpca"][b, 9:], dtype=np.float64)) - 1.0) <= 1e-5, tag + f": histogram of image {b} does not sum to 1"
    if verbose:
        print(tag + f": ok ({ties} pixels counted in a neighbouring bin)", flush=True)
    return ties


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    c = cpu_ref.load_c_oracle()
    ties = 0
    for _ in range(cases):
        ties += run_case(rng, dev, c, verbose=True)
    print(f"hufuzz: {cases} cases, seed {seed}: ok ({ties} edge pixels in all)")


if __name__ == "__main__":
    main()
