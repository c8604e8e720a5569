"""fx_hu_batch_u8 at the ends of its range against the exact CPU oracle: the largest sides (16384), very flat and very
tall images, the tile size and its neighbours, hundreds of thousands of tiny images in one call, gray and 0/1 weights.
usage (GPU box, repository root): python3 tools/huextreme.py"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
eng = importlib.import_module("image-processing_amd.engine")
fxlib = importlib.import_module("image-processing_amd._lib")
from oracle import cpu_ref  # noqa: E402

dev = torch.device("cuda", 0)
c = cpu_ref.load_c_oracle()
rng = np.random.default_rng(1)


def check(imgs, thresh, on, tag):
    got = eng.hu_batch(torch.from_numpy(imgs).to(dev), thresh=thresh, on_value=on, moments=True, hu=True)
    torch.cuda.synchronize()
    ref = c.hu_batch(imgs, thresh=thresh, on_value=on, exact=True)
    m = got["moments24"].cpu().numpy()
    assert np.array_equal(m, ref["moments24"]), tag + f": moments differ in rows {np.nonzero((m != ref['moments24']).any(axis=1))[0][:5].tolist()}"
    assert np.array_equal(got["hu7"].cpu().numpy(), ref["hu7"]), tag + ": invariants"
    print(tag + ": ok", flush=True)


for (H, W) in [(16384, 16), (16, 16384), (16384, 1), (1, 16384), (16384, 1000), (1000, 16384), (9000, 9000), (1024, 1024),
               (1025, 1024), (1024, 1025), (257, 1025), (256, 2048), (255, 2047), (16384, 16384)]:
    for thresh, on in ((-1, 255), (127, 255), (200, 1)):
        if H * W > 100_000_000 and thresh != -1:
            continue
        full = rng.integers(0, 2) == 0
        imgs = np.full((1, H, W), 255, np.uint8) if full else rng.integers(0, 256, (1, H, W), dtype=np.uint8)
        check(imgs, thresh, on, f"{H}x{W} thresh={thresh} on={on} {'all 255' if full else 'noise'}")
for (B, H, W) in [(300000, 1, 1), (200000, 3, 5), (100000, 16, 16), (70000, 17, 31), (65537, 8, 8)]:
    imgs = rng.integers(0, 256, (B, H, W), dtype=np.uint8)
    check(imgs, -1, 255, f"B={B} {H}x{W} gray")
    check(imgs, 127, 255, f"B={B} {H}x{W} binary")
# rows far apart: offsets within an image beyond 32 bits (views of a 5 GB buffer)
big = torch.zeros((1, 300, 1 << 24), dtype=torch.uint8, device=dev)
for (H, W, x0) in [(300, 200, 0), (300, 208, 16), (224, 224, 5), (300, 1500, 3)]:
    imgs = rng.integers(0, 256, (1, H, W), dtype=np.uint8)
    view = big[:, :H, x0:x0 + W]
    view.copy_(torch.from_numpy(imgs).to(dev))
    for thresh, nb in ((-1, None), (127, 8)):
        if nb and W > 1024:
            nb = None
        got = eng.hu_batch(view, thresh=thresh, moments=True, contour_bins=nb)
        torch.cuda.synchronize()
        ref = c.hu_batch(imgs, thresh=thresh, exact=True)
        assert np.array_equal(got["moments24"].cpu().numpy(), ref["moments24"]), f"pitch 2^24, {H}x{W}+{x0} thresh={thresh}"
        if nb:
            want = cpu_ref.contour_pca_features(imgs[0], 127, nb)
            assert np.allclose(got["cpca"].cpu().numpy()[0, :9], want[:9], rtol=1e-6, atol=1e-30), f"pitch 2^24 contour {H}x{W}"
    print(f"pitch 2^24 {H}x{W}+{x0}: ok", flush=True)
# the silhouette stage and the gray fit on such views: the results of the same views stored contiguously
for (H, W, x0) in [(224, 224, 0), (300, 130, 7)]:
    yy, xx = np.mgrid[0:H, 0:W]
    scene = np.clip(235.0 - 195.0 * (((yy - H / 2) / (H / 3)) ** 2 + ((xx - W / 2) / (W / 4)) ** 2 < 1) + rng.normal(0, 8, (H, W)), 0, 255).astype(np.uint8)[None]
    view = big[:, :H, x0:x0 + W]
    view.copy_(torch.from_numpy(scene).to(dev))
    flat = torch.from_numpy(scene).to(dev)
    for backend in ("opencv", "pil"):
        a = eng.silhouette_hu_batch(view, 224, 0.05, True, mask=True, info=True, backend=backend)
        b = eng.silhouette_hu_batch(flat, 224, 0.05, True, mask=True, info=True, backend=backend)
        torch.cuda.synchronize()
        assert all(torch.equal(a[k], b[k]) for k in ("mask", "info", "hulog7")), f"silhouette {backend} pitch 2^24 {H}x{W}"
        assert torch.equal(eng.gray_fit_batch(view, a["info"], 224, 255), eng.gray_fit_batch(flat, b["info"], 224, 255)), f"gray fit pitch 2^24 {H}x{W}"
    print(f"silhouette / gray fit pitch 2^24 {H}x{W}+{x0}: ok", flush=True)
del big
for (H, W) in [(16385, 8), (8, 16385), (0, 8)]:
    try:
        eng.hu_batch(torch.zeros((1, H, W), dtype=torch.uint8, device=dev))
        raise SystemExit(f"{H}x{W} should have been refused")
    except (fxlib.FxError, ValueError) as e:
        print(f"{H}x{W}: refused ({str(e)[:70]})", flush=True)
print("huextreme: ok")
