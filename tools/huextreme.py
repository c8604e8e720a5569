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
for (H, W) in [(16385, 8), (8, 16385), (0, 8)]:
    try:
        eng.hu_batch(torch.zeros((1, H, W), dtype=torch.uint8, device=dev))
        raise SystemExit(f"{H}x{W} should have been refused")
    except (fxlib.FxError, ValueError) as e:
        print(f"{H}x{W}: refused ({str(e)[:70]})", flush=True)
print("huextreme: ok")
