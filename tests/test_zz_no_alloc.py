"""The entry point that allocates nothing (fx_hu_batch_u8_noalloc) against
fx_hu_batch_u8: same kernels, scratch in a block built into the library."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu

KEYS = ("hulog7", "moments24", "hu7", "cpca", "ceig")


def _alloc(B, nb, dev):
    return {"hulog7": torch.full((B, 7), -7.0, dtype=torch.float32, device=dev),
            "moments24": torch.full((B, 24), -7.0, dtype=torch.float64, device=dev),
            "hu7": torch.full((B, 7), -7.0, dtype=torch.float64, device=dev),
            "cpca": torch.full((B, 9 + nb), -7.0, dtype=torch.float32, device=dev),
            "ceig": torch.full((B, 6), -7.0, dtype=torch.float64, device=dev)}


def _same(a, b):
    for k in KEYS:
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        np.testing.assert_array_equal(x, y, err_msg=k)   # NaN rows compare equal here


def test_noalloc_equals_allocating_entry_point_twice(gpu, c_oracle):
    """Two calls in a row, the second on a smaller batch: the block keeps the
    first call's contents and nothing of them may leak into the second."""
    eng = pkg("engine")
    rng = np.random.default_rng(9)
    big = eng.synth_shapes(3000, 512, 512, seed=3, device=gpu)
    # a noise image: its contour queue overflows, so the redo list and its counter are used too
    big[17] = torch.from_numpy((rng.integers(0, 2, size=(512, 512)) * 255).astype(np.uint8)).to(gpu)
    big[2999] = 0
    small = eng.synth_shapes(37, 512, 512, seed=8, first_index=100, device=gpu)
    small[5] = torch.from_numpy((rng.integers(0, 4, size=(512, 512)) == 0).astype(np.uint8) * 255).to(gpu)
    for imgs in (big, small):
        B = imgs.shape[0]
        want = eng.hu_batch(imgs, thresh=127, moments=True, hu=True, contour_bins=8, contour_eig=True)
        out = _alloc(B, 8, gpu)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(gpu)[0]   # device-wide: sees hipMalloc calls made outside torch too
        got = eng.hu_batch(imgs, thresh=127, moments=True, hu=True, contour_bins=8, contour_eig=True, out=out,
                           noalloc=True)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(gpu)[0] >= free_before
        _same(got, want)
        ref = c_oracle.hu_batch(imgs[:64].cpu().numpy(), thresh=127)
        np.testing.assert_array_equal(got["moments24"][:64].cpu().numpy(), ref["moments24"])
        np.testing.assert_array_equal(got["hu7"][:64].cpu().numpy(), ref["hu7"])


def test_noalloc_tiled_and_gray_paths(gpu):
    eng = pkg("engine")
    for (B, H, W, thresh) in ((3, 3000, 2100, 127), (2, 2048, 2048, -1), (5, 100, 260, 90)):
        imgs = eng.synth_shapes(B, H, W, seed=5, device=gpu)
        if thresh < 0:
            imgs = imgs // 3 + 7
        want = eng.hu_batch(imgs, thresh=thresh, moments=True, hu=True)
        out = {k: v for k, v in _alloc(B, 8, gpu).items() if k in ("hulog7", "moments24", "hu7")}
        got = eng.hu_batch(imgs, thresh=thresh, moments=True, hu=True, out=out, noalloc=True)
        for k in ("hulog7", "moments24", "hu7"):
            assert torch.equal(got[k], want[k]), (k, H, W)


def test_noalloc_rejects_batches_beyond_the_block(gpu):
    eng, lib = pkg("engine"), pkg("_lib")
    imgs = torch.zeros((65537, 16, 16), dtype=torch.uint8, device=gpu)
    with pytest.raises(lib.FxError, match="built-in block holds"):
        eng.hu_batch(imgs, noalloc=True)
    eng.hu_batch(imgs[:65536], noalloc=True)  # exactly the limit is accepted
    huge = torch.zeros((1, 16384, 16384), dtype=torch.uint8, device=gpu)   # 16 x 64 tiles: fine
    eng.hu_batch(huge, thresh=127, noalloc=True)
    torch.cuda.synchronize()
