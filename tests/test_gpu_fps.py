"""Point-cloud normalisation and farthest point sampling on the device against
the reference's outputs (golden) and the oracle's NumPy restatement: bit-exact."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, pkg

pytestmark = pytest.mark.gpu


def cloud(rng, M, f32=True, shift=30.0):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    p = (rng.normal(size=(M, 3)) * np.exp(rng.uniform(-1, 2, size=3))) @ q.T + rng.normal(size=3) * shift
    return p.astype(np.float32) if f32 else p


def test_golden_of_the_reference(gpu):
    prep = pkg("prepare_all")
    g = np.load(GOLDEN / "fps_golden.npz")
    for k in range(int(g["n_cases"])):
        pts = g[f"pts_{k}"]
        n, seed = (int(v) for v in g[f"meta_{k}"])
        norm, cen = prep.center_and_scale(pts.astype(np.float64))
        np.testing.assert_array_equal(norm, g[f"norm_{k}"], err_msg=f"case {k}")
        np.testing.assert_array_equal(cen, g[f"cen_{k}"])
        assert norm.dtype == np.float32 and cen.dtype == np.float32
        rng = np.random.RandomState(seed) if seed >= 0 else None
        idx = prep.farthest_point_sampling(norm, n, rng)
        assert idx.dtype == np.int64
        np.testing.assert_array_equal(idx, g[f"idx_{k}"], err_msg=f"case {k}")


def test_center_and_scale_batch_against_oracle(gpu, oracle):
    prep = pkg("prepare_all")
    rng = np.random.default_rng(5)
    clouds = [cloud(rng, M, f32) for M, f32 in ((1, True), (2, True), (3, False), (257, True), (5000, False),
                                                (70000, True), (1000, False))]
    clouds.append(np.tile(np.array([[1.5, -2.0, 0.25]]), (40, 1)))        # all points equal: nothing to scale
    clouds.append(np.zeros((0, 3)))                                        # empty: the reference's early return
    clouds.append(cloud(rng, 4096, False, shift=1e6))                      # far from the origin
    # read_pointcloud hands float64 to center_and_scale whatever the file held
    clouds = [c.astype(np.float64) for c in clouds]
    got = prep.center_and_scale_batch(clouds)
    for k, c in enumerate(clouds):
        want = oracle.center_and_scale(c)
        np.testing.assert_array_equal(got[k][0], want[0], err_msg=f"cloud {k}")
        np.testing.assert_array_equal(got[k][1], want[1], err_msg=f"cloud {k}")


@pytest.mark.parametrize("M,n", [(20, 8), (1025, 1024), (3000, 1024), (16384, 512), (16385, 300), (60000, 1024), (100000, 1024)])
def test_fps_against_oracle(gpu, oracle, M, n):
    prep = pkg("prepare_all")
    rng = np.random.default_rng(M + n)
    pts = oracle.center_and_scale(cloud(rng, M).astype(np.float64))[0]
    for seed in (None, 123):
        r1 = np.random.RandomState(seed) if seed is not None else None
        r2 = np.random.RandomState(seed) if seed is not None else None
        got = prep.farthest_point_sampling(pts, n, r1)
        want = oracle.farthest_point_sampling(pts, n, r2)
        np.testing.assert_array_equal(got, want)
        if seed is not None:                       # the caller's random stream advanced exactly as in the reference
            assert r1.randint(0, 1 << 30) == r2.randint(0, 1 << 30)


def test_fps_ties_and_host_branches(gpu, oracle):
    prep = pkg("prepare_all")
    g = np.stack(np.meshgrid(np.arange(16.0), np.arange(16.0), np.arange(8.0), indexing="ij"), -1).reshape(-1, 3)
    grid = np.concatenate([g, g[::-1]], axis=0).astype(np.float32)        # every point twice: equal maxima throughout
    np.testing.assert_array_equal(prep.farthest_point_sampling(grid, 600), oracle.farthest_point_sampling(grid, 600))
    small = cloud(np.random.default_rng(1), 10)
    for n, seed in ((10, 3), (25, 3), (25, None), (0, 3)):
        r1 = np.random.RandomState(seed) if seed is not None else None
        r2 = np.random.RandomState(seed) if seed is not None else None
        np.testing.assert_array_equal(prep.farthest_point_sampling(small, n, r1),
                                      oracle.farthest_point_sampling(small, n, r2))
    assert prep.farthest_point_sampling(np.zeros((0, 3), np.float32), 5).size == 0
    with pytest.raises(ValueError, match="float32"):
        prep.farthest_point_sampling(np.random.default_rng(0).normal(size=(100, 3)), 10)


def test_fps_large_clouds_structured(gpu, oracle):
    """Clouds beyond the register kernels (cell-sorted kernel): lattices whose distances tie exactly or in the
    last bits, duplicates, clusters far apart, a line, more samples than distinct points."""
    prep = pkg("prepare_all")
    rng = np.random.default_rng(77)
    g = np.stack(np.meshgrid(np.arange(32.0), np.arange(32.0), np.arange(20.0), indexing="ij"), -1).reshape(-1, 3)
    lattice = (g * np.float32(0.1)).astype(np.float32)                      # 20480 points, inexact spacing
    twice = np.concatenate([g, g[::-1]], axis=0).astype(np.float32)        # every point twice
    clusters = np.concatenate([rng.normal(size=(9000, 3)) * 0.01 + c for c in ((0, 0, 0), (5, 0, 0), (0, 7, 1))])
    line = np.zeros((30000, 3)); line[:, 0] = rng.permutation(30000) * 0.001
    few = np.tile(rng.normal(size=(50, 3)), (400, 1))                      # 50 distinct points, 20000 rows
    one = np.tile(np.array([[0.5, -1.0, 2.0]]), (17000, 1))
    cases = [(lattice, 700, None), (twice, 300, 5), (clusters.astype(np.float32), 1024, 1), (line.astype(np.float32), 512, 2),
             (few.astype(np.float32), 120, 3), (one.astype(np.float32), 7, 4)]
    for k, (pts, n, seed) in enumerate(cases):
        r1 = np.random.RandomState(seed) if seed is not None else None
        r2 = np.random.RandomState(seed) if seed is not None else None
        np.testing.assert_array_equal(prep.farthest_point_sampling(pts, n, r1),
                                      oracle.farthest_point_sampling(pts, n, r2), err_msg=f"case {k}")


def test_fps_large_batch_mixed_sizes(gpu, oracle):
    """One call with clouds of every size class, 100 000 and 300 000 points among them (groups of 256 points) and
    one of 600 000 (64^3 grid, cell counters in global memory)."""
    eng = pkg("engine")
    rng = np.random.default_rng(31)
    sizes = [100000, 900, 16385, 300000, 5000, 40000, 600000]
    clouds = [oracle.center_and_scale(cloud(rng, M).astype(np.float64))[0] for M in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)])
    start = np.array([rng.integers(0, M) for M in sizes], dtype=np.int64)
    pts = torch.from_numpy(np.concatenate(clouds)).to(gpu)
    n = 256
    got = eng.fps(pts, off, n, start)
    idx = got["idx"].cpu().numpy()
    for k, c in enumerate(clouds):
        class Fixed:                       # the oracle draws its start index from the caller's generator
            def randint(self, lo, hi, k=k):
                return int(start[k])
        want = oracle.farthest_point_sampling(c, n, Fixed())
        np.testing.assert_array_equal(idx[k], want, err_msg=f"cloud {k}")
        np.testing.assert_array_equal(got["sampled"][k].cpu().numpy(), c[want])


def test_normalize_and_sample_pipeline(gpu, oracle):
    """center_and_scale -> FPS on the float32 result -> gather, as the reference's per-file pipeline."""
    prep = pkg("prepare_all")
    rng = np.random.default_rng(9)
    clouds = [cloud(rng, M, f32=False) for M in (5000, 300, 1500, 20000, 1024)]
    seeds = [5, 6, None, 8, 9]
    got = prep.normalize_and_sample(clouds, 1024, [np.random.RandomState(s) if s is not None else None for s in seeds])
    for k, c in enumerate(clouds):
        norm, cen = oracle.center_and_scale(c)
        idx = oracle.farthest_point_sampling(norm.astype(np.float32), 1024,
                                             np.random.RandomState(seeds[k]) if seeds[k] is not None else None)
        np.testing.assert_array_equal(got[k][2], idx, err_msg=f"cloud {k}")
        np.testing.assert_array_equal(got[k][0], norm[idx].astype(np.float32))
        np.testing.assert_array_equal(got[k][1], cen)


def test_host_arguments_may_be_reused_at_once(gpu, oracle):
    """fx_center_scale / fx_fps_f32 do not wait for the stream: the host offset and start arrays are staged before
    the call returns, so overwriting them right away (and calling again on the same stream) changes nothing."""
    eng = pkg("engine")
    rng = np.random.default_rng(3)
    sizes = [3000, 20000, 700]
    clouds = [cloud(rng, M) for M in sizes]
    pts = torch.from_numpy(np.concatenate(clouds)).to(gpu)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    start = np.array([5, 17, 0], dtype=np.int64)
    results = []
    for rep in range(3):                              # back-to-back calls reuse the stream's staging block
        o, s0 = off.copy(), start.copy()
        cs = eng.center_scale(pts, o)
        fp = eng.fps(cs["points"], o, 64, s0)
        o[:] = 0
        s0[:] = 0                                      # the caller's arrays change while the work is still queued
        results.append((cs["points"], fp["idx"]))
    torch.cuda.synchronize()
    for norm_d, idx_d in results:
        for k, c in enumerate(clouds):
            norm = oracle.center_and_scale(c.astype(np.float64))[0]
            np.testing.assert_array_equal(norm_d[off[k]:off[k + 1]].cpu().numpy(), norm)

            class Fixed:
                def randint(self, lo, hi, k=k):
                    return int(start[k])
            np.testing.assert_array_equal(idx_d[k].cpu().numpy(), oracle.farthest_point_sampling(norm, 64, Fixed()))


def test_fps_rejects_bad_arguments(gpu):
    eng, lib = pkg("engine"), pkg("_lib")
    pts = torch.zeros((100, 3), dtype=torch.float32, device=gpu)
    with pytest.raises(lib.FxError, match="needs more than that many"):
        eng.fps(pts, [0, 100], 100, [0])
    with pytest.raises(lib.FxError, match="start index"):
        eng.fps(pts, [0, 100], 10, [100])
    with pytest.raises(ValueError):
        eng.fps(pts, [0, 50], 10, [0])
    with pytest.raises(ValueError):
        eng.center_scale(pts.to(torch.float16), [0, 100])
