"""Parity of the HIP point-set PCA path (through the C ABI) with the CPU oracle."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu

EIG_RTOL = 1e-9    # eigenvalues in float64, relative to lambda1 (and per value where well conditioned)
OUT_RTOL = 1e-6    # the nine scalars after the float32 cast


def compare_set(got_out, got_eig, ref_out, ref_eig, n, strict_rel=True):
    """Returns the number of histogram counts that moved across a bin edge."""
    lam, lam_ref = got_eig[:3], ref_eig[:3]
    assert np.all(np.abs(lam - lam_ref) <= EIG_RTOL * max(lam_ref[0], 1e-300))
    if strict_rel:
        # per-value relative bound wherever the value is not itself at rounding level
        big = lam_ref >= 1e-4 * lam_ref[0]
        assert np.all(np.abs(lam - lam_ref)[big] <= EIG_RTOL * np.abs(lam_ref)[big])
    gap = lam_ref[0] - lam_ref[1]
    if gap > 1e-6 * lam_ref[0]:
        assert abs(abs(np.dot(got_eig[3:], ref_eig[3:])) - 1.0) <= 1e-9
    # after the float32 cast: relative 1e-6, plus an absolute floor for values that
    # are rounding noise themselves (lambda3, r3 of rank-deficient sets)
    l1 = max(float(lam_ref[0]), 1e-300)
    atol = np.array([l1, l1, l1, 1.0, 1.0, 1.0, 0.0, 0.0, l1]) * 1e-9
    # ratio12 = l1 / (l2 + 1e-12), ratio23 = l2 / (l3 + 1e-12): what the eigenvalue bar (1e-9 l1 absolute) lets the
    # denominator move shows in the ratio; next to nothing for a full-rank set, everything where l3 is rounding noise
    atol[6] = abs(float(ref_out[6])) * EIG_RTOL * l1 / (float(lam_ref[1]) + 1e-12)
    atol[7] = abs(float(ref_out[7])) * EIG_RTOL * l1 / (float(lam_ref[2]) + 1e-12)
    assert np.all(np.abs(got_out[:9].astype(np.float64) - ref_out[:9]) <= OUT_RTOL * np.abs(ref_out[:9]) + atol)
    moved = np.abs(got_out[9:].astype(np.float64) - ref_out[9:].astype(np.float64)) * n
    assert moved.max() <= 2.5, "histogram differs by more than edge ties"
    return int(np.round(moved.sum() / 2))


def test_reference_golden_vectors(gpu, oracle, pca_golden):
    """Outputs of the reference's own compute_pca_features (tests/golden/pca_golden.npz)."""
    mod = pkg("extract_3d_pca_features")
    ties = 0
    for case in pca_golden:
        pts = case["pts"]
        assert np.array_equal(pts.astype(np.float32).astype(np.float64), pts)
        out, eig = mod.compute_pca_features_batch([pts], case["n_bins"], case["normalize"], return_eig=True)
        ref_out, ref_eig = oracle.compute_pca_features(pts, case["n_bins"], case["normalize"], return_eig=True)
        np.testing.assert_array_equal(ref_out, case["out"])  # oracle == reference output
        degenerate = ref_eig[2] <= 1e-9 * max(ref_eig[0], 1e-300)
        ties += compare_set(out[0], eig[0], case["out"], ref_eig, pts.shape[0], strict_rel=not degenerate)
        if np.all(pts[:, 2] == pts[0, 2]):
            assert eig[0][2] == 0.0 and ref_eig[2] == 0.0  # a constant coordinate gives an exact zero, as LAPACK does
    assert ties <= 1


def test_float64_inputs_are_evaluated_as_float64(gpu, oracle, pca_golden_wide):
    """load_sample (reference :48-56) hands compute_pca_features the float64 values of whatever the file holds.
    Inputs that float32 does not hold exactly take fx_pca_batch_f64: outputs of the reference itself
    (tests/golden/pca_golden_wide.npz) at the bar of the float32-exact inputs, one by one and as one ragged batch;
    rounding such an input to float32 first would miss that bar by orders of magnitude far from the origin."""
    import torch
    mod, eng = pkg("extract_3d_pca_features"), pkg("engine")
    ties, narrowed_fails = 0, 0
    for case in pca_golden_wide:
        pts = case["pts"]
        out, eig = mod.compute_pca_features_batch([pts], case["n_bins"], case["normalize"], return_eig=True)
        ref_out, ref_eig = oracle.compute_pca_features(pts, case["n_bins"], case["normalize"], return_eig=True)
        np.testing.assert_array_equal(ref_out, case["out"])
        ties += compare_set(out[0], eig[0], case["out"], ref_eig, pts.shape[0])
        _, eig32 = oracle.compute_pca_features(pts.astype(np.float32).astype(np.float64), case["n_bins"], case["normalize"],
                                               return_eig=True)
        narrowed_fails += int(np.abs(eig32[:3] - ref_eig[:3]).max() > EIG_RTOL * ref_eig[0])
    assert ties <= 2
    assert narrowed_fails >= 3      # the fixture does tell float64 evaluation from float32 rounding
    # one ragged float64 batch through the entry point (default settings only)
    plain = [c for c in pca_golden_wide if c["n_bins"] == 8 and not c["normalize"]]
    flat = torch.from_numpy(np.concatenate([c["pts"] for c in plain])).to(gpu)
    off = np.zeros(len(plain) + 1, dtype=np.int64)
    np.cumsum([len(c["pts"]) for c in plain], out=off[1:])
    for total_known in (True, False):
        pts_arg = flat if total_known else torch.cat([flat, torch.zeros((5, 3), dtype=torch.float64, device=gpu)])
        res = eng.pca_batch(pts_arg, torch.from_numpy(off).to(gpu), 8, False, eig=True)
        for k, c in enumerate(plain):
            _, ref_eig = oracle.compute_pca_features(c["pts"], 8, False, return_eig=True)
            compare_set(res["out"][k].cpu().numpy(), res["eig"][k].cpu().numpy(), c["out"], ref_eig, len(c["pts"]))
    # a uniform float64 batch, sets of fewer than three points: NaN rows
    uni = torch.from_numpy(np.stack([c["pts"][:1024] for c in pca_golden_wide if len(c["pts"]) >= 1024])).to(gpu)
    res = eng.pca_batch(uni, None, 8, False, eig=True)
    for k in range(uni.shape[0]):
        h = uni[k].cpu().numpy()
        ref_out, ref_eig = oracle.compute_pca_features(h, 8, False, return_eig=True)
        compare_set(res["out"][k].cpu().numpy(), res["eig"][k].cpu().numpy(), ref_out, ref_eig, 1024)
    tiny = eng.pca_batch(uni[:, :2].contiguous(), None, 8, False)["out"]
    assert torch.isnan(tiny).all()


def test_single_set_signature_and_errors(gpu, oracle):
    mod = pkg("extract_3d_pca_features")
    pts = pkg("synth").points_host(1, 1024, seed=3)[0].astype(np.float64)
    out = mod.compute_pca_features(pts, n_bins=8, normalize=False)
    assert out.dtype == np.float32 and out.shape == (17,)
    np.testing.assert_allclose(out[:9], oracle.compute_pca_features(pts)[:9], rtol=OUT_RTOL)
    with pytest.raises(ValueError, match=r"too few points \(<3\)"):
        mod.compute_pca_features(np.zeros((2, 3)))


@pytest.mark.parametrize("N,normalize,nb", [(1024, False, 8), (1024, True, 8), (1000, False, 8), (65, False, 4),
                                            (2048, False, 8), (3000, False, 8), (20000, True, 16)])
def test_device_clouds_vs_oracle(gpu, oracle, N, normalize, nb):
    import torch
    eng = pkg("engine")
    B = 24 if N <= 4096 else 4
    pts = eng.synth_points(B, N, seed=42, first_index=5, device=gpu)
    res = eng.pca_batch(pts, None, nb, normalize, eig=True)
    torch.cuda.synchronize()
    host = pts.cpu().numpy().astype(np.float64)
    out, eig = res["out"].cpu().numpy(), res["eig"].cpu().numpy()
    ties = 0
    for b in range(B):
        ref_out, ref_eig = oracle.compute_pca_features(host[b], nb, normalize, return_eig=True)
        ties += compare_set(out[b], eig[b], ref_out, ref_eig, N)
    assert ties <= 2


def test_ragged_batch_and_2d(gpu, oracle):
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(9)
    sizes = [3, 1024, 2, 77, 5000, 2048, 2049, 1, 0, 300]
    sets = [(rng.normal(size=(n, 3)) * np.array([1.0, 0.4, 0.1])).astype(np.float32) for n in sizes]
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    flat = torch.from_numpy(np.concatenate(sets)).to(gpu)
    res = eng.pca_batch(flat, torch.from_numpy(off).to(gpu), 8, False, eig=True)
    out, eig = res["out"].cpu().numpy(), res["eig"].cpu().numpy()
    for b, n in enumerate(sizes):
        if n < 3:
            assert np.isnan(out[b]).all()
            continue
        ref_out, ref_eig = oracle.compute_pca_features(sets[b].astype(np.float64), 8, False, return_eig=True)
        compare_set(out[b], eig[b], ref_out, ref_eig, n)
    # 2-D input == 3-D input with z = 0
    p2 = (rng.normal(size=(4, 500, 2)) * np.array([2.0, 0.5])).astype(np.float32)
    r2 = eng.pca_batch(torch.from_numpy(p2).to(gpu), None, 8, False, eig=True)
    p3 = np.concatenate([p2, np.zeros((4, 500, 1), np.float32)], axis=2)
    r3 = eng.pca_batch(torch.from_numpy(p3).to(gpu), None, 8, False, eig=True)
    # the 2-D call loads point by point, the 3-D one four points at a time: same values up to summation order
    assert torch.allclose(r2["out"], r3["out"], rtol=1e-6, atol=0)
    assert torch.allclose(r2["eig"][:, :3], r3["eig"][:, :3], rtol=1e-12, atol=0)
    assert torch.all((r2["eig"][:, 3:] * r3["eig"][:, 3:]).sum(dim=1).abs() > 1 - 1e-12)
    assert torch.all(r2["eig"][:, 2] == 0) and torch.all(r3["eig"][:, 2] == 0)


@pytest.mark.parametrize("normalize", [False, True])
def test_variable_size_clouds_tail_config(gpu, oracle, normalize):
    """Clouds with log-uniform sizes from 1e3 to 1.5e6 points and one of 4e6 in one ragged batch
    (the load-imbalance configuration): sets up to 2048 points take the register kernels, larger
    ones are cut into 8192-point chunks spread over the chip."""
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(31)
    sizes = [int(v) for v in np.exp(rng.uniform(np.log(1e3), np.log(1.5e6), size=12))] + [16384, 16385, 2049, 8192, 8193, 1_200_000, 4_000_000]
    sets = []
    for n in sizes:
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        sets.append(((rng.normal(size=(n, 3)) * np.array([0.9, 0.3, 0.12])) @ q.T + rng.normal(size=3)).astype(np.float32))
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    flat = torch.from_numpy(np.concatenate(sets)).to(gpu)
    d_off = torch.from_numpy(off).to(gpu)
    res = eng.pca_batch(flat, d_off, 8, normalize, eig=True)     # work lists built on the device, no host round trip
    out, eig = res["out"].cpu().numpy(), res["eig"].cpu().numpy()
    ties = 0
    for b, n in enumerate(sizes):
        ref_out, ref_eig = oracle.compute_pca_features(sets[b].astype(np.float64), 8, normalize, return_eig=True)
        ties += compare_set(out[b], eig[b], ref_out, ref_eig, n)
    assert ties <= 3
    # the same batch with the point count withheld (N = 0): the library reads the offsets back and builds the
    # lists on the host; both routes run the same kernels on the same sets
    out2 = torch.empty_like(res["out"])
    eig2 = torch.empty_like(res["eig"])
    pkg("_lib").call("fx_pca_batch_f32", flat.data_ptr(), d_off.data_ptr(), len(sizes), 0, 3, 8, int(normalize),
                     out2.data_ptr(), eig2.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2.cpu().numpy(), out, rtol=1e-6, atol=0)
    np.testing.assert_allclose(eig2.cpu().numpy()[:, :3], eig[:, :3], rtol=1e-12, atol=0)


@pytest.mark.parametrize("copies", [1, 400])
def test_point_count_smaller_than_the_offsets_say(gpu, copies):
    """The caller's point count bounds what may be read: sets that reach beyond it get NaN rows, whichever
    kernel their size would have sent them to; the sets inside it are computed as ever.  (A batch of few sets
    takes one launch for all its sets below the large clouds, a larger one a launch per size class: both.)"""
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(77)
    sizes = [100, 5000, 300, 20000, 3, 1500] * copies
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    flat = torch.from_numpy(rng.normal(size=(int(off[-1]), 3)).astype(np.float32)).to(gpu)
    d_off = torch.from_numpy(off).to(gpu)
    full = eng.pca_batch(flat, d_off, 8, False)["out"].cpu().numpy()
    assert np.isfinite(full).all()
    out = torch.zeros((len(sizes), 17), dtype=torch.float32, device=gpu)
    first = len(sizes) - 6   # (the cuts fall into the last copy of the six sizes)
    for cut in (int(off[first + 2]) + 10, int(off[first + 3]) + 4096, int(off[first + 1])):
        out.zero_()
        pkg("_lib").call("fx_pca_batch_f32", flat.data_ptr(), d_off.data_ptr(), len(sizes), cut, 3, 8, 0,
                         out.data_ptr(), 0, int(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for b in range(len(sizes)):
            if off[b + 1] <= cut:
                np.testing.assert_array_equal(got[b], full[b])
            else:
                assert np.isnan(got[b]).all(), (cut, b, got[b])


def test_offsets_that_overlap_cannot_overrun_a_list(gpu):
    """The launches over the device-built lists are sized by the caller's point count.  Offsets that go back and
    forth can name more sets of a size than that count has room for: the surplus gets NaN rows, the sets the
    lists hold are computed as ever, nothing is read or written out of bounds."""
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(5)
    flat = torch.from_numpy(rng.normal(size=(3000, 3)).astype(np.float32)).to(gpu)
    for n in (600, 1500, 2500):   # register list, LDS list, large clouds
        B = 4101   # (more sets than a batch that takes one launch for all its sets may have)
        off = np.zeros(B + 1, dtype=np.int64)
        off[1::2] = n                       # sets 0, 2, 4, ... are points 0 .. n; the odd ones run backwards
        d_off = torch.from_numpy(off).to(gpu)
        out = eng.pca_batch(flat, d_off, 8, False)["out"].cpu().numpy()
        ref = eng.pca_batch(flat[:n].reshape(1, n, 3).contiguous(), None, 8, False)["out"].cpu().numpy()[0]
        assert np.isnan(out[1::2]).all()
        rows = out[0::2]
        good = ~np.isnan(rows).any(axis=1)
        assert 1 <= good.sum() <= len(rows) and (np.isnan(rows[~good])).all()
        np.testing.assert_allclose(rows[good], np.broadcast_to(ref, rows[good].shape), rtol=1e-6, atol=1e-7)


def test_full_size_properties(gpu):
    """C4-sized batch: invariants that need no CPU pass.  Rotating a cloud
    leaves the eigenvalues unchanged; the histogram sums to one; r1+r2+r3 = 1."""
    import torch
    eng = pkg("engine")
    B, N = 100_000, 1024
    pts = eng.synth_points(B, N, seed=42, device=gpu)
    res = eng.pca_batch(pts, None, 8, False, eig=True)
    out, eig = res["out"], res["eig"]
    assert torch.all(torch.isfinite(out))
    assert torch.all((out[:, 9:].sum(dim=1) - 1).abs() < 1e-6)
    assert torch.all((out[:, 3:6].sum(dim=1) - 1).abs() < 1e-6)
    assert torch.all(eig[:, 0] >= eig[:, 1]) and torch.all(eig[:, 1] >= eig[:, 2]) and torch.all(eig[:, 2] >= 0)
    # permutation of the coordinate axes is an exact rotation in float32
    perm = pts[:, :, [2, 0, 1]].contiguous()
    eig2 = eng.pca_batch(perm, None, 8, False, eig=True)["eig"]
    assert torch.all((eig2[:, :3] - eig[:, :3]).abs() <= 1e-12 * eig[:, :1])


def test_full_batch_one_million_sets(gpu, oracle):
    """The whole C4 batch in one launch (250 000 blocks): invariants over all
    rows, the oracle on rows from both ends and the middle of the grid, and
    equality with the same sets computed in small launches."""
    import torch
    eng = pkg("engine")
    B, N = 1_000_000, 1024
    pts = eng.synth_points(B, N, seed=42, device=gpu)
    res = eng.pca_batch(pts, None, 8, False, eig=True)
    out, eig = res["out"], res["eig"]
    assert torch.all(torch.isfinite(out))
    assert torch.all((out[:, 9:].sum(dim=1) - 1).abs() < 1e-6)
    assert torch.all((out[:, 3:6].sum(dim=1) - 1).abs() < 1e-6)
    assert torch.all(eig[:, 0] >= eig[:, 1]) and torch.all(eig[:, 1] >= eig[:, 2]) and torch.all(eig[:, 2] >= 0)
    rows = [0, 1, 2, 3, 4, 499_999, 500_000, B - 5, B - 4, B - 3, B - 2, B - 1]
    sel = torch.tensor(rows, device=gpu)
    host, o, e = pts[sel].cpu().numpy().astype(np.float64), out[sel].cpu().numpy(), eig[sel].cpu().numpy()
    ties = 0
    for k in range(len(rows)):
        ref_out, ref_eig = oracle.compute_pca_features(host[k], 8, False, return_eig=True)
        ties += compare_set(o[k], e[k], ref_out, ref_eig, N)
    assert ties <= 2
    # the tail of the grid computed again as a launch of its own: bit-identical rows
    again = eng.pca_batch(pts[B - 4099:].contiguous(), None, 8, False, eig=True)
    assert torch.equal(again["out"], out[B - 4099:]) and torch.equal(again["eig"], eig[B - 4099:])


def test_ragged_batch_of_many_small_sets_and_bad_offsets(gpu, oracle):
    """A CSR batch whose lists are built on the device: 20 000 sets of 0 .. 1500
    points (empty and tiny sets give NaN rows), 3-D and 2-D; and offsets that
    decrease somewhere, which the device route answers with a NaN row."""
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(8)
    sizes = rng.integers(0, 1500, size=20000)
    sizes[:6] = [0, 1, 2, 3, 1024, 1025]
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    for dims in (3, 2):
        pts = (rng.normal(size=(int(off[-1]), dims)) * np.array([1.0, 0.4, 0.1][:dims])).astype(np.float32)
        res = eng.pca_batch(torch.from_numpy(pts).to(gpu), torch.from_numpy(off).to(gpu), 8, False, eig=True)
        out, eig = res["out"].cpu().numpy(), res["eig"].cpu().numpy()
        assert np.isnan(out[:3]).all() and not np.isnan(out[3:][sizes[3:] >= 3]).any()
        assert np.isnan(out[sizes < 3]).all()
        ties = 0
        for b in list(range(3, 40)) + list(range(19950, 20000)):
            if sizes[b] < 3:
                continue
            p = pts[off[b]:off[b + 1]].astype(np.float64)
            if dims == 2:
                p = np.concatenate([p, np.zeros((len(p), 1))], axis=1)
            ref_out, ref_eig = oracle.compute_pca_features(p, 8, False, return_eig=True)
            got = out[b].copy()
            if ref_eig[2] < 1e-9 * ref_eig[0]:
                # lambda3 of a planar set (2-D input, or three points) is rounding noise: exactly 0 here, 1e-17
                # in LAPACK; ratio23 = lambda2 / (lambda3 + 1e-12) shows that noise at the 1e-5 level
                assert eig[b, 2] <= 1e-9 * ref_eig[0] and abs(got[7] / ref_out[7] - 1.0) < 1e-3
                got[7] = ref_out[7]
            ties += compare_set(got, eig[b], ref_out, ref_eig, int(sizes[b]))
        assert ties <= 4
    bad = off.copy()
    bad[11] = bad[10] - 5          # set 10 gets a negative size, set 11 a larger one
    pts = rng.normal(size=(int(off[-1]), 3)).astype(np.float32)
    out = eng.pca_batch(torch.from_numpy(pts).to(gpu), torch.from_numpy(bad).to(gpu), 8, False)["out"].cpu().numpy()
    assert np.isnan(out[10]).all() and not np.isnan(out[12]).any()


@pytest.mark.parametrize("N,dims,normalize,nb", [(3, 3, False, 8), (5, 3, True, 8), (16, 3, False, 4), (17, 2, False, 8),
                                                 (64, 3, False, 8), (64, 2, True, 8), (65, 3, False, 8),
                                                 (100, 3, True, 16), (128, 3, True, 8), (129, 2, False, 8),
                                                 (255, 3, False, 8), (256, 3, False, 8),
                                                 (256, 2, True, 8), (200, 3, False, 64)])
def test_small_sets_four_per_wave(gpu, oracle, N, dims, normalize, nb):
    """Uniform batches of up to 256 points per set (four sets share a wave), including the sets that path hands
    on: identical points, collinear points, two tight clusters, a set far from the origin; 2-D sets are (x, y, 0).
    The batch size leaves the last wave with one set.  (The kernel keeps 4, 8 or 16 points per lane: sets of up to
    64, 128, 256 points.)"""
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(1000 * N + dims)
    B = 41
    pts = (rng.normal(size=(B, N, dims)) * rng.uniform(0.05, 3.0, size=(B, 1, dims))).astype(np.float32)
    pts[1] = pts[1, 0]                                                  # identical points
    pts[2] = np.outer(np.linspace(-1.0, 2.0, N), pts[2, 0])             # a line through the origin
    pts[3, : N // 2] = pts[3, 0]                                        # two clusters (points repeated)
    pts[3, N // 2:] = pts[3, N - 1]
    pts[4] += np.float32(4096.0)                                        # far from the origin
    pts[5] *= np.float32(1e-12)                                         # tiny
    res = eng.pca_batch(torch.from_numpy(pts).to(gpu), None, nb, normalize, eig=True)
    torch.cuda.synchronize()
    out, eig = res["out"].cpu().numpy(), res["eig"].cpu().numpy()
    ties = 0
    for b in range(B):
        p3 = np.zeros((N, 3), dtype=np.float64)
        p3[:, :dims] = pts[b]
        ref_out, ref_eig = oracle.compute_pca_features(p3, nb, normalize, return_eig=True)
        degenerate = ref_eig[2] <= 1e-9 * max(ref_eig[0], 1e-300)
        if b in (1, 2, 3):   # every point on a bin edge or two: the counts themselves
            lam1 = max(ref_eig[0], 1e-300)
            assert np.all(np.abs(eig[b][:3] - ref_eig[:3]) <= 1e-9 * lam1)
            if b == 1:
                np.testing.assert_array_equal(out[b, 9:], ref_out[9:])
            continue
        got = out[b].copy()
        if degenerate:
            # lambda3 is rounding noise (up to 1e-16 lambda1 on either side of zero, a clipped negative where it
            # falls below), and ratio23 = lambda2 / (lambda3 + 1e-12) shows it at the 1e-3 level
            assert abs(got[7] - ref_out[7]) <= 2e-2 * abs(ref_out[7])
            got[7] = ref_out[7]
        ties += compare_set(got, eig[b], ref_out, ref_eig, N, strict_rel=not degenerate)
    assert ties <= 3


@pytest.mark.parametrize("dims", [3, 2])
def test_ragged_small_sets_of_all_three_sizes_in_one_launch(gpu, oracle, dims):
    """A CSR batch whose small sets fall into the three lists of the small-set launch (up to 64, 128, 256 points),
    the sizes next to the limits among them, list lengths that are not multiples of four; every set against the
    oracle."""
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(31 + dims)
    sizes = np.concatenate([[3, 4, 63, 64, 65, 127, 128, 129, 255, 256, 257, 0, 2], rng.integers(3, 300, size=2150)])
    rng.shuffle(sizes)
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    pts = (rng.normal(size=(int(off[-1]), dims)) * np.array([1.0, 0.4, 0.1][:dims])).astype(np.float32)
    res = eng.pca_batch(torch.from_numpy(pts).to(gpu), torch.from_numpy(off).to(gpu), 8, False, eig=True)
    out, eig = res["out"].cpu().numpy(), res["eig"].cpu().numpy()
    ties = 0
    for b, n in enumerate(sizes):
        if n < 3:
            assert np.isnan(out[b]).all()
            continue
        p = np.zeros((n, 3), dtype=np.float64)
        p[:, :dims] = pts[off[b]:off[b + 1]]
        ref_out, ref_eig = oracle.compute_pca_features(p, 8, False, return_eig=True)
        got = out[b].copy()
        degenerate = ref_eig[2] <= 1e-9 * ref_eig[0]
        if degenerate:   # (2-D sets, three points: lambda3 is rounding noise, and ratio23 shows it)
            assert abs(got[7] - ref_out[7]) <= 2e-2 * abs(ref_out[7])
            got[7] = ref_out[7]
        ties += compare_set(got, eig[b], ref_out, ref_eig, int(n), strict_rel=not degenerate)
    assert ties <= 3


@pytest.mark.parametrize("N", [64, 256, 700, 1024, 1500])
def test_non_finite_points_do_not_disturb_their_neighbours(gpu, oracle, N):
    """A NaN or an infinity in one set (the reference's eigh raises or returns NaN for it): the call returns, that
    set's row is not finite, the other sets of the batch are what they are without it."""
    import torch
    eng = pkg("engine")
    rng = np.random.default_rng(N)
    pts = (rng.normal(size=(12, N, 3)) * np.array([1.0, 0.5, 0.2])).astype(np.float32)
    clean = eng.pca_batch(torch.from_numpy(pts).to(gpu), None, 8, False)["out"].cpu().numpy()
    bad = pts.copy()
    bad[3, N // 2, 1] = np.nan
    bad[7, 0, 0] = np.inf
    bad[9, N - 1, 2] = -np.inf
    out = eng.pca_batch(torch.from_numpy(bad).to(gpu), None, 8, False)["out"].cpu().numpy()
    torch.cuda.synchronize()
    for b in range(12):
        if b in (3, 7, 9):
            assert not np.isfinite(out[b]).all()
        else:
            np.testing.assert_array_equal(out[b], clean[b])


def test_random_batch_geometry_against_oracle(gpu):
    """tools/pcafuzz.py's cases (equal-size and CSR batches, sets from none to tens of thousands of points in one call,
    2-D / 3-D, float32 / float64, 1 .. 64 bins, both normalize settings, identical and collinear points), a fixed
    seed."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import pcafuzz
    rng = np.random.default_rng(20261112)
    for _ in range(80):
        pcafuzz.run_case(rng, gpu)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_csr_batch_without_a_single_point(gpu, dtype):
    """Every set empty: the caller has no coordinates to point at.  NaN rows, as for any set of fewer than three
    points; offsets that speak of points while the pointer is null are refused."""
    import torch
    eng, lib = pkg("engine"), pkg("_lib")
    pts = torch.zeros((0, 3), dtype=torch.float64 if dtype == np.float64 else torch.float32, device=gpu)
    off = torch.zeros(6, dtype=torch.int64, device=gpu)
    res = eng.pca_batch(pts, off, 8, False, eig=True)
    torch.cuda.synchronize()
    assert tuple(res["out"].shape) == (5, 17) and bool(torch.isnan(res["out"]).all()) and bool(torch.isnan(res["eig"]).all())
    with pytest.raises(lib.FxError, match="NULL argument"):
        eng.pca_batch(pts, torch.tensor([0, 4, 9], dtype=torch.int64, device=gpu), 8, False)
