"""Feature binarisation thresholds on the device against the reference's outputs
(golden) and the oracle's NumPy restatement: bit-exact float64."""
import numpy as np
import pandas as pd
import pytest
import torch

from conftest import pkg
from test_oracle import threshold_cases

pytestmark = pytest.mark.gpu


def frame(table):
    X = pd.DataFrame(table, columns=[f"f{c}" for c in range(table.shape[1])])
    X.insert(0, "id", [f"s{r}" for r in range(table.shape[0])])
    return X


def test_golden_of_the_reference(gpu):
    b = pkg("binarize")
    for k, table, y, method, strategy, q, thr, bits in threshold_cases():
        X = frame(table)
        got = b.fit_thresholds(X, pd.Series(y), method=method, quantile=q, strategy=strategy)
        assert list(got) == [c for c in X.columns if c != "id"]
        vec = np.array([got[c] for c in got])
        np.testing.assert_array_equal(vec, thr, err_msg=f"case {k}: {method}/{strategy}/{q} {table.shape}")
        out = b.apply_thresholds(X, got)
        assert out.dtype == np.uint8
        np.testing.assert_array_equal(out, bits, err_msg=f"case {k}")


def test_large_tables_against_oracle(gpu, oracle):
    """Sizes of a full extraction run (rows of a 31-column feature table), wide
    tables (more than one column group), values of mixed sign and magnitude,
    negative zeros, denormals and infinities."""
    b = pkg("binarize")
    rng = np.random.default_rng(12)
    for rows, cols, ncls in ((100000, 31, 7), (99999, 3, 2), (5000, 40, 3), (4097, 17, 1)):
        t = rng.normal(size=(rows, cols)) * np.exp(rng.uniform(-8, 8, size=cols))
        t = t.astype(np.float32).astype(np.float64)
        t[rng.integers(0, rows, 50), 0] = 0.0
        t[rng.integers(0, rows, 50), 0] = -0.0
        t[rng.integers(0, rows, 5), cols - 1] = np.inf
        t[rng.integers(0, rows, 5), cols - 1] = -np.inf
        t[:, 1] = rng.integers(-3, 4, rows) * 5e-324
        y = rng.integers(0, ncls, rows)
        x = torch.from_numpy(t).to(gpu)
        classes, codes = b.class_codes(y)
        labels = torch.from_numpy(codes).to(gpu)
        for method, q in (("median", 0.5), ("quantile", 0.123), ("quantile", 1.0), ("quantile", 0.0)):
            for strategy in ("global", "per_class"):
                got = b.fit_thresholds_device(x, labels, len(classes), method, q, strategy)
                want = oracle.fit_thresholds(t, y, method, q, strategy)
                np.testing.assert_array_equal(got, want, err_msg=f"{rows}x{cols} {method} {q} {strategy}")
        thr = torch.from_numpy(oracle.fit_thresholds(t, y, "median")).to(gpu)
        np.testing.assert_array_equal(b.apply_thresholds_device(x, thr).cpu().numpy(), oracle.apply_thresholds(t, thr.cpu().numpy()))
    # Otsu on finite data
    t = (rng.normal(size=(50000, 9)) * np.exp(rng.uniform(-3, 3, size=9)) + rng.normal(size=9)).astype(np.float32).astype(np.float64)
    t[:25000, 4] += 10.0
    y = rng.integers(0, 4, 50000)
    x = torch.from_numpy(t).to(gpu)
    classes, codes = b.class_codes(y)
    for strategy in ("global", "per_class"):
        got = b.fit_thresholds_device(x, torch.from_numpy(codes).to(gpu), len(classes), "otsu", 0.5, strategy)
        np.testing.assert_array_equal(got, oracle.fit_thresholds(t, y, "otsu", 0.5, strategy))


def test_nan_strided_and_errors(gpu, oracle):
    b, eng = pkg("binarize"), pkg("engine")
    rng = np.random.default_rng(3)
    t = rng.normal(size=(101, 5))
    t[7, 2] = np.nan
    x = torch.from_numpy(t).to(gpu)
    with np.errstate(invalid="ignore"):
        want = oracle.fit_thresholds(t, np.zeros(101), "median")
    got = b.fit_thresholds_device(x)
    assert np.isnan(got[2]) and np.isnan(want[2])
    np.testing.assert_array_equal(np.delete(got, 2), np.delete(want, 2))
    with pytest.raises(ValueError, match="not finite"):
        b.fit_thresholds_device(x, method="otsu")
    # a view with padded rows
    wide = torch.zeros((101, 8), dtype=torch.float64, device=gpu)
    wide[:, :5] = x
    res = eng.column_thresholds(wide[:, :5], "quantile", 0.25)
    np.testing.assert_array_equal(np.delete(res["thr"].cpu().numpy(), 2),
                                  np.delete(oracle.fit_thresholds(t, np.zeros(101), "quantile", 0.25), 2))
    assert int(res["count"].item()) == 101
    with pytest.raises(ValueError):
        eng.column_thresholds(x, "mean")
    with pytest.raises(ValueError):
        eng.column_thresholds(x.float())
    with pytest.raises(ValueError):
        b.fit_thresholds(frame(t), pd.Series(np.zeros(101)), method="mode")
