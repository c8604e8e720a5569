import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name: str = ""):
    """The product package (its directory name has a hyphen)."""
    return importlib.import_module("image-processing_amd" + ("." + name if name else ""))


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_ref
    return cpu_ref


@pytest.fixture(scope="session")
def c_oracle(oracle):
    return oracle.load_c_oracle()


@pytest.fixture(scope="session")
def hu_panels():
    d = np.load(GOLDEN / "hu_panels.npz")
    masks = (np.unpackbits(d["masks_packed"], axis=2)[:, :, :224].astype(np.uint8) * 255)
    return {"ids": [str(s) for s in d["ids"]], "masks": masks, "preproc": d["preproc"], "raw_views": d["raw_views"]}


@pytest.fixture(scope="session")
def pca_golden():
    d = np.load(GOLDEN / "pca_golden.npz")
    cases = []
    for k in range(int(d["n_cases"])):
        nb, nz = d[f"meta_{k}"]
        cases.append({"pts": d[f"pts_{k}"], "out": d[f"out_{k}"], "n_bins": int(nb), "normalize": bool(nz)})
    return cases


@pytest.fixture(scope="session")
def pca_golden_wide():
    """Reference outputs for float64 inputs that float32 does not hold exactly."""
    d = np.load(GOLDEN / "pca_golden_wide.npz")
    cases = []
    for k in range(int(d["n_cases"])):
        nb, nz = d[f"meta_{k}"]
        cases.append({"pts": d[f"pts_{k}"], "out": d[f"out_{k}"], "n_bins": int(nb), "normalize": bool(nz)})
    return cases


@pytest.fixture(scope="session")
def gpu():
    """torch device of the MI355X; the tests that use it are marked gpu."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu ran without a HIP device")
    return torch.device("cuda", 0)
