"""Random geometry for the stages in front of the feature path (view renderer, centre/scale, farthest point
sampling) against the CPU oracle: tools/stagefuzz.py's cases with a fixed seed."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_geometry_of_the_front_stages(gpu):
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import stagefuzz
    rng = np.random.default_rng(20261112)
    for _ in range(60):
        stagefuzz.run_case(rng)


def test_random_geometry_of_the_remaining_entry_points(gpu, c_oracle):
    """tools/miscfuzz.py's cases (binarisation thresholds, gray fit, row bands, per-sample mean/std), a fixed seed."""
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import miscfuzz
    rng = np.random.default_rng(20261112)
    for _ in range(80):
        miscfuzz.run_case(rng, c_oracle)
