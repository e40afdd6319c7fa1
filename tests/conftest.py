import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def object_dataset(tmp_path_factory):
    from tests.support import synthetic as syn

    return syn.make_object_dataset(tmp_path_factory.mktemp("meshes"), n_objects=3, seed=0)


@pytest.fixture(scope="session")
def engine_meshes(object_dataset):
    from megapose6d_amd import mesh_io

    return [mesh_io.load_rigid_object(o) for o in object_dataset.list_objects]


def assert_logits_close(got, ref, scale):
    """Classifier logits of the HIP path vs the oracle / the reference goldens.  The renders are bit-identical for identical
    cameras, but the crop cameras themselves agree only to the last ulp (fmaf chains on the device, separate torch ops in the
    reference), so once in a while ONE silhouette sample (a quarter of a pixel's 8-bit value under 4x MSAA) flips and moves a logit
    a little further.  `scale` = max(1, |logit|) -- the seeded networks' features are O(1), there is no feature-scale factor.
    Bound: 90 % of the logits within 1e-4 * scale (pure fp32 round-off), every logit within 2e-4 * scale."""
    import numpy as np

    err = np.abs(np.asarray(got, dtype=np.float64).ravel() - np.asarray(ref, dtype=np.float64).ravel())
    assert err.size > 0
    assert np.quantile(err, 0.9) < 1e-4 * scale, (np.quantile(err, 0.9), scale)
    assert err.max() < 2e-4 * scale, (err.max(), scale)
