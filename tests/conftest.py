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


@pytest.fixture(scope="session")
def oracle_meshes(object_dataset):
    """the same objects through the ORACLE's own reader (oracle/mesh_loader.py): what the independent C rasteriser / the oracle pose math
    are fed in the parity tests, so that a defect of the product loader cannot cancel out on both sides"""
    from oracle import mesh_loader

    return [mesh_loader.load_object(o) for o in object_dataset.list_objects]


def assert_logits_close(got, ref, scale):
    """Classifier logits of the HIP path vs the oracle / the reference goldens: `oracle.harness.logit_flip_rule` -- every logit within
    1e-4 x scale (scale = max(1, |logit|): the seeded networks' features are O(1), there is no feature-scale factor) EXCEPT at most one
    row per 64 whose silhouette sample flipped (the crop cameras agree to 1 ulp only), and those within 2e-4 x scale."""
    import numpy as np

    from oracle.harness import logit_flip_rule

    err = np.abs(np.asarray(got, dtype=np.float64).ravel() - np.asarray(ref, dtype=np.float64).ravel())
    r = logit_flip_rule(err, scale)
    assert r["ok"], (r, scale)
