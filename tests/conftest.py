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
    from megapose6d_amd import synthetic as syn

    return syn.make_object_dataset(tmp_path_factory.mktemp("meshes"), n_objects=3, seed=0)


@pytest.fixture(scope="session")
def engine_meshes(object_dataset):
    from megapose6d_amd import mesh_io

    return [mesh_io.load_rigid_object(o) for o in object_dataset.list_objects]
