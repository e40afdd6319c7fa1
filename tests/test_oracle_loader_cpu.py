"""CPU: the oracle builds its inputs with its OWN code (oracle/mesh_loader.py) -- and that code and the product's loaders agree bit for bit
on what the parity tests feed them: mesh arrays (incl. the derived smooth normals), padded point sets, labels, SO(3) grids.  Reference
path: lib3d/rigid_mesh_database.py:49-104 (trimesh.load + pad_stack_tensors), utils/transform_utils.py:27-50 (grid)."""
import re
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_no_oracle_module_imports_the_product_package():
    """The dependency goes one way only: tests import both; the oracle imports nothing of megapose6d_amd (a data file path is not code)."""
    pat = re.compile(r"^\s*(from\s+megapose6d_amd|import\s+megapose6d_amd)", re.M)
    offenders = [p.name for p in (ROOT / "oracle").glob("*.py") if pat.search(p.read_text())]
    assert offenders == [], offenders
    # ... and the product never imports the oracle
    pat2 = re.compile(r"^\s*(from\s+oracle|import\s+oracle)", re.M)
    assert [p.name for p in (ROOT / "megapose6d_amd").glob("*.py") if pat2.search(p.read_text())] == []


def test_oracle_loader_equals_product_loader_on_the_parity_datasets(tmp_path):
    from megapose6d_amd import mesh_io
    from megapose6d_amd.mesh_db import MeshDataBase
    from oracle import mesh_loader
    from tests.support import synthetic as syn

    ds = syn.make_object_dataset(tmp_path, n_objects=3, seed=5, n_theta=40, n_z=31)
    ds2 = syn.make_object_dataset(tmp_path / "b", n_objects=2, seed=9, n_theta=24, n_z=20)
    for o in ds2.list_objects:
        o.label = "b_" + o.label
    objs = list(ds.list_objects) + list(ds2.list_objects)                               # -> ragged point sets: the padding path runs

    class _DS:
        list_objects = objs

        def __len__(self):
            return len(objs)

        def __getitem__(self, i):
            return objs[i]

    meshes, sets = mesh_loader.load_dataset(_DS())
    for o in objs:
        ref = mesh_io.load_rigid_object(o)
        for key in ("vertices", "normals", "colors", "faces", "points"):
            assert meshes[o.label][key].dtype == ref[key].dtype and np.array_equal(meshes[o.label][key], ref[key]), (o.label, key)
    db = MeshDataBase.from_object_ds(_DS()).batched()
    assert list(sets.labels) == list(db.labels)
    assert sets.points.dtype == db.points.dtype and torch.equal(sets.points, db.points)
    assert sets.points.shape[1] == max(len(m["points"]) for m in meshes.values()) > min(len(m["points"]) for m in meshes.values())


def test_oracle_ply_reader_on_a_hand_written_ascii_file(tmp_path):
    """ascii PLY with normals, uchar colours, a comment, a quad (fan-triangulated) and an extra scalar face property"""
    from megapose6d_amd import mesh_io
    from oracle import mesh_loader

    text = """ply
format ascii 1.0
comment hand written
element vertex 5
property float x
property float y
property float z
property float nx
property float ny
property float nz
property uchar red
property uchar green
property uchar blue
element face 2
property list uchar int vertex_indices
property int flags
end_header
0 0 0 0 0 1 255 0 0
10 0 0 0 0 1 0 255 0
10 10 0 0 0 1 0 0 255
0 10 0 0 0 1 128 128 128
5 5 7.5 0 1 0 10 20 30
4 0 1 2 3 7
3 0 1 4 9
"""
    p = tmp_path / "hand.ply"
    p.write_text(text)
    got = mesh_loader.read_ply_arrays(p)
    assert got["vertices"].tolist() == [[0, 0, 0], [10, 0, 0], [10, 10, 0], [0, 10, 0], [5, 5, 7.5]]
    assert got["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4]]
    assert got["normals"].tolist() == [[0, 0, 1]] * 4 + [[0, 1, 0]]
    assert np.array_equal(got["colors"], np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128], [10, 20, 30]]) / 255.0)
    ref = mesh_io.read_ply(p)
    for key in ("vertices", "faces", "normals", "colors"):
        assert np.array_equal(np.asarray(ref[key], dtype=np.float64), np.asarray(got[key], dtype=np.float64)), key

    class Obj:
        label, mesh_path, scale = "hand", p, 0.001

    m = mesh_loader.load_object(Obj)
    assert np.allclose(m["vertices"][4], [0.005, 0.005, 0.0075]) and m["faces"].dtype == np.int32
    # smooth normals (a file without normals): the flat square's vertices get +z
    n = mesh_loader.smooth_normals(got["vertices"][:4], got["faces"][:2])
    assert np.array_equal(n, np.array([[0.0, 0.0, 1.0]] * 4))


def test_oracle_so3_grid_equals_product_grid():
    from megapose6d_amd.pose_estimator import load_SO3_grid
    from oracle import mesh_loader

    for n in (72, 512, 576, 4608):
        path = ROOT / "megapose6d_amd" / "data" / f"so3_grid_{n}_xyzw.npy"
        if not path.is_file():
            continue
        assert torch.equal(mesh_loader.load_so3_grid(path), load_SO3_grid(n)), n
