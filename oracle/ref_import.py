"""Import the reference's own Python (read-only, from /root/reference/src) so its
pure-torch hot-path math can generate golden vectors.  TEST INFRASTRUCTURE ONLY and
CONTAINER-ONLY: /root/reference does not exist on the GPU box, so nothing that runs
there (tests -m gpu, smoke(), bench.py) may import this module.  Only
``oracle/make_golden.py`` and the (skipped-when-absent) pinning tests use it.

Recipe:
  * env CONDA_PREFIX / MEGAPOSE_DATA_DIR / CUDA_VISIBLE_DEVICES so megapose.config imports
  * np.float_ shim for NumPy 2 (lib3d/symmetries.py:39-40,49)
  * a sys.meta_path finder serving permissive stub modules for the un-installed
    third-party roots; a minimal real `pinocchio` (Transform(...) runs at import time,
    panda3d_renderer/types.py:40,62,120)
  * then the stubs' hot-path attributes are replaced by the CPU restatements in
    oracle/thirdparty.py: torchvision.ops.roi_align, roma.unitquat_to_rotmat,
    trimesh.load, megapose.lib3d.multiview._get_views_TCO_pos_sphere
  * .cuda()/.pin_memory() patched to identity (no GPU in the container).
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

REFERENCE_SRC = Path("/root/reference/src")

_STUB_ROOTS = {
    "cv2", "transforms3d", "panda3d", "direct", "roma", "trimesh", "torchvision", "omegaconf", "bokeh",
    "simplejson", "open3d", "torchnet", "webdataset", "imageio", "seaborn", "xarray", "colorama",
    "meshcat", "png", "plyfile", "teaserpp_python", "structlog", "selenium", "pytorch3d",
    "bop_toolkit_lib", "PIL", "matplotlib", "ipdb", "pypng", "tqdm_joblib",
}


def available() -> bool:
    return (REFERENCE_SRC / "megapose" / "__init__.py").is_file()


class _Stub(types.ModuleType):
    """Permissive module: any attribute is another stub / a dummy class."""

    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if name[:1].isupper():
            obj = type(name, (), {"__init__": lambda self, *a, **k: None, "__module__": self.__name__})
        else:
            obj = _Stub(full)
            obj.__spec__ = importlib.machinery.ModuleSpec(full, None, is_package=True)
            sys.modules[full] = obj
        setattr(self, name, obj)
        return obj

    def __call__(self, *a, **k):
        return None


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS:
            try:  # prefer the real package when it exists (e.g. PIL, matplotlib)
                if root in ("PIL", "matplotlib") and importlib.util.find_spec(root) is not None:  # type: ignore
                    return None
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


_installed = False


def install() -> None:
    """Make `import megapose` work in this container.  Idempotent."""
    global _installed
    if _installed:
        return
    assert available(), "reference not present (this only runs in the build container)"
    data_dir = Path(tempfile.gettempdir()) / "megapose_oracle_data"
    data_dir.mkdir(exist_ok=True)
    os.environ.setdefault("CONDA_PREFIX", "/usr")
    os.environ.setdefault("MEGAPOSE_DATA_DIR", str(data_dir))
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "0")
    os.environ.setdefault("HOME", "/root")
    if not hasattr(np, "float_"):
        np.float_ = np.float64  # type: ignore[attr-defined]

    from oracle import thirdparty as tp

    pin = types.ModuleType("pinocchio")
    pin.SE3 = tp.SE3
    pin.Quaternion = tp.Quaternion
    sys.modules["pinocchio"] = pin

    import importlib.util  # noqa: F401

    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, str(REFERENCE_SRC))
    omp, mkl = os.environ.get("OMP_NUM_THREADS"), os.environ.get("MKL_NUM_THREADS")
    nthreads = torch.get_num_threads()

    import megapose  # noqa: F401  (forces OMP/MKL threads = 1: src/megapose/__init__.py:39-40)

    # restore the caller's thread settings; the CPU baseline sets them explicitly
    for k, v in (("OMP_NUM_THREADS", omp), ("MKL_NUM_THREADS", mkl)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    torch.set_num_threads(nthreads)

    import roma
    import torchvision
    import trimesh

    torchvision.ops.roi_align = tp.roi_align
    roma.unitquat_to_rotmat = tp.unitquat_to_rotmat
    trimesh.load = tp.trimesh_load
    trimesh.Scene = type("Scene", (), {})
    trimesh.PointCloud = type("PointCloud", (), {})

    import megapose.lib3d.multiview as mv

    mv._get_views_TCO_pos_sphere = tp.get_views_TCO_pos_sphere

    # no GPU here: device moves become identity
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self  # type: ignore[assignment]
        torch.nn.Module.cuda = lambda self, *a, **k: self  # type: ignore[assignment]
        torch.Tensor.pin_memory = lambda self, *a, **k: self  # type: ignore[assignment]
        import megapose.utils.tensor_collection as tcm

        tcm.TensorCollection.cuda = lambda self: self
    _installed = True


def ref():
    """Namespace of the reference symbols the golden generator uses."""
    install()
    ns = types.SimpleNamespace()
    import megapose.inference.pose_estimator as pe
    import megapose.inference.types as ty
    import megapose.inference.utils as iu
    import megapose.lib3d.camera_geometry as cg
    import megapose.lib3d.cosypose_ops as co
    import megapose.lib3d.cropping as cr
    import megapose.lib3d.mesh_ops as mo
    import megapose.lib3d.multiview as mv
    import megapose.lib3d.rigid_mesh_database as rmd
    import megapose.lib3d.rotations as rot
    import megapose.lib3d.transform_ops as to
    import megapose.models.pose_rigid as pr
    import megapose.models.torchvision_resnet as tvr
    import megapose.models.wide_resnet as wr
    import megapose.training.pose_models_cfg as pmc
    import megapose.utils.tensor_collection as tc
    import megapose.utils.transform_utils as tu
    from megapose.datasets.object_dataset import RigidObject, RigidObjectDataset

    ns.pe, ns.ty, ns.iu, ns.cg, ns.co, ns.cr, ns.mo, ns.mv, ns.rmd = pe, ty, iu, cg, co, cr, mo, mv, rmd
    ns.rot, ns.to, ns.pr, ns.tvr, ns.wr, ns.pmc, ns.tc, ns.tu = rot, to, pr, tvr, wr, pmc, tc, tu
    ns.RigidObject, ns.RigidObjectDataset = RigidObject, RigidObjectDataset
    return ns
