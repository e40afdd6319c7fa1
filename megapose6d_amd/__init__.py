"""megapose6d_amd -- MI355X-native render-and-compare pose engine behind the megapose6d inference API.

Only what the hot path needs: the HIP/C-ABI engine (`csrc/`, `_lib`, `engine`) and the host-side mirror
of the reference's renderer / PosePredictor / PoseEstimator interfaces.  Importing the package never touches the GPU; the
first engine call loads `libmp_engine.so` and raises if it is missing (there is no CPU fallback).
"""
from . import (detector, distributed, engine, icp_refiner, load_model, mask_rcnn, mesh_db, mesh_io, object_dataset, pose_estimator,  # noqa: F401
               pose_rigid, prediction_runner, renderer, tcoll, types)
from .detector import Detector  # noqa: F401
from .icp_refiner import DepthRefiner, ICPRefiner  # noqa: F401
from .mask_rcnn import DetectorMaskRCNN  # noqa: F401
from .object_dataset import RigidObject, RigidObjectDataset  # noqa: F401
from .load_model import NAMED_MODELS, create_model_pose, load_named_model, load_pose_models  # noqa: F401
from .pose_estimator import CoarseRefinePoseEstimator, PoseEstimator  # noqa: F401
from .pose_rigid import PosePredictor  # noqa: F401
from .prediction_runner import PredictionRunner  # noqa: F401
from .renderer import Panda3dBatchRenderer  # noqa: F401
from .types import BatchRenderOutput, ObservationTensor, Panda3dLightData, PosePredictorOutput  # noqa: F401

__version__ = "0.1.0"
