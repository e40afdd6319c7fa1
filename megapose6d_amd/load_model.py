"""Model registry / construction / checkpoint ingestion with the reference's signatures.

NAMED_MODELS, load_named_model: reference src/megapose/utils/load_model.py:8-89
load_pose_models: src/megapose/inference/utils.py:80-148 (run_dir/config.yaml + checkpoint.pth.tar["state_dict"])
check_update_config, create_model_pose: src/megapose/training/pose_models_cfg.py:36-138
change_keys_of_older_models: src/megapose/utils/models_compat.py:17-27
The state_dict layout ingested is exactly the reference checkpoints' (SURVEY.md App. F).
"""
from __future__ import annotations

import os
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple

import torch
import yaml

from .mesh_db import BatchedMeshes, MeshDataBase
from .pose_estimator import PoseEstimator
from .pose_rigid import HipBackbone, PosePredictor
from .renderer import Panda3dBatchRenderer

LOCAL_DATA_DIR = Path(os.environ.get("MEGAPOSE_DATA_DIR", Path(__file__).resolve().parent.parent / "local_data"))

NAMED_MODELS = {
    "megapose-1.0-RGB": {
        "coarse_run_id": "coarse-rgb-906902141", "refiner_run_id": "refiner-rgb-653307694", "requires_depth": False,
        "inference_parameters": {"n_refiner_iterations": 5, "n_pose_hypotheses": 1},
    },
    "megapose-1.0-RGBD": {
        "coarse_run_id": "coarse-rgb-906902141", "refiner_run_id": "refiner-rgbd-288182519", "requires_depth": True,
        "inference_parameters": {"n_refiner_iterations": 5, "n_pose_hypotheses": 1},
    },
    "megapose-1.0-RGB-multi-hypothesis": {
        "coarse_run_id": "coarse-rgb-906902141", "refiner_run_id": "refiner-rgb-653307694", "requires_depth": False,
        "inference_parameters": {"n_refiner_iterations": 5, "n_pose_hypotheses": 5},
    },
    "megapose-1.0-RGB-multi-hypothesis-icp": {
        "coarse_run_id": "coarse-rgb-906902141", "refiner_run_id": "refiner-rgb-653307694", "requires_depth": True,
        "depth_refiner": "ICP",
        "inference_parameters": {"n_refiner_iterations": 5, "n_pose_hypotheses": 5, "run_depth_refiner": True},
    },
}


class Config(SimpleNamespace):
    """Attribute + `in` access, enough for the fields create_model_pose reads (OmegaConf is not required at inference)."""

    def __contains__(self, key) -> bool:
        return hasattr(self, key)

    @staticmethod
    def from_any(cfg) -> "Config":
        if isinstance(cfg, Config):
            return cfg
        if isinstance(cfg, dict):
            return Config(**cfg)
        if hasattr(cfg, "items"):  # OmegaConf DictConfig
            return Config(**{k: v for k, v in cfg.items()})
        return Config(**{k: getattr(cfg, k) for k in dir(cfg) if not k.startswith("_") and not callable(getattr(cfg, k))})


class _CfgLoader(yaml.SafeLoader):
    """SafeLoader + the python-tagged nodes the reference's run directories contain.  The reference reads config.yaml with
    yaml.UnsafeLoader (inference/utils.py:71-75), which instantiates arbitrary classes; released / cosypose-era runs store the
    config as `!!python/object:...TrainingConfig` or `!!python/object:argparse.Namespace` with `pathlib.PosixPath` fields.  Those
    tags are mapped to plain data here (object -> mapping, PosixPath -> str, tuple -> tuple, python/name -> its dotted name):
    no class is imported and no callable is invoked."""


def _construct_py_object(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        m = loader.construct_mapping(node, deep=True)
        for key in ("__dict__", "dictitems", "state"):   # object/new + state layouts
            if isinstance(m.get(key), dict) and len(m) <= 3:
                return m[key]
        return m
    if isinstance(node, yaml.SequenceNode):
        return loader.construct_sequence(node, deep=True)
    return loader.construct_scalar(node)


def _construct_py_apply(loader, suffix, node):
    if isinstance(node, yaml.SequenceNode):
        args = loader.construct_sequence(node, deep=True)
    elif isinstance(node, yaml.MappingNode):
        m = loader.construct_mapping(node, deep=True)
        args = m.get("args", [])
        if "Namespace" in suffix or "dict" in suffix:
            return m.get("kwds", m.get("state", m))
    else:
        args = [loader.construct_scalar(node)]
    if "Path" in suffix:
        return str(Path(*[str(a) for a in args])) if args else ""
    if len(args) == 1:
        return args[0]
    return list(args)


_CfgLoader.add_multi_constructor("tag:yaml.org,2002:python/object:", _construct_py_object)
_CfgLoader.add_multi_constructor("tag:yaml.org,2002:python/object/new:", _construct_py_object)
_CfgLoader.add_multi_constructor("tag:yaml.org,2002:python/object/apply:", _construct_py_apply)
_CfgLoader.add_multi_constructor("tag:yaml.org,2002:python/name:", lambda loader, suffix, node: suffix)
_CfgLoader.add_constructor("tag:yaml.org,2002:python/tuple", lambda loader, node: tuple(loader.construct_sequence(node, deep=True)))


def load_cfg(path) -> Config:
    """run_dir/config.yaml -> Config (reference inference/utils.py:71-75).  Plain (OmegaConf-style) YAML mappings and the
    python-tagged object dumps of older runs both load; anything else raises a ValueError naming the file."""
    try:
        data = yaml.load(Path(path).read_text(), Loader=_CfgLoader)
    except yaml.YAMLError as e:
        raise ValueError(f"{path}: unsupported config.yaml layout ({e}); expected a YAML mapping or a python-tagged "
                         "TrainingConfig / argparse.Namespace dump") from e
    if not isinstance(data, dict):
        raise ValueError(f"{path}: expected a YAML mapping")
    return Config(**data)


def check_update_config(cfg) -> Config:
    """Back-compat defaults for older training configs (pose_models_cfg.py:36-87)."""
    cfg = Config.from_any(cfg)
    cfg.is_coarse_compat = False
    if getattr(cfg, "input_strategy", None) == "input=obs+one_render":
        cfg.is_coarse_compat = True
        cfg.n_rendered_views = 1
        cfg.multiview_type = "1view_TCO"
        cfg.predict_rendered_views_logits = True
        cfg.remove_TCO_rendering = True
        cfg.predict_pose_update = False
    renames = {"front_3views": "TCO+front_3views", "front_5views": "TCO+front_5views", "front_1view": "TCO+front_1view"}
    if getattr(cfg, "multiview_type", None) in renames:
        cfg.multiview_type = renames[cfg.multiview_type]
    defaults = {"predict_pose_update": True, "remove_TCO_rendering": False, "predict_rendered_views_logits": False,
                "render_normals": False, "render_depth": False, "input_depth": False}
    for k, v in defaults.items():
        if k not in cfg:
            setattr(cfg, k, v)
    if "n_rendered_views" not in cfg:
        cfg.n_rendered_views = getattr(cfg, "n_views", 1)
    if "multiview_type" not in cfg:
        cfg.multiview_type = "TCO"
    cfg.views_inplane_rotations = getattr(cfg, "views_inplane_rotations", False)
    if "depth_augmentation" not in cfg:
        cfg.depth_normalization_type = "tCR_scale"  # forced for pre-depth-augmentation configs, pose_models_cfg.py:81-82
    if "renderer" not in cfg:
        cfg.renderer = "panda3d"
    return cfg


def n_inputs_from_cfg(cfg) -> int:
    n_in = 3 + (1 if cfg.input_depth else 0)
    per_view = 3 + (3 if cfg.render_normals else 0) + (1 if cfg.render_depth else 0)
    return n_in + per_view * cfg.n_rendered_views


def create_model_pose(cfg, renderer: Panda3dBatchRenderer, mesh_db: BatchedMeshes) -> PosePredictor:
    backbone_str = cfg.backbone_str
    backbone = HipBackbone(backbone_str, n_inputs_from_cfg(cfg))   # incl. "resnet34_width=N" (pose_models_cfg.py:114-116)
    return PosePredictor(
        backbone=backbone, renderer=renderer, mesh_db=mesh_db, render_size=(240, 320), n_rendered_views=cfg.n_rendered_views,
        views_inplane_rotations=cfg.views_inplane_rotations, multiview_type=cfg.multiview_type, render_normals=cfg.render_normals,
        render_depth=cfg.render_depth, input_depth=cfg.input_depth, predict_rendered_views_logits=cfg.predict_rendered_views_logits,
        remove_TCO_rendering=cfg.remove_TCO_rendering, predict_pose_update=cfg.predict_pose_update,
        depth_normalization_type=cfg.depth_normalization_type)


def change_keys_of_older_models(state_dict: Dict[str, Any]) -> Dict[str, Any]:
    remapped = {}
    for key, value in state_dict.items():
        if key.startswith("backbone.backbone"):
            key = "backbone." + key[len("backbone.backbone."):]
        elif key.startswith("backbone.head.0."):
            key = "views_logits_head." + key[len("backbone.head.0."):]
        remapped[key] = value
    return remapped


def build_pose_model(cfg, state_dict: Dict[str, torch.Tensor], renderer, mesh_db_batched) -> PosePredictor:
    cfg = check_update_config(cfg)
    model = create_model_pose(cfg, renderer=renderer, mesh_db=mesh_db_batched)
    model.load_state_dict(change_keys_of_older_models(state_dict))
    model = model.cuda().eval()
    model.cfg = cfg
    model.config = cfg
    return model


def load_pose_models(coarse_run_id: str, refiner_run_id: str, object_dataset, force_panda3d_renderer: bool = False,
                     renderer_kwargs: Optional[dict] = None, models_root: Path = LOCAL_DATA_DIR / "experiments"
                     ) -> Tuple[PosePredictor, PosePredictor, MeshDataBase]:
    kwargs = dict(renderer_kwargs or {})
    kwargs.setdefault("split_objects", True)
    kwargs.setdefault("preload_cache", False)
    kwargs.setdefault("n_workers", 4)
    mesh_db = MeshDataBase.from_object_ds(object_dataset)
    renderer = Panda3dBatchRenderer(object_dataset=object_dataset, **kwargs)  # one renderer serves both models
    mesh_db_batched = mesh_db.batched().cuda()

    def load_model(run_id: str) -> Optional[PosePredictor]:
        if run_id is None:
            return None
        run_dir = Path(models_root) / run_id
        cfg = load_cfg(run_dir / "config.yaml")
        ckpt = torch.load(run_dir / "checkpoint.pth.tar", map_location="cpu")["state_dict"]
        return build_pose_model(cfg, ckpt, renderer, mesh_db_batched)

    return load_model(coarse_run_id), load_model(refiner_run_id), mesh_db


def load_named_model(model_name: str, object_dataset, n_workers: int = 4, bsz_images: int = 128) -> PoseEstimator:
    model = NAMED_MODELS[model_name]
    coarse_model, refiner_model, mesh_db = load_pose_models(
        coarse_run_id=model["coarse_run_id"], refiner_run_id=model["refiner_run_id"], object_dataset=object_dataset,
        force_panda3d_renderer=True, renderer_kwargs={"preload_cache": False, "split_objects": False, "n_workers": n_workers},
        models_root=LOCAL_DATA_DIR / "megapose-models")
    depth_refiner = None
    if model.get("depth_refiner", None) == "ICP":
        from .icp_refiner import ICPRefiner

        depth_refiner = ICPRefiner(mesh_db, refiner_model.renderer)
    return PoseEstimator(refiner_model=refiner_model, coarse_model=coarse_model, detector_model=None, depth_refiner=depth_refiner,
                         bsz_objects=8, bsz_images=bsz_images)


def check_update_config_detector(cfg) -> Config:
    """reference training/detector_models_cfg.py:24-28: labels are prefixed with the dataset name (`ycbv-obj_000001`)"""
    cfg = Config.from_any(cfg)
    obj_prefix = cfg.train_ds_names[0][0].split(".")[0]
    cfg.label_to_category_id = {f"{obj_prefix}-{k}": v for k, v in dict(cfg.label_to_category_id).items()}
    return cfg


def create_model_detector(cfg, n_classes: int):
    """reference training/detector_models_cfg.py:31-38"""
    from .mask_rcnn import DetectorMaskRCNN

    return DetectorMaskRCNN(input_resize=tuple(cfg.input_resize), n_classes=n_classes, backbone_str=cfg.backbone_str,
                            anchor_sizes=tuple(tuple(s) for s in cfg.anchor_sizes))


def load_detector(run_id: str, models_root: Path = LOCAL_DATA_DIR / "experiments"):
    """reference inference/utils.py:57-70: run directory (config.yaml + checkpoint.pth.tar) -> Detector over the HIP Mask R-CNN"""
    from .detector import Detector

    run_dir = Path(models_root) / run_id
    cfg = check_update_config_detector(load_cfg(run_dir / "config.yaml"))
    label_to_category_id = cfg.label_to_category_id
    model = create_model_detector(cfg, len(label_to_category_id))
    ckpt = torch.load(run_dir / "checkpoint.pth.tar", map_location="cpu")["state_dict"]
    model.load_state_dict(ckpt)
    model = model.cuda().eval()
    model.cfg = cfg
    model.config = cfg
    return Detector(model)


def save_run(run_dir, cfg, state_dict: Dict[str, torch.Tensor]) -> None:
    """Write config.yaml + checkpoint.pth.tar in the layout load_pose_models reads (used for the seeded synthetic runs)."""
    run_dir = Path(run_dir)
    run_dir.mkdir(parents=True, exist_ok=True)
    data = {k: v for k, v in vars(Config.from_any(cfg)).items()}
    (run_dir / "config.yaml").write_text(yaml.safe_dump(data))
    torch.save({"state_dict": state_dict, "epoch": 0}, run_dir / "checkpoint.pth.tar")


def make_detections(labels, bboxes, batch_im_ids=None):
    """reference inference/utils.py:214-225 (make_detections_from_object_data) for plain arrays."""
    import numpy as np
    import pandas as pd

    from .tcoll import PandasTensorCollection

    n = len(labels)
    infos = pd.DataFrame(dict(label=list(labels), batch_im_id=0 if batch_im_ids is None else list(batch_im_ids), instance_id=np.arange(n)))
    if batch_im_ids is not None:
        infos["instance_id"] = infos.groupby(["batch_im_id", "label"]).cumcount()
    return PandasTensorCollection(infos=infos, bboxes=torch.as_tensor(np.asarray(bboxes)))
