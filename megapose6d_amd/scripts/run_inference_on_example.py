"""The reference's inference script on the HIP engine, stand-alone (no reference install needed).

Same directory layout, functions and command line as src/megapose/scripts/run_inference_on_example.py:
    <MEGAPOSE_DATA_DIR>/examples/<name>/{image_rgb.png, [image_depth.png,] camera_data.json, inputs/object_data.json, meshes/<label>/*.{obj,ply}}
    python -m megapose6d_amd.scripts.run_inference_on_example <name> --run-inference [--model megapose-1.0-RGB-multi-hypothesis]
        -> outputs/object_data.json   ([{"label": ..., "TWO": [[qx, qy, qz, qw], [x, y, z]]}, ...], the reference's format)
    ... --vis-detections / --vis-outputs  -> visualizations/{detections,mesh_overlay,contour_overlay,all_results}.png
The visualisations are drawn with PIL from the engine's own rasteriser (the reference uses bokeh + a Panda3D scene render,
run_inference_on_example.py:94-106, 151-195); file names and contents (boxes; mesh overlay; contour overlay; the three side by side)
follow it.  (With the reference installed, its own script also runs unmodified on the engine through the class swap of INTEGRATION.md.)
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

from ..load_model import LOCAL_DATA_DIR, NAMED_MODELS, load_named_model
from ..scene_data import CameraData, ObjectData, Transform, make_detections_from_object_data
from ..synthetic import RigidObject, RigidObjectDataset
from ..types import ObservationTensor, Panda3dLightData


def load_observation(example_dir: Path, load_depth: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray], CameraData]:
    from PIL import Image

    camera_data = CameraData.from_json((example_dir / "camera_data.json").read_text())
    rgb = np.array(Image.open(example_dir / "image_rgb.png").convert("RGB"), dtype=np.uint8)
    assert rgb.shape[:2] == camera_data.resolution
    depth = None
    if load_depth:
        depth = np.array(Image.open(example_dir / "image_depth.png"), dtype=np.float32) / 1000   # millimetres, as the reference's examples
        assert depth.shape[:2] == camera_data.resolution
    return rgb, depth, camera_data


def load_observation_tensor(example_dir: Path, load_depth: bool = False) -> ObservationTensor:
    rgb, depth, camera_data = load_observation(example_dir, load_depth)
    return ObservationTensor.from_numpy(rgb, depth, camera_data.K)


def load_object_data(data_path: Path) -> List[ObjectData]:
    return [ObjectData.from_json(d) for d in json.loads(data_path.read_text())]


def load_detections(example_dir: Path):
    return make_detections_from_object_data(load_object_data(example_dir / "inputs/object_data.json")).cuda()


def make_object_dataset(example_dir: Path) -> RigidObjectDataset:
    rigid_objects = []
    mesh_units = "mm"
    for object_dir in sorted((example_dir / "meshes").iterdir()):
        label = object_dir.name
        mesh_path = None
        for fn in sorted(object_dir.glob("*")):
            if fn.suffix in {".obj", ".ply"}:
                assert not mesh_path, f"there multiple meshes in the {label} directory"
                mesh_path = fn
        assert mesh_path, f"couldnt find a obj or ply mesh for {label}"
        rigid_objects.append(RigidObject(label=label, mesh_path=mesh_path, mesh_units=mesh_units))
    return RigidObjectDataset(rigid_objects)


def save_predictions(example_dir: Path, pose_estimates) -> None:
    labels = pose_estimates.infos["label"]
    poses = pose_estimates.poses.cpu().numpy()
    object_data = [ObjectData(label=label, TWO=Transform(pose)) for label, pose in zip(labels, poses)]
    output_fn = example_dir / "outputs" / "object_data.json"
    output_fn.parent.mkdir(exist_ok=True)
    output_fn.write_text(json.dumps([x.to_json() for x in object_data]))
    print(f"Wrote predictions: {output_fn}")


def run_inference(example_dir: Path, model_name: str) -> None:
    model_info = NAMED_MODELS[model_name]
    observation = load_observation_tensor(example_dir, load_depth=model_info["requires_depth"]).cuda()
    detections = load_detections(example_dir).cuda()
    object_dataset = make_object_dataset(example_dir)
    print(f"Loading model {model_name}.")
    pose_estimator = load_named_model(model_name, object_dataset).cuda()
    print("Running inference.")
    output, _ = pose_estimator.run_inference_pipeline(observation, detections=detections, **model_info["inference_parameters"])
    save_predictions(example_dir, output)


# -- visualisations (PIL; the reference draws the same figures with bokeh) -------------------------------------------------------
def make_detections_visualization(example_dir: Path) -> None:
    from PIL import Image, ImageDraw

    rgb, _, _ = load_observation(example_dir, load_depth=False)
    im = Image.fromarray(rgb)
    draw = ImageDraw.Draw(im)
    for d in load_object_data(example_dir / "inputs/object_data.json"):
        x1, y1, x2, y2 = (float(v) for v in d.bbox_modal)
        draw.rectangle([x1, y1, x2, y2], outline=(255, 0, 0), width=2)
        draw.text((x1 + 3, y1 + 3), d.label, fill=(255, 0, 0))
    output_fn = example_dir / "visualizations" / "detections.png"
    output_fn.parent.mkdir(exist_ok=True)
    im.save(output_fn)
    print(f"Wrote detections visualization: {output_fn}")


def contour_overlay(rgb: np.ndarray, rendered_rgb: np.ndarray, mask: np.ndarray, color=(0, 255, 0), dilate_iterations: int = 1) -> np.ndarray:
    """green outline of the rendered silhouette over the image (visualization/utils.py make_contour_overlay: Canny edges of the
    render, dilated once; here: the mask's boundary pixels, dilated the same number of times)"""
    m = mask.astype(bool)
    inner = m.copy()
    inner[1:] &= m[:-1]; inner[:-1] &= m[1:]; inner[:, 1:] &= m[:, :-1]; inner[:, :-1] &= m[:, 1:]
    edge = m & ~inner
    for _ in range(dilate_iterations):
        e = edge.copy()
        e[1:] |= edge[:-1]; e[:-1] |= edge[1:]; e[:, 1:] |= edge[:, :-1]; e[:, :-1] |= edge[:, 1:]
        edge = e
    out = rgb.copy()
    out[edge] = np.asarray(color, np.uint8)
    return out


def render_predictions(example_dir: Path) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (rgb uint8 [H,W,3], rendered rgb uint8 [H,W,3], mask [H,W]) of outputs/object_data.json under the example's camera"""
    import torch

    from ..renderer import Panda3dBatchRenderer

    rgb, _, camera_data = load_observation(example_dir, load_depth=False)
    object_datas = load_object_data(example_dir / "outputs" / "object_data.json")
    renderer = Panda3dBatchRenderer(make_object_dataset(example_dir), n_workers=1)
    n = len(object_datas)
    TCO = torch.from_numpy(np.stack([d.TWO.matrix for d in object_datas]).astype(np.float32)).cuda()   # camera = world (TWC = I)
    K = torch.from_numpy(np.repeat(camera_data.K[None].astype(np.float32), n, 0)).cuda()
    out = renderer.render([d.label for d in object_datas], TCO, K, [[Panda3dLightData("ambient", (1.0, 1.0, 1.0, 1.0))]] * n,
                          camera_data.resolution, render_depth=True)
    depth = out.depths[:, 0]
    zbuf = torch.full_like(depth[0], float("inf"))
    img = torch.zeros(3, *depth.shape[1:], device=depth.device)
    for i in range(n):   # nearest surface wins
        closer = (depth[i] > 0) & (depth[i] < zbuf)
        img = torch.where(closer[None], out.rgbs[i], img)
        zbuf = torch.where(closer, depth[i], zbuf)
    mask = torch.isfinite(zbuf).cpu().numpy()
    rendered = (img.permute(1, 2, 0).cpu().numpy() * 255).round().astype(np.uint8)
    return rgb, rendered, mask


def make_output_visualization(example_dir: Path) -> None:
    from PIL import Image

    rgb, rendered, mask = render_predictions(example_dir)
    overlay = rgb.copy()
    overlay[mask] = (0.4 * rgb[mask] + 0.6 * rendered[mask]).astype(np.uint8)   # BokehPlotter.plot_overlay: the render blended over the image
    contour = contour_overlay(rgb, rendered, mask, color=(0, 255, 0), dilate_iterations=1)
    vis_dir = example_dir / "visualizations"
    vis_dir.mkdir(exist_ok=True)
    Image.fromarray(overlay).save(vis_dir / "mesh_overlay.png")
    Image.fromarray(contour).save(vis_dir / "contour_overlay.png")
    Image.fromarray(np.concatenate([rgb, contour, overlay], axis=1)).save(vis_dir / "all_results.png")
    print(f"Wrote visualizations to {vis_dir}.")


def main(argv=None) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("example_name")
    parser.add_argument("--model", type=str, default="megapose-1.0-RGB-multi-hypothesis")
    parser.add_argument("--vis-detections", action="store_true")
    parser.add_argument("--run-inference", action="store_true")
    parser.add_argument("--vis-outputs", action="store_true")
    args = parser.parse_args(argv)
    example_dir = LOCAL_DATA_DIR / "examples" / args.example_name
    if args.vis_detections:
        make_detections_visualization(example_dir)
    if args.run_inference:
        run_inference(example_dir, args.model)
    if args.vis_outputs:
        make_output_visualization(example_dir)


if __name__ == "__main__":
    main()
