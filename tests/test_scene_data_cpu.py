"""CPU: the annotation types and file handling of the inference caller (megapose6d_amd.scene_data, scripts/run_inference_on_example):
JSON layouts of the reference (datasets/scene_dataset.py:67-165), Transform algebra and its Eigen-style quaternion extraction, the
example directory -> observation / detections / object dataset path, and the written predictions."""
import json

import numpy as np
import pandas as pd
import pytest
import torch


def _rand_rot(rng):
    q = rng.randn(4)
    q /= np.linalg.norm(q)
    from megapose6d_amd.scene_data import quaternion_xyzw_to_matrix

    return quaternion_xyzw_to_matrix(q)


def test_transform_algebra_and_quaternion_extraction():
    from megapose6d_amd.scene_data import Transform, matrix_to_quaternion_xyzw, quaternion_xyzw_to_matrix

    rng = np.random.RandomState(0)
    for i in range(200):
        R = _rand_rot(rng)
        if i % 4 == 0:   # force the negative-trace branches: rotations by ~pi about an axis
            ax = rng.randn(3); ax /= np.linalg.norm(ax)
            R = 2 * np.outer(ax, ax) - np.eye(3) + 1e-3 * 0   # rotation by pi about ax (trace = -1)
        q = matrix_to_quaternion_xyzw(R)
        assert abs(np.linalg.norm(q) - 1) < 1e-12 and np.abs(quaternion_xyzw_to_matrix(q) - R).max() < 1e-12
        if np.trace(R) > 0:
            assert q[3] > 0   # Eigen's trace branch returns w > 0
    A, B = Transform(_rand_rot(rng), rng.randn(3)), Transform(_rand_rot(rng), rng.randn(3))
    assert np.abs((A * B).matrix - A.matrix @ B.matrix).max() < 1e-12
    assert np.abs((A * A.inverse()).matrix - np.eye(4)).max() < 1e-12
    assert np.abs(Transform(A.matrix).matrix - A.matrix).max() == 0 and np.abs(Transform(torch.from_numpy(A.matrix)).matrix - A.matrix).max() == 0
    T = Transform((0.0, 0.0, 2.0, 2.0), (1, 2, 3))   # un-normalised xyzw quaternion: normalised on the way in (90 degrees about z)
    assert np.abs(T.matrix[:3, :3] - np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])).max() < 1e-12 and T.translation.tolist() == [1, 2, 3]
    assert T.toHomogeneousMatrix().shape == (4, 4)
    with pytest.raises(ValueError):
        Transform(np.eye(3))
    # the reference README's known answer (README.md:259) survives a round trip through the matrix form
    q = np.array([0.5453961536730983, 0.6226545207599095, -0.43295293693197473, 0.35692612413663855])
    back = Transform(tuple(q), (0.107, 0.073, 0.457)).quaternion.coeffs()
    assert np.abs(back - q / np.linalg.norm(q)).max() < 1e-12


def test_object_and_camera_json_round_trip():
    from megapose6d_amd.scene_data import CameraData, ObjectData, Transform, make_detections_from_object_data

    d = {"label": "barbecue-sauce", "bbox_modal": [384, 234, 522, 455], "visib_fract": 0.5, "unique_id": 3,
         "TWO": [[0.0, 0.0, 0.0, 1.0], [0.1, 0.2, 0.3]]}
    o = ObjectData.from_json(d)
    assert o.label == "barbecue-sauce" and o.bbox_modal.tolist() == [384, 234, 522, 455] and o.bbox_amodal is None
    j = o.to_json()
    assert j["label"] == d["label"] and j["bbox_modal"] == d["bbox_modal"] and j["visib_fract"] == 0.5 and j["unique_id"] == 3
    assert np.allclose(j["TWO"][0], [0, 0, 0, 1]) and np.allclose(j["TWO"][1], [0.1, 0.2, 0.3]) and "TWO_init" not in j
    cam = CameraData.from_json(json.dumps({"K": [[605.9, 0, 319.0], [0, 605.8, 249.9], [0, 0, 1]], "resolution": [480, 640], "camera_id": "c"}))
    assert cam.resolution == (480, 640) and cam.K.shape == (3, 3) and cam.TWC is None
    back = json.loads(cam.to_json())
    assert back["resolution"] == [480, 640] and back["camera_id"] == "c" and back["K"][0][0] == 605.9
    cam.TWC = Transform(np.eye(4))
    assert json.loads(cam.to_json())["TWC"] == [[0.0, 0.0, 0.0, 1.0], [0.0, 0.0, 0.0]]
    det = make_detections_from_object_data([o, ObjectData.from_json({"label": "b", "bbox_modal": [1, 2, 3, 4]})])
    assert det.infos["label"].tolist() == ["barbecue-sauce", "b"] and det.infos["batch_im_id"].tolist() == [0, 0]
    assert det.infos["instance_id"].tolist() == [0, 1] and det.bboxes.shape == (2, 4)


def test_example_directory_round_trip(tmp_path):
    """build an example directory in the reference's layout, read it back, write predictions and the detection figure"""
    from PIL import Image

    from megapose6d_amd import synthetic as syn
    from megapose6d_amd.scripts import run_inference_on_example as ex
    from megapose6d_amd.tcoll import PandasTensorCollection

    d = tmp_path / "examples" / "bottle"
    (d / "inputs").mkdir(parents=True)
    (d / "meshes" / "bottle").mkdir(parents=True)
    verts, faces, colors = syn.make_lathe_mesh(seed=1, n_theta=12, n_z=8)[:3]
    syn.write_ply(d / "meshes" / "bottle" / "mesh.ply", verts, faces, colors)
    rgb = (np.random.RandomState(0).rand(48, 64, 3) * 255).astype(np.uint8)
    Image.fromarray(rgb).save(d / "image_rgb.png")
    Image.fromarray((np.full((48, 64), 750)).astype(np.uint16)).save(d / "image_depth.png")   # millimetres
    (d / "camera_data.json").write_text(json.dumps({"K": [[60.0, 0, 32.0], [0, 60.0, 24.0], [0, 0, 1]], "resolution": [48, 64]}))
    (d / "inputs" / "object_data.json").write_text(json.dumps([{"label": "bottle", "bbox_modal": [10, 8, 40, 44]}]))
    obs = ex.load_observation_tensor(d, load_depth=True)
    assert obs.images.shape == (1, 4, 48, 64) and abs(float(obs.images[0, 3, 0, 0]) - 0.75) < 1e-6 and obs.K.shape == (1, 3, 3)
    assert torch.equal((obs.images[0, :3] * 255).round().to(torch.uint8), torch.from_numpy(rgb).permute(2, 0, 1))
    assert ex.load_observation_tensor(d).images.shape == (1, 3, 48, 64)
    ds = ex.make_object_dataset(d)
    assert len(ds) == 1 and ds[0].label == "bottle" and ds[0].mesh_units == "mm" and abs(ds[0].scale - 1e-3) < 1e-12
    objs = ex.load_object_data(d / "inputs/object_data.json")
    assert objs[0].bbox_modal.tolist() == [10, 8, 40, 44]
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    T[:3, 3] = [0.01, -0.02, 0.6]
    ex.save_predictions(d, PandasTensorCollection(infos=pd.DataFrame(dict(label=["bottle"])), poses=torch.from_numpy(T)[None]))
    out = json.loads((d / "outputs" / "object_data.json").read_text())
    assert out[0]["label"] == "bottle" and np.allclose(out[0]["TWO"][0], [0, 0, np.sqrt(0.5), np.sqrt(0.5)]) and np.allclose(out[0]["TWO"][1], [0.01, -0.02, 0.6])
    ex.make_detections_visualization(d)
    im = np.array(Image.open(d / "visualizations" / "detections.png"))
    assert im.shape == (48, 64, 3) and (im[8, 10:41] == [255, 0, 0]).all()   # the top edge of the box
    c = ex.contour_overlay(rgb, rgb, np.pad(np.ones((10, 10), bool), ((5, 33), (5, 49))), dilate_iterations=0)
    assert (c[5, 5:15] == [0, 255, 0]).all() and (c[10, 10] == rgb[10, 10]).all()
