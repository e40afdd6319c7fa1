"""-m gpu: UV-textured objects -- HIP rasteriser vs the oracle's texture contract (oracle/raster.c), bit for bit,
and through the Panda3dBatchRenderer API."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def textured(tmp_path_factory):
    from megapose6d_amd import mesh_io
    from tests.support import synthetic as syn

    d = tmp_path_factory.mktemp("tex")
    objs = [syn.make_textured_object(d / "a", "tex_a", seed=3, fmt="obj", with_vertex_colors=True),
            syn.make_textured_object(d / "b", "tex_b", seed=4, fmt="ply_face"),
            syn.make_textured_object(d / "c", "tex_c", seed=5, fmt="ply_vertex")]
    plain = syn.make_object_dataset(d / "p", n_objects=1, seed=9).list_objects
    objs += plain                                                 # a vertex-colour object in the same database
    return objs, [mesh_io.load_rigid_object(o) for o in objs]


@pytest.mark.parametrize("msaa", [0, 16])
@pytest.mark.parametrize("res,zr", [((240, 320), (0.3, 0.6)), ((60, 80), (0.5, 0.9)), ((24, 32), (0.8, 1.2))])
def test_textured_raster_bit_exact_vs_oracle(textured, res, zr, msaa):
    """three resolutions/distances so that the per-pixel LOD spans mip levels 0..4 (trilinear); ambient light -> bit-identical"""
    from megapose6d_amd import engine as eng
    from tests.support import synthetic as syn
    from oracle import raster as orr

    _, meshes = textured
    db = eng.MeshDB(meshes)
    rng = np.random.RandomState(11)
    ids = np.array([0, 1, 2, 3, 0, 1], np.int32)
    n = len(ids)
    T = np.stack([syn.random_pose(rng, z_range=zr) for _ in range(n)])
    h, w = res
    K = np.repeat(syn.K_EXAMPLE[None].astype(np.float32), n, 0)
    K[:, :2] *= w / 640.0
    out = torch.full((n, h, w, 8), -1.0, device="cuda")
    eng.raster_render(db, torch.from_numpy(ids).cuda(), torch.from_numpy(T).cuda(), torch.from_numpy(K).cuda(), h, w, 3 | msaa, eng.make_lights(), out,
                      h * w * 8, w * 8, 8, 0, 3, 6)
    got = out.cpu().numpy()
    for i in range(n):
        rgb, nrm, dep = orr.render(meshes[ids[i]], T[i : i + 1], K[i : i + 1], h, w, 3 | msaa)
        assert (dep[0] > 0).sum() > 5
        assert np.array_equal(got[i, :, :, 0:3], rgb[0]) and np.array_equal(got[i, :, :, 3:6], nrm[0]) and np.array_equal(got[i, :, :, 6], dep[0])
    # the texture matters: rendering the same mesh without uvs differs
    m0 = dict(meshes[0])
    m0.pop("uvs")
    rgb_plain, _, _ = orr.render(m0, T[0:1], K[0:1], h, w, msaa)
    assert np.abs(got[0, :, :, 0:3] - rgb_plain[0]).max() > 0.2


def test_textured_objects_through_renderer_api_with_point_lights(textured):
    from tests.support import synthetic as syn
    from megapose6d_amd.renderer import Panda3dBatchRenderer
    from megapose6d_amd.types import make_scene_lights
    from oracle import raster as orr

    objs, meshes = textured
    r = Panda3dBatchRenderer(syn.RigidObjectDataset(objs), n_workers=1)
    rng = np.random.RandomState(2)
    labels = ["tex_b", "tex_a", objs[3].label]
    T = np.stack([syn.random_pose(rng, z_range=(0.3, 0.5)) for _ in labels]).astype(np.float32)
    K = np.repeat(syn.K_EXAMPLE[None].astype(np.float32), 3, 0)
    K[:, :2] *= 0.5
    lights = make_scene_lights()
    out = r.render(labels, torch.from_numpy(T).cuda(), torch.from_numpy(K).cuda(), [lights] * 3, (240, 320), render_normals=True)
    Lo = orr.lights_struct((0.1, 0.1, 0.1), orr.POINT_DIRS, [(0.4, 0.4, 0.4)] * 6)
    by_label = {o.label: m for o, m in zip(objs, meshes)}
    for i, lab in enumerate(labels):
        rgb, nrm, _ = orr.render(by_label[lab], T[i : i + 1], K[i : i + 1], 240, 320, 1 | 16, Lo)   # the renderer defaults to 4x MSAA
        d = np.abs(out.rgbs[i].permute(1, 2, 0).cpu().numpy() - rgb[0])
        assert d.max() <= 1.0 / 255 + 1e-7 and (d > 0).mean() < 1e-3   # sqrt/div chains of the point lights: 1 LSB on a few pixels
        assert np.array_equal(out.normals[i].permute(1, 2, 0).cpu().numpy(), nrm[0])
