"""CPU: host logic, the C-ABI surface (load + symbols, no compute), the rasteriser oracle's analytic properties,
and the world_size-2 gloo path of the row sharding."""
import ctypes
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_c_abi_library_exports_every_declared_symbol():
    from megapose6d_amd import _lib

    header = (ROOT / "include" / "mp_engine.h").read_text() + (ROOT / "include" / "mp_engine_debug.h").read_text()
    declared = set(re.findall(r"\b(mp_[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"mp_stream"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()  # raises if the .so is missing or a symbol does not resolve
    assert lib.mp_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mp_[A-Za-z0-9_]+)", out))
    assert declared <= exported


def test_integration_md_lists_exactly_the_exported_entry_points():
    """INTEGRATION.md's entry-point table is generated from include/mp_engine.h (scripts/gen_entry_points.py): it must be current, name
    every symbol the library exports and none that it does not (round 4's list had drifted: deleted entry points still listed, new ones
    missing)."""
    from megapose6d_amd import _lib

    sys.path.insert(0, str(ROOT / "scripts"))
    import gen_entry_points as gep

    text = (ROOT / "INTEGRATION.md").read_text()
    block = gep.current_block(text)
    assert block is not None, "entry-point markers missing from INTEGRATION.md"
    assert block == gep.render_block(gep.parse_header()), "stale: run python scripts/gen_entry_points.py"
    listed = set(re.findall(r"^\| `(mp_[A-Za-z0-9_]+)` \|", block, flags=re.M))
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mp_[A-Za-z0-9_]+)", out))
    assert listed == exported, (sorted(listed - exported), sorted(exported - listed))
    # and the prose around the table may only name entry points that exist (wildcards like mp_mesh_db_* aside)
    prose = text.replace(block, "")
    named = set(re.findall(r"`(mp_[A-Za-z0-9_]+)(?:\(|`)", prose)) - {"mp_stream", "mp_lights", "mp_conv_desc", "mp_mesh_desc", "mp_mesh_db", "mp_backbone", "mp_detector_config", "mp_detector"}
    unknown = {n for n in named if n not in exported and not any(e.startswith(n) for e in exported)}
    assert not unknown, sorted(unknown)


def test_one_rank_emulation_shards_like_a_rank_and_gathers_locally():
    """bench.py --emulate-rank-of N (a measurement rig): the process owns rows r::N like rank r of N, a gather returns a full-size table
    whose rows r::N are its own (the other ranks' rows are copies); switching it off restores the single-process behaviour."""
    from megapose6d_amd import distributed as mpd

    assert mpd.world_size() == 1 and mpd.rank() == 0 and not mpd.emulated()
    mpd.emulate(3, 8)
    try:
        assert mpd.emulated() and mpd.rank() == 3 and mpd.world_size() == 8
        n = 45   # ragged: ranks 0..4 own 6 rows, ranks 5..7 own 5
        idx = mpd.shard_indices(n, 3, 8)
        assert list(idx[:3]) == [3, 11, 19] and len(idx) == mpd.shard_size(n, 3, 8) == 6
        local = torch.arange(len(idx) * 2, dtype=torch.float32).view(len(idx), 2) + 100.0
        mpd.stats.reset()
        full = mpd.gather_rows(local, n, 3, 8)
        assert full.shape == (n, 2) and torch.equal(full[idx], local)
        assert mpd.stats.calls == 1 and mpd.stats.backend == "emulated"
        mpd.emulate(7, 8)   # a rank with the shorter shard: the padded tail must not leak zeros into its own rows
        idx7 = mpd.shard_indices(n, 7, 8)
        local7 = torch.ones(len(idx7), 3)
        full7 = mpd.gather_rows(local7, n, 7, 8)
        assert full7.shape == (n, 3) and torch.equal(full7[idx7], local7) and bool((full7 == 1).all())
    finally:
        mpd.emulate(None)
    assert mpd.world_size() == 1 and not mpd.emulated()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from megapose6d_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("MP_ENGINE_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EngineError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_conv_weight_packing_layout():
    """host-side packing: [nblk][chunk][BN][32] over the concatenated (kh)(kw,c) K axis, zero padded"""
    from megapose6d_amd import _lib

    lib = _lib.load()
    Cout, Cin, K, Cp = 64, 9, 7, 12
    w = np.random.RandomState(0).randn(Cout, Cin, K, K).astype(np.float32)
    scale = np.random.RandomState(1).rand(Cout).astype(np.float32) + 0.5
    n = lib.mp_conv_packed_floats(Cp, Cout, K, K)
    run = K * Cp
    n_chunks = -(-K * run // 32)
    assert n == 1 * n_chunks * 64 * 32
    out = np.empty(n, np.float32)
    _lib.check(lib.mp_conv_pack_weights(w.ctypes.data, Cout, Cin, K, K, Cp, scale.ctypes.data, out.ctypes.data))
    p = out.reshape(1, n_chunks, 64, 32)
    for (co, ci, kh, kw) in [(0, 0, 0, 0), (63, 8, 6, 6), (17, 3, 2, 5)]:
        kidx = kh * run + kw * Cp + ci  # K walks the concatenated (kw, c) runs of the kernel rows
        assert p[0, kidx // 32, co, kidx % 32] == np.float32(w[co, ci, kh, kw] * scale[co])
    assert p[0, 0, 0, 9] == 0 and p[0, n_chunks - 1, 0, 31] == 0  # padded channel / K tail


def test_tensor_collection_semantics():
    from megapose6d_amd.tcoll import PandasTensorCollection, concatenate

    c = PandasTensorCollection(pd.DataFrame({"label": ["a", "b", "c"], "s": [1.0, 3.0, 2.0]}, index=[5, 6, 7]), poses=torch.arange(3.0))
    assert c.infos.index.tolist() == [0, 1, 2]
    d = c[torch.tensor([2, 0])]
    assert d.poses.tolist() == [2.0, 0.0] and d.infos["label"].tolist() == ["c", "a"]
    assert len(concatenate([c, d])) == 5 and len(concatenate([])) == 0
    with pytest.raises(AttributeError):
        _ = c.nope


def test_filter_pose_estimates_topk_and_ties():
    from megapose6d_amd.pose_estimator import PoseEstimator
    from megapose6d_amd.tcoll import PandasTensorCollection

    df = pd.DataFrame(dict(batch_im_id=[0] * 6, label=["a"] * 3 + ["b"] * 3, instance_id=[0] * 6, s=[0.1, 0.9, 0.9, 0.3, 0.2, 0.5]))
    data = PandasTensorCollection(df, poses=torch.arange(6.0))
    out = PoseEstimator.filter_pose_estimates(None, data, top_K=2, filter_field="s")
    assert sorted(out.poses.tolist()) == [1.0, 2.0, 3.0, 5.0]
    top1 = PoseEstimator.filter_pose_estimates(None, data, top_K=1, filter_field="s")
    assert sorted(top1.poses.tolist()) == [1.0, 5.0]  # exact tie -> lowest row wins (stable sort)


def test_fast_topk_reproduces_the_pandas_group_by_row_for_row():
    """PoseEstimator.filter_pose_estimates keeps the rows of `sort_values(kind="stable").groupby(...).head(K)` (reference
    inference/pose_estimator.py:643-667 with a stable sort) through an integer group key instead of pandas' group-by: same rows, same order --
    ties, descending / ascending, shuffled row order, many groups, string labels; and it steps aside (None) for what the key cannot carry."""
    from megapose6d_amd.pose_estimator import _topk_rows_fast

    def ref(df, K, field, asc):
        return df.sort_values(field, ascending=asc, kind="stable").groupby(["batch_im_id", "label", "instance_id"]).head(K).index.to_numpy()

    rng = np.random.RandomState(0)
    for B, M, K in ((1, 576, 576), (1, 576, 5), (8, 576, 576), (64, 36, 5), (64, 5, 1), (3, 7, 100)):
        df = pd.DataFrame(dict(label=np.repeat([f"obj_{i % 5:06d}" for i in range(B)], M), batch_im_id=np.repeat(np.arange(B) // 8, M),
                               instance_id=np.repeat(np.arange(B), M), hypothesis_id=np.tile(np.arange(M), B),
                               s=rng.randn(B * M).astype(np.float32)))
        df.loc[::7, "s"] = 0.5   # exact ties
        for asc in (False, True):
            assert np.array_equal(_topk_rows_fast(df, K, "s", asc), ref(df, K, "s", asc)), (B, M, K, asc)
        sh = df.sample(frac=1.0, random_state=1).reset_index(drop=True)
        assert np.array_equal(_topk_rows_fast(sh, K, "s", False), ref(sh, K, "s", False))
    nan = df.copy()
    nan.loc[3, "s"] = np.nan
    assert _topk_rows_fast(nan, 2, "s", False) is None
    big = df.copy()
    big["instance_id"] = big["instance_id"] + (1 << 30)
    assert _topk_rows_fast(big, 2, "s", False) is None
    assert _topk_rows_fast(df.iloc[:0], 2, "s", False) is None


def test_config_back_compat_and_named_models():
    from megapose6d_amd import load_model as lm

    cfg = lm.check_update_config(dict(backbone_str="vanilla_resnet34", multiview_type="front_3views", n_views=4, render_normals=True))
    assert cfg.multiview_type == "TCO+front_3views" and cfg.n_rendered_views == 4 and cfg.predict_pose_update
    assert cfg.depth_normalization_type == "tCR_scale" and lm.n_inputs_from_cfg(cfg) == 27
    assert set(lm.NAMED_MODELS) == {"megapose-1.0-RGB", "megapose-1.0-RGBD", "megapose-1.0-RGB-multi-hypothesis", "megapose-1.0-RGB-multi-hypothesis-icp"}
    sd = {"backbone.backbone.conv1.weight": 1, "backbone.head.0.bias": 2, "pose_fc.bias": 3}
    assert set(lm.change_keys_of_older_models(sd)) == {"backbone.conv1.weight", "views_logits_head.bias", "pose_fc.bias"}


def test_mesh_io_roundtrip_and_formats(tmp_path):
    from megapose6d_amd import mesh_io
    from tests.support import synthetic as syn

    v, f, c = syn.make_lathe_mesh(3, n_theta=40, n_z=60)
    assert len(v) >= 2000
    syn.write_ply(tmp_path / "m.ply", v, f, c)
    m = mesh_io.read_ply(tmp_path / "m.ply")
    assert np.allclose(m["vertices"], v.astype(np.float32)) and np.array_equal(m["faces"], f)
    assert np.allclose(m["colors"], c / 255.0)
    with open(tmp_path / "m.obj", "w") as fh:
        for p in v[:4]:
            fh.write(f"v {p[0]} {p[1]} {p[2]}\n")
        fh.write("f 1 2 3 4\n")
    o = mesh_io.read_obj(tmp_path / "m.obj")
    assert o["faces"].tolist() == [[0, 1, 2], [0, 2, 3]]
    obj = syn.RigidObject("x", tmp_path / "m.ply", mesh_units="mm", ypr_offset_deg=(90.0, 0.0, 0.0))
    e = mesh_io.load_rigid_object(obj)
    assert np.allclose(e["points"], v.astype(np.float32) * 0.001, atol=1e-7)
    assert np.allclose(e["vertices"][:, 0], -e["points"][:, 1], atol=1e-6)  # heading 90 deg about z
    n = np.linalg.norm(e["normals"], axis=1)
    assert np.allclose(n, 1.0, atol=1e-5)


def test_oracle_rasteriser_analytic_properties():
    """a fronto-parallel quad: exact pixel footprint (top-left rule), metric depth, albedo passthrough, normal LUT"""
    from oracle import raster as orr

    v = np.array([[-0.1, -0.1, 0], [0.1, -0.1, 0], [0.1, 0.1, 0], [-0.1, 0.1, 0]], np.float32)
    mesh = {"vertices": v, "normals": np.tile(np.array([[0, 0, -1.0]], np.float32), (4, 1)),
            "colors": np.tile(np.array([[0.2, 0.4, 0.6]], np.float32), (4, 1)), "faces": np.array([[0, 1, 2], [0, 2, 3]], np.int32)}
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 0.5
    K = np.array([[100, 0, 32], [0, 100, 24], [0, 0, 1]], np.float32)
    rgb, nrm, dep = orr.render(mesh, T[None], K[None], 48, 64, 3)
    m = dep[0] > 0
    # quad spans u in [12, 52], v in [4, 44]: pixel centres x+.5 in [12,52) -> x = 12..51 (right/bottom edges excluded)
    ys, xs = np.nonzero(m)
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (12, 51, 4, 43) and m.sum() == 40 * 40
    assert np.allclose(dep[0][m], 0.5, atol=1e-6)
    assert np.allclose(rgb[0][m], np.round(np.array([0.2, 0.4, 0.6]) * 255) / 255, atol=1e-7)
    # normal (0,0,-1) in camera frame -> Panda view (x, z, -y) = (0,-1,0): frac -> (0, 0, 0) -> LUT texel blend of 0 and 247
    assert nrm[0][m].min() >= 0 and nrm[0][m].max() <= 1
    # invalid pose -> zeros
    Tb = T.copy(); Tb[0, 0] = np.inf
    r2, _, d2 = orr.render(mesh, Tb[None], K[None], 48, 64, 3)
    assert r2.max() == 0 and d2.max() == 0
    # beyond far plane -> nothing; two-sided: flipped winding renders the same
    Tf = T.copy(); Tf[2, 3] = 10.5
    assert orr.render(mesh, Tf[None], K[None], 48, 64, 2)[2].max() == 0
    mesh2 = dict(mesh, faces=mesh["faces"][:, ::-1].copy())
    d3 = orr.render(mesh2, T[None], K[None], 48, 64, 3)[2]
    assert np.array_equal(d3 > 0, dep > 0) and np.allclose(d3, dep, atol=1e-6)


def _gloo_worker(rank, world, port, n, k, q):
    import torch.distributed as dist

    from megapose6d_amd import distributed as mpd

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(n * k, dtype=torch.float32).view(n, k)
    mine = torch.as_tensor(mpd.shard_indices(n, rank, world))
    out = mpd.gather_rows(full[mine], n, rank, world)
    q.put((rank, bool(torch.equal(out, full)), len(mine)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 576, 1])
def test_row_sharding_and_gather_gloo_world2(n):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n, 17, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert sum(c for _, _, c in res) == n and abs(res[0][2] - res[1][2]) <= 1


def test_textured_mesh_ingestion(tmp_path):
    """UV-textured assets: OBJ+MTL+PNG, PLY per-face texcoord (BOP layout), PLY per-vertex s,t -> per-corner uvs +
    RGBA8 mip chain; vertex order untouched"""
    from megapose6d_amd import mesh_io
    from tests.support import synthetic as syn

    ms = {fmt: mesh_io.load_rigid_object(syn.make_textured_object(tmp_path / fmt, seed=3, fmt=fmt)) for fmt in ("obj", "ply_face", "ply_vertex")}
    v, f, _ = syn.make_lathe_mesh(3, n_theta=48, n_z=40, height_mm=140.0, radius_mm=35.0)
    for m in ms.values():
        assert m["uvs"].shape == (len(f), 3, 2) and m["uvs"].dtype == np.float32
        assert np.array_equal(m["faces"], f) and np.allclose(m["points"], v * 1e-3, atol=1e-6)
        mips = m["texture_mips"]
        assert [l.shape for l in mips] == [(128 >> k if 128 >> k else 1, 256 >> k if 256 >> k else 1) for k in range(9)]
        assert all(l.dtype == np.uint32 and (l >> 24 == 255).all() for l in mips)
    assert np.array_equal(ms["obj"]["uvs"], ms["ply_face"]["uvs"])
    # image rows are flipped so that v grows with the row index: texel row 0 = bottom row of the picture
    img = syn.make_texture_image(3)
    lvl0 = ms["obj"]["texture_mips"][0]
    assert np.array_equal(lvl0 & 255, img[::-1, :, 0]) and np.array_equal((lvl0 >> 16) & 255, img[::-1, :, 2])
    # mip level 1 = rounded 2x2 means
    r0 = img[::-1, :, 0].astype(np.uint32)
    want = (r0[0::2, 0::2] + r0[1::2, 0::2] + r0[0::2, 1::2] + r0[1::2, 1::2] + 2) >> 2
    assert np.array_equal(ms["obj"]["texture_mips"][1] & 255, want)
    # odd sizes clamp at the edge and end in a single texel
    odd = mesh_io.build_mip_chain((np.arange(37 * 50 * 3) % 251).astype(np.uint8).reshape(37, 50, 3))
    assert [l.shape for l in odd] == [(37, 50), (18, 25), (9, 12), (4, 6), (2, 3), (1, 1)]
    # a missing texture file is an error, an OBJ without vt stays untextured
    (tmp_path / "obj" / "tex_000000.png").unlink()
    with pytest.raises(FileNotFoundError):
        mesh_io.load_rigid_object(syn.RigidObject("x", tmp_path / "obj" / "tex_000000.obj", mesh_units="mm"))


def test_oracle_textured_quad_known_answer():
    """oracle/raster.c texture contract: a fronto-parallel quad mapped 1:1 onto a 2-colour texture reproduces the texels at level 0,
    and their mean when minified"""
    from megapose6d_amd import mesh_io
    from oracle import raster as orr

    tex = np.zeros((64, 64, 3), np.uint8)
    tex[:, :32] = (255, 0, 0)
    tex[:, 32:] = (0, 0, 255)
    s = 0.1
    mesh = {"vertices": np.array([[-s, -s, 0], [s, -s, 0], [s, s, 0], [-s, s, 0]], np.float32), "normals": np.tile([[0, 0, -1.0]], (4, 1)).astype(np.float32),
            "colors": np.ones((4, 3), np.float32), "faces": np.array([[0, 1, 2], [0, 2, 3]], np.int32),
            "uvs": np.array([[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]], np.float32), "texture_mips": mesh_io.build_mip_chain(tex)}
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 1.0
    K = np.array([[320, 0, 32], [0, 320, 32], [0, 0, 1]], np.float32)   # quad = 64 x 64 px: one texel per pixel
    rgb, _, _ = orr.render(mesh, T[None], K[None], 64, 64, 0)
    assert np.array_equal(rgb[0, 5:60, 2:30], np.broadcast_to(np.float32([1, 0, 0]), (55, 28, 3)))
    assert np.array_equal(rgb[0, 5:60, 34:62], np.broadcast_to(np.float32([0, 0, 1]), (55, 28, 3)))
    K2 = np.array([[5, 0, 1], [0, 5, 1], [0, 0, 1]], np.float32)          # quad = 1 x 1 px -> coarsest levels: purple
    rgb2, _, _ = orr.render(mesh, T[None], K2[None], 2, 2, 0)
    px = rgb2[0][rgb2[0].sum(-1) > 0]
    assert len(px) >= 1 and (np.abs(px[:, 0] - 0.5) < 0.02).all() and (np.abs(px[:, 2] - 0.5) < 0.02).all()


class _FakeEstimator:
    """Duck type of PoseEstimator.run_inference_pipeline for the PredictionRunner host-logic tests: the 'pose' of a detection encodes
    (frame mean colour, bbox) so that mixing up frames / rows would show."""

    def run_inference_pipeline(self, observation, detections=None, run_detector=False, coarse_estimates=None, n_refiner_iterations=5,
                               n_pose_hypotheses=1, run_depth_refiner=False, bsz_images=None, bsz_objects=None):
        from megapose6d_amd.tcoll import PandasTensorCollection

        infos = detections.infos.copy().reset_index(drop=True)
        infos["instance_id"] = infos.groupby(["batch_im_id", "label"]).cumcount()
        im = torch.as_tensor(infos["batch_im_id"].values)
        poses = torch.eye(4).repeat(len(infos), 1, 1)
        poses[:, 0, 3] = observation.images[im, :3].mean(dim=(1, 2, 3))
        poses[:, 1, 3] = detections.bboxes[:, 0]
        poses[:, 2, 3] = observation.K[im, 0, 0]
        final = PandasTensorCollection(infos, poses=poses)
        coarse = PandasTensorCollection(infos.loc[infos.index.repeat(2)].reset_index(drop=True), poses=poses.repeat_interleave(2, 0))
        extra = {"refiner": {"preds": final}, "coarse": {"preds": coarse}, "depth_refiner": {"preds": final}}
        return final, extra


def _fake_scene_ds(n_frames=5):
    from megapose6d_amd.tcoll import PandasTensorCollection

    rng = np.random.RandomState(0)
    ds = []
    for i in range(n_frames):
        n_det = 1 + i % 3
        infos = pd.DataFrame(dict(label=[f"obj_{j:06d}" for j in range(n_det)], scene_id=48 + i // 3, view_id=100 + i))
        ds.append(dict(rgb=rng.randint(0, 255, (12, 16, 3)).astype(np.uint8), depth=rng.rand(12, 16).astype(np.float32),
                       K=np.diag([50.0 + i, 50.0, 1.0]), gt_detections=PandasTensorCollection(infos, bboxes=torch.from_numpy(rng.rand(n_det, 4).astype(np.float32)) + i)))
    return ds


def test_prediction_runner_batching_keeps_rows_and_frame_ids():
    """evaluation caller (reference evaluation/prediction_runner.py:79-209): any batch size gives the same rows, each carrying the
    scene_id / view_id of ITS frame; keys as in the reference"""
    from megapose6d_amd.prediction_runner import PredictionRunner
    from megapose6d_amd.types import InferenceConfig

    ds = _fake_scene_ds()
    cfg = InferenceConfig(detection_type="gt", n_refiner_iterations=3, run_depth_refiner=True)
    ref = PredictionRunner(ds, cfg, batch_size=1, device="cpu").get_predictions(_FakeEstimator())
    assert set(ref) == {"final", "refiner/iteration=3", "refiner/final", "coarse", "depth_refiner"}
    assert len(ref["final"]) == sum(1 + i % 3 for i in range(5)) and len(ref["coarse"]) == 2 * len(ref["final"])
    assert ref["final"].infos["view_id"].tolist() == [100 + i for i in range(5) for _ in range(1 + i % 3)]
    assert ref["final"].infos["scene_id"].tolist() == [48 + i // 3 for i in range(5) for _ in range(1 + i % 3)]
    for bs in (2, 5):
        got = PredictionRunner(ds, cfg, batch_size=bs, device="cpu").get_predictions(_FakeEstimator())
        for k in ref:
            assert torch.equal(got[k].poses, ref[k].poses)
            assert got[k].infos[["label", "scene_id", "view_id"]].equals(ref[k].infos[["label", "scene_id", "view_id"]])
    with pytest.raises(ValueError):
        PredictionRunner(ds, InferenceConfig(detection_type="nope"), device="cpu").get_predictions(_FakeEstimator())


def _runner_worker(rank, world, port, q):
    import torch.distributed as dist

    from megapose6d_amd.prediction_runner import PredictionRunner
    from megapose6d_amd.types import InferenceConfig

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    runner = PredictionRunner(_fake_scene_ds(), InferenceConfig(detection_type="gt"), batch_size=2, device="cpu")
    preds = runner.get_predictions(_FakeEstimator())
    q.put((rank, runner.frame_ids, sorted(preds["final"].infos["view_id"].tolist()), preds["final"].poses.sum().item()))
    dist.destroy_process_group()


def test_prediction_runner_two_ranks_gloo_gather():
    """frames dealt rank::world, results exchanged with one all_gather_object: every rank ends with all rows"""
    import torch.multiprocessing as mp

    from megapose6d_amd.prediction_runner import PredictionRunner
    from megapose6d_amd.types import InferenceConfig

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    single = PredictionRunner(_fake_scene_ds(), InferenceConfig(detection_type="gt"), batch_size=2, device="cpu").get_predictions(_FakeEstimator())
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    want = sorted(single["final"].infos["view_id"].tolist())
    assert res[0][2] == want and res[1][2] == want
    assert abs(res[0][3] - single["final"].poses.sum().item()) < 1e-4 and abs(res[1][3] - res[0][3]) < 1e-6


def _plan(N, H, W, C, Cout, K, stride, ws_floats=12 << 20, n_cu=256):
    import ctypes as C_

    from megapose6d_amd import _lib

    lib = _lib.load()
    d = _lib.ConvDesc()
    dummy = 0x1000  # pointers are only tested for NULL by the planner
    d.d_x, d.N, d.H, d.W, d.C, d.in_border = dummy, N, H, W, C, K // 2
    d.d_w, d.Cout, d.KH, d.KW, d.stride, d.pad = dummy, Cout, K, K, stride, K // 2
    d.d_y, d.out_border = dummy, 1
    if ws_floats:
        d.d_splitk_ws, d.splitk_ws_floats = dummy, ws_floats
    out = (C_.c_int32 * 5)()
    assert lib.mp_conv2d_plan(C_.byref(d), n_cu, out) == 0, lib.mp_last_error()
    return tuple(out)


def test_conv_launch_plan_small_grids_and_half_empty_last_rounds():
    """host-side planner of mp_conv2d_nhwc (no GPU work): which launches split K, and how"""
    # layer 3 at 576 rows: 1350 x 2 = 2700 tiles on 512 resident workgroups = 5 rounds + 140 tiles -> tail split 3 ways
    mode, S, cps, n_main, m_begin = _plan(576, 15, 20, 256, 256, 3, 1)
    assert (mode, S, n_main, m_begin) == (2, 3, 2560, 1280 * 128) and cps == 24
    # layer 2 (5400 tiles = 10 rounds + 280: more than half a round) and layer 4 (1440 = 2 rounds + 416): single pass
    assert _plan(576, 30, 40, 128, 128, 3, 1)[0] == 0 and _plan(576, 8, 10, 512, 512, 3, 1)[0] == 0
    # layer 1 (21 600 tiles of 128 x 64 = 42 rounds + 96 tiles, 18 chunks): tail split 3 ways
    assert _plan(576, 60, 80, 64, 64, 3, 1)[:2] == (2, 3)
    # batch 1: layer 4 has 4 tiles -> every tile split (144 chunks / 4 = 36 ways at most, 128 wanted -> 36)
    mode, S, cps, n_main, _ = _plan(1, 8, 10, 512, 512, 3, 1)
    assert mode == 1 and n_main == 0 and S * cps >= 144 and S <= 36 and cps >= 4
    # no scratch -> never split; tiny K (1x1 conv with 2 chunks) -> never split
    assert _plan(1, 8, 10, 512, 512, 3, 1, ws_floats=0)[0] == 0
    assert _plan(1, 30, 40, 64, 128, 1, 2)[0] == 0
    # scratch too small for two partial copies -> single pass
    assert _plan(1, 8, 10, 512, 512, 3, 1, ws_floats=80 * 512)[0] == 0


def test_conv_launch_plan_half_precision_input_is_single_pass():
    """mp_conv_desc.x_f16 (the stems of the "fp16 renders" mode): one single-pass launch whatever the grid; the same stem in fp32 at
    batch 1 (150 tiles) splits K"""
    from megapose6d_amd import engine as eng

    kw = dict(N=1, H=240, W=320, Cp=28, in_border=3, Cout=64, K=7, stride=2, pad=3, n_cu=256, ws_floats=12 << 20)
    assert eng.conv2d_plan(**kw)["mode"] == 1
    assert eng.conv2d_plan(**kw, x_f16=True) == dict(mode=0, k_split=1, chunks_per_split=43, n_main=0, m_begin=0)   # ceil(7 * 196 / 32)


def test_load_cfg_reads_python_tagged_legacy_configs_without_instantiating_them(tmp_path):
    """reference inference/utils.py:71-75 reads config.yaml with yaml.UnsafeLoader; older runs are python-tagged object dumps"""
    import argparse
    import pathlib

    import yaml

    from megapose6d_amd.load_model import load_cfg

    ns = argparse.Namespace(backbone_str="resnet34", n_rendered_views=4, save_dir=pathlib.PosixPath("/a/b"), hw=(240, 320), opt=dict(lr=0.1))
    (tmp_path / "ns.yaml").write_text(yaml.dump(ns))
    c = load_cfg(tmp_path / "ns.yaml")
    assert (c.backbone_str, c.n_rendered_views, c.save_dir, c.hw, c.opt) == ("resnet34", 4, "/a/b", (240, 320), {"lr": 0.1})
    (tmp_path / "plain.yaml").write_text("backbone_str: vanilla_resnet34\nn_rendered_views: 1\n")
    assert load_cfg(tmp_path / "plain.yaml").backbone_str == "vanilla_resnet34"
    # a tag that would execute code under UnsafeLoader is mapped to data, never called
    (tmp_path / "evil.yaml").write_text("a: !!python/object/apply:os.system ['echo pwned > /tmp/mp_pwned']\n")
    import os

    if os.path.exists("/tmp/mp_pwned"):
        os.remove("/tmp/mp_pwned")
    c = load_cfg(tmp_path / "evil.yaml")
    assert not os.path.exists("/tmp/mp_pwned") and c.a == "echo pwned > /tmp/mp_pwned"
    (tmp_path / "bad.yaml").write_text("- 1\n- 2\n")
    with pytest.raises(ValueError):
        load_cfg(tmp_path / "bad.yaml")


def test_detector_wrapper_builds_the_detections_collection():
    """Detector.get_detections (reference inference/detector.py:63-136) on a stand-in model with torchvision's Mask R-CNN output
    format; the same cases are compared with the imported reference class in tests/_ref_api_check.py (build container)."""
    from types import SimpleNamespace

    from megapose6d_amd.detector import Detector
    from megapose6d_amd.types import ObservationTensor

    class Fake(torch.nn.Module):
        config = SimpleNamespace(label_to_category_id={"a": 1, "b": 2})

        def forward(self, images):
            assert isinstance(images, list) and images[0].shape == (3, 6, 8)
            return [dict(boxes=torch.tensor([[1.0, 2, 5, 6], [0, 0, 3, 3], [2, 2, 4, 4]]), labels=torch.tensor([2, 1, 2]),
                         scores=torch.tensor([0.9, 0.2, 0.6]), masks=torch.full((3, 1, 6, 8), 0.85)),
                    dict(boxes=torch.zeros(0, 4), labels=torch.zeros(0, dtype=torch.long), scores=torch.zeros(0), masks=torch.zeros(0, 1, 6, 8))]

    det = Detector(Fake())
    obs = ObservationTensor(images=torch.rand(2, 4, 6, 8))   # RGBD frames: only the first three channels reach the model
    d = det.get_detections(obs, output_masks=True)
    assert d.infos["label"].tolist() == ["b", "a", "b"] and d.infos["batch_im_id"].tolist() == [0, 0, 0]
    assert d.infos["instance_id"].tolist() == [0, 0, 1] and d.bboxes.shape == (3, 4) and d.masks.shape == (3, 6, 8) and d.masks.all()
    assert len(det(obs, detection_th=0.5)) == 2
    one = det.get_detections(obs, one_instance_per_class=True)
    assert sorted(zip(one.infos["label"], one.infos["score"].round(3))) == [("a", 0.2), ("b", 0.9)]
    assert not det.get_detections(obs, output_masks=True, mask_th=0.9).masks.any()
    assert det.image_tensor_from_numpy(np.zeros((6, 8, 3), np.uint8)).shape == (3, 6, 8)


def test_detector_wrapper_accepts_models_that_skip_the_mask_head():
    """DetectorMaskRCNN(compute_masks=False) returns no "masks": fine unless the caller asks for them"""
    from types import SimpleNamespace

    from megapose6d_amd.detector import Detector
    from megapose6d_amd.types import ObservationTensor

    class NoMasks(torch.nn.Module):
        config = SimpleNamespace(label_to_category_id={"a": 1})

        def forward(self, images):
            return [dict(boxes=torch.tensor([[1.0, 2, 5, 6]]), labels=torch.tensor([1]), scores=torch.tensor([0.9]))]

    det = Detector(NoMasks())
    obs = ObservationTensor(images=torch.rand(1, 3, 6, 8))
    d = det.get_detections(obs)
    assert len(d) == 1 and "masks" not in d.tensors and d.infos["label"].tolist() == ["a"]
    with pytest.raises(ValueError):
        det.get_detections(obs, output_masks=True)
