"""-m gpu: the exact-piece bf16 stem path (csrc/conv_stem.hip + MP_RASTER_XREC + mp_backbone_forward_xrec).

What it replaces: the first convolution behind `self.backbone(x)` (reference src/megapose/models/torchvision_resnet.py:213-216,
models/wide_resnet.py:65-67) fed by the renders `uint8 / 255` of panda3d_batch_renderer.py:261-274.  Statements checked here:
  * the rasteriser's record output holds EXACTLY the values of its fp32 output (k with k / 255 == the fp32 value, x1 + x2 + x3 == the
    fp32 crop value): bit-exact decode;
  * the stem convolution on records equals torch's fp32 conv2d on the decoded tensor to fp32 round-off (1e-5 of the output scale: the
    products are exact, only the order of the fp32 additions differs), for 7x7 and 5x5 stems, ragged tiles, every record length;
  * a whole backbone forward / a whole pose step on records equals the fp32-tensor path to 1e-5 of the feature scale.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from megapose6d_amd import engine

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return engine


def split3(v: torch.Tensor):
    """exact truncation split of fp32 values into three bf16-representable fp32 pieces"""
    def trunc(t):
        return (t.view(torch.int32) & -65536).view(torch.float32)
    h = trunc(v)
    r = v - h
    m = trunc(r)
    lo = r - m
    assert torch.equal((h + m) + lo, v)
    return h, m, lo


def to_records(eng, x_nchw: torch.Tensor, n_f32: int, border: int, f32_mask=None) -> torch.Tensor:
    """fp32 [n, c, h, w] -> bf16 record tensor (padded NHWC).  The fp32-kind channels (three exact pieces each) are the first `n_f32`
    or, with `f32_mask`, the channels whose bit is set; every other channel must hold k / 255 and is stored as the integer k."""
    n, c, h, w = x_nchw.shape
    mask = (1 << n_f32) - 1 if f32_mask is None else f32_mask
    f_ch = [ch for ch in range(c) if (mask >> ch) & 1]
    u_ch = [ch for ch in range(c) if not (mask >> ch) & 1]
    nf = len(f_ch)
    R = eng.xrec_elements(nf, c - nf)
    buf = eng.padded_nhwc(n, h, w, R, border, "cuda", dtype=torch.bfloat16)
    v = eng.padded_view(buf, n, h, w, R, border)
    xl = x_nchw.permute(0, 2, 3, 1).cpu()
    k = torch.round(xl[..., u_ch] * 255.0)
    assert torch.equal(k / 255.0, xl[..., u_ch])   # (on the CPU: torch's GPU division by a scalar multiplies by the reciprocal)
    xl, k = xl.cuda(), k.cuda()
    pieces = torch.stack(split3(xl[..., f_ch].contiguous()), dim=-1).reshape(n, h, w, 3 * nf)
    v[..., : 3 * nf] = pieces.to(torch.bfloat16)
    v[..., 3 * nf : 3 * nf + len(u_ch)] = k.to(torch.bfloat16)
    assert torch.equal(v[..., : 3 * nf].float(), pieces)   # the pieces really are bf16 values
    return buf


RGBD_MASK = 0x8102040F   # 32-channel RGBD refiner: crop rgb + depth (0..3), rendered depth of the four views (10, 17, 24, 31)


STEM_CASES = [
    # N, n_f32, n_u8, H, W, K, Cout
    (2, 3, 6, 32, 48, 7, 64),      # coarse stem, record of 16
    (3, 3, 24, 30, 44, 7, 64),     # refiner stem, record of 40, ragged tiles (15 x 22 outputs)
    (2, 3, 24, 26, 38, 5, 128),    # WideResNet stem, two channel blocks
    (1, 3, 12, 24, 40, 7, 64),     # 2 views: record of 24
    (2, 3, 18, 18, 34, 5, 64),     # 3 views: record of 32
    (1, 4, 21, 17, 33, 7, 64),     # four fp32 channels (an RGBD-style crop without depth renders): record of 40
    (2, 8, 24, 19, 37, 7, 64),     # 48-element record (the RGBD refiner's size), 7x7: the patch only fits unpadded (2-way LDS conflicts)
    (1, 8, 24, 26, 22, 5, 128),    # 48-element record, WideResNet 5x5 stem: padded pixel pitch
    (2, 3, 15, 20, 30, 7, 64),     # record of 24 -> Q = 3 (odd: no padding)
    (1, 3, 21, 16, 34, 5, 64),     # record of 32 -> Q = 4 (padded pitch 5)
]


@pytest.mark.parametrize("case", STEM_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_stem_conv_on_records_matches_torch_fp32(eng, case, relu):
    N, nf, nu, H, W, K, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.cat([torch.randn(N, nf, H, W, generator=g) * 0.3 + 0.5,
                   torch.randint(0, 256, (N, nu, H, W), generator=g).float() / 255.0], dim=1)
    x[:, nf:, : H // 3] = 0.0   # background
    w = torch.randn(Cout, nf + nu, K, K, generator=g) * (2.0 / ((nf + nu) * K * K)) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    rec = to_records(eng, x, nf, pad)
    wp = torch.from_numpy(eng.conv_stem_pack_weights(w.numpy(), nf, scale.numpy())).cuda()
    yb = eng.padded_nhwc(N, Ho, Wo, Cout, 1, "cuda")
    eng.padded_view(yb, N, Ho, Wo, Cout, 1)[:] = 7.0   # poison: the interior must be fully overwritten, the border untouched
    eng.conv_stem_xrec(rec, N, H, W, nf + nu, nf, pad, wp, bias.cuda(), Cout, K, pad, yb, 1, relu=relu)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), (w * scale.view(-1, 1, 1, 1)).double(), bias.double(), stride=2, padding=pad)
    if relu:
        ref = F.relu(ref)
    got = eng.padded_view(yb, N, Ho, Wo, Cout, 1).permute(0, 3, 1, 2).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err
    full = yb[: N * (Ho + 2) * (Wo + 2) * Cout].view(N, Ho + 2, Wo + 2, Cout)
    assert full[:, 0].abs().max() == 0 and full[:, -1].abs().max() == 0 and full[:, :, 0].abs().max() == 0 and full[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("K", [7, 5])
def test_stem_conv_on_rgbd_records_with_scattered_fp32_channels(eng, K):
    """The RGBD refiner's input (training/pose_models_cfg.py:101-103: 32 channels = crop rgb + depth, 4 x (rgb, normals, depth)): the
    fp32-kind channels -- the crop and every depth channel, signed after tCR_scale_clamp_center -- are not the leading ones
    (mp_conv_stem_pack_weights_mask, mask 0x8102040F); 48-element records; vs torch's convolution in float64."""
    N, C, H, W, Cout = 2, 32, 30, 44, 64
    g = torch.Generator().manual_seed(40 + K)
    x = torch.randint(0, 256, (N, C, H, W), generator=g).float() / 255.0
    f_ch = [c for c in range(C) if (RGBD_MASK >> c) & 1]
    assert f_ch == [0, 1, 2, 3, 10, 17, 24, 31]
    x[:, f_ch] = torch.rand(N, len(f_ch), H, W, generator=g) * 2.0 - 1.0
    x[:, [c for c in range(C) if c not in f_ch], : H // 4] = 0.0
    w = torch.randn(Cout, C, K, K, generator=g) * (2.0 / (C * K * K)) ** 0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    rec = to_records(eng, x, 0, pad, f32_mask=RGBD_MASK)
    assert eng.xrec_elements(8, 24) == 48
    wp = torch.from_numpy(eng.conv_stem_pack_weights(w.numpy(), 0, scale.numpy(), f32_mask=RGBD_MASK)).cuda()
    yb = eng.padded_nhwc(N, Ho, Wo, Cout, 1, "cuda")
    eng.conv_stem_xrec(rec, N, H, W, C, 8, pad, wp, bias.cuda(), Cout, K, pad, yb, 1, relu=False)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), (w * scale.view(-1, 1, 1, 1)).double(), bias.double(), stride=2, padding=pad)
    got = eng.padded_view(yb, N, Ho, Wo, Cout, 1).permute(0, 3, 1, 2).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("case", [STEM_CASES[1], STEM_CASES[0], (2, 3, 24, 34, 70, 7, 64), (1, 3, 24, 240, 320, 7, 64),
                                  STEM_CASES[2], STEM_CASES[4], (1, 3, 24, 240, 320, 5, 64), (2, 8, 24, 34, 70, 5, 64)])   # + the WideResNets' 5x5 stem (round 6)
def test_fused_max_pool_equals_the_separate_pool_kernel_bit_for_bit(eng, case):
    """mp_conv_stem_xrec_pool (max pool taken from the tile in LDS, windows that straddle tiles combined with atomicMax) against
    mp_conv_stem_xrec + mp_maxpool3x3s2: the same fp32 maxima -> identical bits, whatever the order of the atomics; ragged tiles, odd
    output sizes, with and without the stem map written as well.  Reference: models/torchvision_resnet.py:213-216."""
    N, nf, nu, H, W, K, Cout = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.cat([torch.rand(N, nf, H, W, generator=g), torch.randint(0, 256, (N, nu, H, W), generator=g).float() / 255.0], dim=1)
    w = torch.randn(Cout, nf + nu, K, K, generator=g) * (2.0 / ((nf + nu) * K * K)) ** 0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    rec = to_records(eng, x, nf, pad)
    wp = torch.from_numpy(eng.conv_stem_pack_weights(w.numpy(), nf, scale.numpy())).cuda()
    y = eng.padded_nhwc(N, Ho, Wo, Cout, 1, "cuda")
    eng.conv_stem_xrec(rec, N, H, W, nf + nu, nf, pad, wp, bias.cuda(), Cout, K, pad, y, 1, relu=True)
    ref = eng.padded_nhwc(N, Hq, Wq, Cout, 1, "cuda")
    eng.maxpool3x3s2(y, N, Ho, Wo, Cout, 1, ref, 1)
    for with_map in (False, True):
        y2 = eng.padded_nhwc(N, Ho, Wo, Cout, 1, "cuda") if with_map else None
        got = eng.padded_nhwc(N, Hq, Wq, Cout, 1, "cuda")
        eng.padded_view(got, N, Hq, Wq, Cout, 1)[:] = 7.0   # stale values: every interior position must be cleared / overwritten
        eng.conv_stem_xrec(rec, N, H, W, nf + nu, nf, pad, wp, bias.cuda(), Cout, K, pad, y2, 1, relu=True, y_pool=got, pool_border=1)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        if with_map:
            assert torch.equal(y2, y)
    tref = F.max_pool2d(F.relu(F.conv2d(x, w * scale.view(-1, 1, 1, 1), bias, stride=2, padding=pad)), 3, 2, 1)
    assert (eng.padded_view(ref, N, Hq, Wq, Cout, 1).permute(0, 3, 1, 2).cpu() - tref).abs().max().item() < 1e-5 * max(1.0, tref.abs().max().item())
    # the pre-activation a WideResNet's first block reads (models/wide_resnet.py:29-44): second output of the separate pool kernel ==
    # mp_bn_relu_nhwc over the fused pool's map, bit for bit
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.2).cuda()
    act_ref = eng.padded_nhwc(N, Hq, Wq, Cout, 1, "cuda")
    eng.maxpool3x3s2(y, N, Ho, Wo, Cout, 1, None, 1, y_act=act_ref, sc=sc, sh=sh)
    act = eng.padded_nhwc(N, Hq, Wq, Cout, 1, "cuda")
    eng.bn_relu_nhwc(got, N, Hq, Wq, Cout, 1, act, sc, sh)
    torch.cuda.synchronize()
    assert torch.equal(act, act_ref) and act.abs().max() > 0


def test_stem_map_of_a_batch_beyond_4_gb_is_addressed_per_image(eng):
    """A coarse launch of `max_rows_per_launch` = 1000 rows (tests/test_gpu_full_size.py chunking case) has a 5 GB stem map: the kernel's
    32-bit buffer offsets are per image, the batch index is applied in 64 bits.  860 x 240 x 320 coarse records (4.3 GB map): the last
    and the first image equal the same images convolved alone, with and without the fused pool."""
    N, nf, nu, H, W, K, Cout = 860, 3, 6, 240, 320, 7, 64
    g = torch.Generator().manual_seed(5)
    x2 = torch.cat([torch.rand(2, nf, H, W, generator=g), torch.randint(0, 256, (2, nu, H, W), generator=g).float() / 255.0], dim=1)
    w = torch.randn(Cout, nf + nu, K, K, generator=g) * (2.0 / ((nf + nu) * K * K)) ** 0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, (torch.randn(Cout, generator=g) * 0.1).cuda()
    pad = K // 2
    Ho, Wo = H // 2, W // 2
    Hq, Wq = Ho // 2, Wo // 2
    R = eng.xrec_elements(nf, nu)
    rec2 = to_records(eng, x2, nf, pad)
    wp = torch.from_numpy(eng.conv_stem_pack_weights(w.numpy(), nf, scale.numpy())).cuda()
    y2, q2 = eng.padded_nhwc(2, Ho, Wo, Cout, 1, "cuda"), eng.padded_nhwc(2, Hq, Wq, Cout, 1, "cuda")
    eng.conv_stem_xrec(rec2, 2, H, W, nf + nu, nf, pad, wp, bias, Cout, K, pad, y2, 1, relu=True, y_pool=q2, pool_border=1)
    rec = eng.padded_nhwc(N, H, W, R, pad, "cuda", dtype=torch.bfloat16)
    per = (H + 2 * pad) * (W + 2 * pad) * R
    rec[: N * per].view(N, per)[:] = rec2[:per]          # image 0 everywhere ...
    rec[(N - 1) * per : N * per] = rec2[per : 2 * per]    # ... image 1 in the last slot
    y, q = eng.padded_nhwc(N, Ho, Wo, Cout, 1, "cuda"), eng.padded_nhwc(N, Hq, Wq, Cout, 1, "cuda")
    assert y.numel() * 4 > 2 ** 32
    eng.conv_stem_xrec(rec, N, H, W, nf + nu, nf, pad, wp, bias, Cout, K, pad, y, 1, relu=True, y_pool=q, pool_border=1)
    torch.cuda.synchronize()
    py, pq = (Ho + 2) * (Wo + 2) * Cout, (Hq + 2) * (Wq + 2) * Cout
    for n_big, n_small in ((0, 0), (N // 2, 0), (N - 1, 1)):
        assert torch.equal(y[n_big * py : (n_big + 1) * py], y2[n_small * py : (n_small + 1) * py]), n_big
        assert torch.equal(q[n_big * pq : (n_big + 1) * pq], q2[n_small * pq : (n_small + 1) * pq]), n_big
    assert eng.padded_view(y2, 2, Ho, Wo, Cout, 1).abs().max() > 0


@pytest.mark.parametrize("kind,c_in", [("vanilla_resnet34", 9), ("vanilla_resnet34", 27), ("resnet34", 27), ("resnet34", 32), ("vanilla_resnet34", 32)])
def test_backbone_forward_on_records_matches_the_fp32_tensor_path(eng, kind, c_in):
    """c_in = 32: the RGBD refiner's channel layout (fp32-kind channels = RGBD_MASK: crop rgb + depth, one rendered depth per view)"""
    from tests.support import synthetic as syn

    sd = syn.make_state_dict(kind, c_in, "pose", 9, seed=4)
    bb = eng.Backbone(kind, c_in, "pose", 9, sd)
    mask = RGBD_MASK if c_in == 32 else 0b111
    f_ch = [c for c in range(c_in) if (mask >> c) & 1]
    R = bb.xrec_elements(f32_mask=mask)
    assert R == eng.xrec_elements(len(f_ch), c_in - len(f_ch)) and R in (16, 40, 48)
    b, h, w = 3, 240, 320
    g = torch.Generator().manual_seed(c_in)
    x = torch.randint(0, 256, (b, c_in, h, w), generator=g).float() / 255.0
    x[:, f_ch] = torch.rand(b, len(f_ch), h, w, generator=g) * (2.0 if c_in == 32 else 1.0) - (1.0 if c_in == 32 else 0.0)
    xb = eng.padded_nhwc(b, h, w, bb.c_in_p, bb.in_border, "cuda")
    eng.padded_view(xb, b, h, w, bb.c_in_p, bb.in_border)[..., :c_in] = x.permute(0, 2, 3, 1).cuda()
    rec = to_records(eng, x, 0, bb.in_border, f32_mask=mask)
    nf = 512
    out0, out1 = torch.empty(b, 9, device="cuda"), torch.empty(b, 9, device="cuda")
    f0, f1 = torch.empty(b, nf, device="cuda"), torch.empty(b, nf, device="cuda")
    bb.forward(xb, b, h, w, out0, None, f0)
    bb.forward(rec, b, h, w, out1, None, f1, f32_mask=mask)
    torch.cuda.synchronize()
    fs = max(1.0, f0.abs().max().item())
    assert (f0 - f1).abs().max().item() < 1e-5 * fs, ((f0 - f1).abs().max().item(), fs)
    assert (out0 - out1).abs().max().item() < 1e-5 * max(1.0, out0.abs().max().item())


def test_rasteriser_record_output_holds_exactly_the_fp32_values(object_dataset):
    """one refiner step (4 views + the fused crop) staged as records vs as the fp32 tensor: decoding the records gives the fp32 CNN
    input bit for bit, and the network output agrees to fp32 round-off"""
    from megapose6d_amd.load_model import build_pose_model
    from megapose6d_amd.mesh_db import MeshDataBase
    from megapose6d_amd.renderer import Panda3dBatchRenderer
    from tests.support import synthetic as syn

    for role in ("refiner", "coarse", "coarse_no_normals", "refiner_rgbd", "refiner_rgbd_wide"):
        cfg = syn.make_cfg(role.split("_")[0], rgbd="rgbd" in role)
        backbone = "resnet34" if role.endswith("wide") else "vanilla_resnet34"
        if role == "coarse_no_normals":
            # the 6-channel input of the legacy / *-no_normals configs (utils/load_model.py check_update_config_pose defaults: one view,
            # render_normals False): 3 crop + 3 lit render channels = a 16-element record for a 6-float channel run
            cfg.render_normals = False
            assert syn.n_inputs_for(cfg) == 6
        role = role.split("_")[0]
        head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
        cfg.backbone_str = backbone
        sd = syn.make_state_dict(backbone, syn.n_inputs_for(cfg), head, n_out, seed=5)
        renderer = Panda3dBatchRenderer(object_dataset, n_workers=1, preload_cache=True)
        model = build_pose_model(cfg, sd, renderer, MeshDataBase.from_object_ds(object_dataset).batched().cuda())
        rng = np.random.RandomState(3)
        rows = 5
        labels = [object_dataset[int(i)].label for i in rng.randint(0, len(object_dataset.list_objects), rows)]
        T0 = torch.from_numpy(np.stack([syn.random_pose(rng, (0.45, 0.6), 0.1) for _ in labels])).cuda()
        K = torch.from_numpy(np.repeat(syn.K_EXAMPLE[None], rows, 0)).float().cuda()
        images = (torch.round(torch.rand(rows, 3, 480, 640, generator=torch.Generator().manual_seed(1)) * 255) / 255).cuda()
        if cfg.input_depth:   # observation depth (metres) around the objects' distance, 5 % invalid (0): the RGBD validity rule of the crop
            gd = torch.Generator().manual_seed(2)
            depth = 0.3 + 0.5 * torch.rand(rows, 1, 480, 640, generator=gd)
            depth[torch.rand(rows, 1, 480, 640, generator=gd) < 0.05] = 0.0
            images = torch.cat([images, depth.cuda()], dim=1)
        res = {}
        for records in (True, False):
            model.stem_records = records
            model._x.clear()
            if role == "refiner":
                out = model(images=images, K=K, labels=labels, TCO=T0, n_iterations=1)["iteration=1"]
                net = out.network_outputs["pose"]
            else:
                out = model.forward_coarse(images=images, K=K, labels=labels, TCO_input=T0, return_debug_data=True)
                net = out["logits"]
            assert (model._x[0].dtype == torch.bfloat16) == records
            n_in = syn.n_inputs_for(cfg)
            res[records] = (model._nchw_view(rows, 0, n_in).clone(), net.clone())
        assert torch.equal(res[True][0], res[False][0]), role
        assert res[False][0][:, 3:].abs().max() > 0.1   # the renders are not empty
        if cfg.render_depth:   # normalised depth channels (tCR_scale_clamp_center): background = -1, object pixels around 0
            d = res[True][0][:, 4 + 6]
            assert d.min().item() == -1.0 and (d > -0.5).float().mean().item() > 0.01
        s = max(1.0, res[False][1].abs().max().item())
        assert (res[True][1] - res[False][1]).abs().max().item() < 1e-5 * s


def test_record_mode_is_refused_where_it_does_not_apply(eng):
    """records outside 16..48 elements keep the fp32 tensor; the C entry point refuses what it cannot do (and never packs a blob inside a
    forward: mp_backbone_xrec_prepare is the explicit step)"""
    from megapose6d_amd._lib import EngineError
    from tests.support import synthetic as syn

    assert eng.xrec_elements(3, 24) == 40 and eng.xrec_elements(3, 6) == 16 and eng.xrec_elements(4, 28) == 40
    sd = syn.make_state_dict("resnet34", 32, "pose", 9, seed=1)
    bb = eng.Backbone("resnet34", 32, "pose", 9, sd)
    assert bb.xrec_elements(3) == 40  # 3 * 3 + 29 = 38 -> 40 (the library could; PosePredictor never asks for a depth model)
    assert bb.xrec_elements(32) == 0  # 96 elements: no
    rec = eng.padded_nhwc(1, 240, 320, 40, 2, "cuda", dtype=torch.bfloat16)
    out = torch.empty(1, 9, device="cuda")
    with pytest.raises(EngineError):
        bb.forward(rec, 1, 240, 320, out, None, None, n_f32=32)


def test_record_launch_refuses_inconsistent_channel_masks(eng, object_dataset):
    """mp_raster_render_xrec: which channels are fp32-kind is the caller's statement and must agree with what the launch writes where --
    crop + depth channels fp32-kind, rgb / normals integers; the record length must be mp_xrec_elements of that split; depth
    normalisation needs tCR.  Error behaviour only (the values are covered by the record-output test above)."""
    from megapose6d_amd import mesh_io
    from megapose6d_amd._lib import EngineError

    db = eng.MeshDB([mesh_io.load_rigid_object(o) for o in object_dataset.list_objects])
    V, h, w, C, nper = 4, 240, 320, 4, 7
    mask = RGBD_MASK
    R = 48
    T = torch.eye(4, device="cuda").repeat(V, 1, 1)
    T[:, 2, 3] = 0.5
    K = torch.tensor([[500.0, 0, 160], [0, 500.0, 120], [0, 0, 1]], device="cuda").repeat(V, 1, 1)
    ids = torch.zeros(V, dtype=torch.int32, device="cuda")
    images = eng.PackedObservation(torch.rand(1, C, 480, 640, device="cuda"))
    im_ids = torch.zeros(1, dtype=torch.int32, device="cuda")
    boxes = torch.tensor([[100.0, 80, 400, 305]], device="cuda")
    tCR = torch.tensor([[0.0, 0.0, 0.5]], device="cuda")

    def launch(mask_, R_, tcr_, mode=2):
        x = torch.zeros(1, h, w, R_, device="cuda", dtype=torch.bfloat16)
        eng.raster_render(db, ids, T, K, h, w, 1 | 2 | 16, eng.make_lights(), x, h * w * R_, w * R_, R_, C, C + 3, C + 6, views_per_item=V,
                          stride_view=nper, crop=(images, im_ids, boxes, 0), xrec=(mask_, tcr_, mode))
        torch.cuda.synchronize()
        return x

    assert launch(mask, R, tCR).abs().sum() > 0                      # the consistent call goes through
    with pytest.raises(EngineError, match="fp32-kind channel mask"):
        launch(mask & ~(1 << 10), R, tCR)                            # a depth channel not marked fp32-kind
    with pytest.raises(EngineError, match="fp32-kind channel mask"):
        launch(mask | (1 << 5), R, tCR)                              # an rgb channel marked fp32-kind
    with pytest.raises(EngineError, match="record length"):
        launch(mask, 40, tCR)                                        # wrong record length for 8 fp32-kind + 24 integer channels
    with pytest.raises(EngineError, match="d_tCR"):
        launch(mask, R, None)                                        # depth normalisation without the reference depths
    with pytest.raises(EngineError):
        launch(mask, R, tCR, mode=7)                                 # unknown normalisation mode


@pytest.mark.parametrize("K,pool", [(7, True), (7, False), (5, False)])
def test_stem_background_tiles_take_the_short_walk(eng, K, pool):
    """Round 5: a stem workgroup whose input patch lies in 8x8-pixel tiles no view reaches (the rasteriser's job flags) walks only the
    record chunks of the observation crop (mp_conv_stem_xrec_sparse).  Refiner-shaped records (3 fp32-kind + 24 integer channels = 40
    elements, q_use = 2 of 5 chunks), an "object" blob in the middle of every image, flags derived from the data.  Against the dense walk
    (same products, the evaluated ones grouped into MFMAs differently: fp32 round-off), against float64, bit-identical when every flag
    says "geometry", and the background workgroups are counted."""
    N, nf, nu, H, W, Cout = 3, 3, 24, 96, 128, 64
    g = torch.Generator().manual_seed(70 + K)
    x = torch.cat([torch.rand(N, nf, H, W, generator=g), torch.randint(0, 256, (N, nu, H, W), generator=g).float() / 255.0], dim=1)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for n in range(N):   # geometry only inside an ellipse; image 2 is all background
        inside = ((yy - 48 - 6 * n) / 22.0) ** 2 + ((xx - 60 + 9 * n) / 30.0) ** 2 < 1.0 if n < 2 else torch.zeros(H, W, dtype=torch.bool)
        x[n, nf:] *= inside
    flags = (x[:, nf:] != 0).any(dim=1).view(N, H // 8, 8, W // 8, 8).any(dim=4).any(dim=2).to(torch.uint8).cuda().contiguous()
    assert 0.05 < flags.float().mean().item() < 0.5 and flags[2].sum().item() == 0
    w = torch.randn(Cout, nf + nu, K, K, generator=g) * (2.0 / ((nf + nu) * K * K)) ** 0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    rec = to_records(eng, x, nf, pad)
    wp = torch.from_numpy(eng.conv_stem_pack_weights(w.numpy(), nf, scale.numpy())).cuda()
    ws = torch.from_numpy(eng.conv_stem_pack_weights_sparse(w.numpy(), nf, scale.numpy())).cuda()
    assert ws.numel() < wp.numel() // 2
    assert eng.conv_stem_pack_weights_sparse(torch.randn(64, 9, K, K).numpy(), 3) is None   # coarse records (16 elements): nothing to skip

    def run(w_sparse, fl):
        y = eng.padded_nhwc(N, Ho, Wo, Cout, 1, "cuda")
        q = eng.padded_nhwc(N, Hq, Wq, Cout, 1, "cuda") if pool else None
        eng.conv_stem_xrec(rec, N, H, W, nf + nu, nf, pad, wp, bias.cuda(), Cout, K, pad, y, 1, relu=True, y_pool=q, pool_border=1,
                           w_sparse=w_sparse, tile_flags=fl)
        torch.cuda.synchronize()
        return y, q

    y_dense, q_dense = run(None, None)
    eng.profile_begin()
    eng.conv_stem_bg_stats(reset=True)
    y_sp, q_sp = run(ws, flags)
    eng.profile_end()
    bg, total = eng.conv_stem_bg_stats(reset=True)
    assert total > 0 and 0.2 * total < bg < total, (bg, total)          # a good part of the workgroups took the short walk, not all
    y_all, q_all = run(ws, torch.ones_like(flags))                       # every tile "has geometry": the dense walk, bit for bit
    assert torch.equal(y_all, y_dense) and (not pool or torch.equal(q_all, q_dense))
    ref = F.relu(F.conv2d(x.double(), (w * scale.view(-1, 1, 1, 1)).double(), bias.double(), stride=2, padding=pad))
    sc = max(1.0, ref.abs().max().item())
    got = eng.padded_view(y_sp, N, Ho, Wo, Cout, 1).permute(0, 3, 1, 2).cpu().double()
    assert (got - ref).abs().max().item() < 1e-5 * sc
    assert (y_sp - y_dense).abs().max().item() < 2e-6 * sc              # same exact products, another grouping of the fp32 additions
    if pool:
        assert (q_sp - q_dense).abs().max().item() < 2e-6 * sc
        pref = F.max_pool2d(ref, 3, 2, 1)
        assert (eng.padded_view(q_sp, N, Hq, Wq, Cout, 1).permute(0, 3, 1, 2).cpu().double() - pref).abs().max().item() < 1e-5 * sc
