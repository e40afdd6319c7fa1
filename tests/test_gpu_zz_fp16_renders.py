"""-m gpu: the "fp16 renders" mode (BASELINE.json configs[4]; MP_RASTER_F16 + mp_backbone_forward_f16 / mp_conv_desc.x_f16).

The mode stores the CNN input (renders + observation crop) as IEEE binary16 and widens it inside the stem convolution, so each
piece has an EXACT statement in terms of the fp32 path, checked bit for bit through the C-ABI:
  * rasteriser:   f16 output == round-to-nearest-even(fp32 output) of the same launch;
  * stem conv:    conv(x as halves) == conv(the same values as floats)  (identical tile, K order and MFMA sequence);
  * backbone:     forward_f16(x) == forward(float(x)) at a batch that runs the stem single-pass;
  * normalize_depth_f16 == round(normalize_depth(float(x))).
End to end the HIP pipeline in this mode is compared with the oracle run with `input_f16` (the same rounding applied to its
CNN input, oracle/pipeline.py) in tests/test_gpu_zzzz_oracle_heavy.py."""
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from megapose6d_amd import engine

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return engine


@pytest.fixture(scope="module")
def scene72():
    from tests.support import synthetic as syn
    from tests.support.scene import make_scene

    tmp = tempfile.mkdtemp(prefix="mp_t16_")
    est, obs, det, gt = make_scene(n_objects=1, seed=0, SO3_grid_size=72, tmp_dir=tmp)
    ds = syn.make_object_dataset(tmp, n_objects=1, seed=0)
    return ds, est, obs, det, gt


def _bits(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().view(torch.int16 if t.dtype == torch.float16 else torch.int32)


@pytest.mark.parametrize("msaa", [1, 4])
def test_raster_f16_is_the_rounded_f32_render(scene72, msaa):
    from megapose6d_amd.types import Panda3dLightData, make_scene_lights

    ds, est, obs, det, gt = scene72
    r = est.coarse_model.renderer
    n = 3
    T = torch.from_numpy(np.stack([gt[0]] * n).astype(np.float32)).cuda()
    T[1, 0, 3] += 0.03
    T[2, :3, :3] = T[2, :3, :3] @ torch.tensor([[0.8, -0.6, 0], [0.6, 0.8, 0], [0, 0, 1.0]], device="cuda")
    K = obs.K[:1].repeat(n, 1, 1)
    labels = [ds.list_objects[0].label] * n
    old = r.msaa
    r.msaa = msaa
    try:
        for lights in ([[Panda3dLightData("ambient", (1.0, 1.0, 1.0, 1.0))]] * n, [make_scene_lights()] * n):
            a = r.render(labels, T, K, lights, (240, 320), render_depth=True, render_normals=True)
            b = r.render(labels, T, K, lights, (240, 320), render_depth=True, render_normals=True, output_dtype=torch.float16)
            assert b.rgbs.dtype == torch.float16 and b.depths.dtype == torch.float16
            assert a.rgbs.abs().sum() > 0
            for fa, fb in ((a.rgbs, b.rgbs), (a.normals, b.normals), (a.depths, b.depths)):
                assert torch.equal(_bits(fa.half()), _bits(fb))
    finally:
        r.msaa = old


def test_fused_crop_and_renders_f16_is_the_rounded_f32_cnn_input(scene72):
    """one refiner step in both modes: every element of the half-precision CNN input is the rounded element of the fp32 one"""
    ds, est, obs, det, gt = scene72
    ref = est.refiner_model
    T0 = torch.from_numpy(np.stack([gt[0]] * 3).astype(np.float32)).cuda()
    T0[:, :3, 3] += torch.tensor([0.01, -0.01, 0.02], device="cuda")
    T0[1, 0, 3] += 0.02
    kw = dict(images=obs.images, K=obs.K.repeat(3, 1, 1), labels=[ds.list_objects[0].label] * 3, TCO=T0, n_iterations=1,
              im_ids=torch.zeros(3, dtype=torch.int32, device="cuda"))
    assert ref.render_dtype == torch.float32
    a = ref(**kw)["iteration=1"]
    ref.render_dtype = torch.float16
    try:
        b = ref(**kw)["iteration=1"]
        assert ref._x[0].dtype == torch.float16
    finally:
        ref.render_dtype = torch.float32
    xa = torch.cat([a.images_crop, a.renders], 1)
    xb = torch.cat([b.images_crop, b.renders], 1)
    assert xb.dtype == torch.float32   # callers always see fp32 crops / renders
    assert torch.equal(xa.half().float(), xb)
    # and the network sees it: outputs differ from the fp32 run only by the effect of the input rounding (<= 2^-11 relative per element)
    pa, pb = a.network_outputs["pose"], b.network_outputs["pose"]
    assert torch.isfinite(pb).all() and (pa - pb).abs().max().item() < 1e-2 * max(1.0, pa.abs().max().item())


STEM_CASES = [
    # N, Cin, H, W, K, stride, pad, in_border
    (2, 9, 48, 64, 7, 2, 3, 3),      # coarse stem: C 9 -> 12, run 84 (ragged)
    (1, 27, 30, 40, 7, 2, 3, 3),     # refiner stem: 27 -> 28, run 196 (ragged)
    (3, 32, 30, 40, 5, 2, 2, 2),     # WideResNet RGBD stem: run 160 (chunk aligned)
    (2, 32, 24, 32, 7, 2, 3, 3),     # vanilla RGBD stem: run 224 (chunk aligned)
    (2, 27, 240, 320, 7, 2, 3, 3),   # full-size rows: 300 tiles, 5 400+ loads per lane offset range
]


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_conv_on_half_input_is_bit_identical(eng, case):
    import torch.nn.functional as F

    N, Cin, H, W, K, s, p, ib = case
    Cout = 64
    g = torch.Generator().manual_seed(7 + Cin + K)
    x = torch.randn(N, Cin, H, W, generator=g).half()            # exactly representable in both formats
    w = torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (Cin * K * K)) ** 0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    cp = (Cin + 3) // 4 * 4
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    wp = torch.from_numpy(eng.conv_pack_weights(w.numpy(), cp, scale.numpy())).cuda()
    outs = {}
    for dt in (torch.float32, torch.float16):
        xb = eng.padded_nhwc(N, H, W, cp, ib, "cuda", dtype=dt)
        eng.padded_view(xb, N, H, W, cp, ib)[..., :Cin] = x.permute(0, 2, 3, 1).cuda().to(dt)
        yb = eng.padded_nhwc(N, Ho, Wo, Cout, 1, "cuda")
        eng.conv2d_nhwc(xb, N, H, W, cp, ib, wp, bias.cuda(), Cout, K, s, p, yb, 1, relu=True)   # no split-K scratch: single pass
        outs[dt] = yb
    torch.cuda.synchronize()
    assert torch.equal(_bits(outs[torch.float32]), _bits(outs[torch.float16]))
    ref = F.relu(F.conv2d(x.float(), w * scale.view(-1, 1, 1, 1), bias, stride=s, padding=p))
    got = eng.padded_view(outs[torch.float16], N, Ho, Wo, Cout, 1).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


def test_half_input_is_refused_where_it_is_not_implemented(eng):
    from megapose6d_amd._lib import EngineError

    x = eng.padded_nhwc(1, 8, 8, 64, 1, "cuda", dtype=torch.float16)
    w = np.zeros((128, 64, 3, 3), np.float32)
    y = eng.padded_nhwc(1, 8, 8, 128, 1, "cuda")
    with pytest.raises(EngineError):   # Cout > 64
        eng.conv2d_nhwc(x, 1, 8, 8, 64, 1, torch.from_numpy(eng.conv_pack_weights(w, 64)).cuda(), None, 128, 3, 1, 1, y, 1)


@pytest.mark.parametrize("kind,c_in", [("vanilla_resnet34", 9), ("vanilla_resnet34", 27), ("resnet34", 32)])
def test_backbone_forward_f16_equals_forward_on_the_widened_input(eng, kind, c_in):
    from tests.support import synthetic as syn

    head, n_out = ("pose", 9) if c_in != 9 else ("logits", 1)
    sd = syn.make_state_dict(kind, c_in, head, n_out, seed=5)
    bb = eng.Backbone(kind, c_in, head, n_out, sd)
    b, h, w = 3, 240, 320   # 450 stem tiles: the fp32 forward runs the stem single-pass too (no split-K), so the sums are ordered alike
    g = torch.Generator().manual_seed(c_in)
    x = torch.rand(b, h, w, c_in, generator=g).half()
    res = {}
    for dt in (torch.float32, torch.float16):
        xb = eng.padded_nhwc(b, h, w, bb.c_in_p, bb.in_border, "cuda", dtype=dt)
        eng.padded_view(xb, b, h, w, bb.c_in_p, bb.in_border)[..., :c_in] = x.cuda().to(dt)
        out = torch.empty(b, n_out, device="cuda")
        feat = torch.empty(b, 512, device="cuda")
        bb.forward(xb, b, h, w, out, feat=feat)
        res[dt] = (out, feat)
    torch.cuda.synchronize()
    assert torch.isfinite(res[torch.float16][1]).all() and res[torch.float16][1].abs().max() > 0
    assert torch.equal(res[torch.float32][0], res[torch.float16][0]) and torch.equal(res[torch.float32][1], res[torch.float16][1])


def test_normalize_depth_f16(eng):
    b, h, w, C, border = 2, 12, 20, 8, 2
    g = torch.Generator().manual_seed(3)
    vals = (torch.rand(b, h, w, C, generator=g) * 2.0).half()
    tCR = torch.tensor([[0.0, 0.0, 0.7], [0.1, 0.0, 1.3]])
    for mode in (1, 2, 3):
        xh = eng.padded_nhwc(b, h, w, C, border, "cuda", dtype=torch.float16)
        xf = eng.padded_nhwc(b, h, w, C, border, "cuda")
        eng.padded_view(xh, b, h, w, C, border)[:] = vals.cuda()
        eng.padded_view(xf, b, h, w, C, border)[:] = vals.cuda().float()
        eng.normalize_depth(xh, b, h, w, border, C, [3, 6], tCR.cuda(), mode)
        eng.normalize_depth(xf, b, h, w, border, C, [3, 6], tCR.cuda(), mode)
        torch.cuda.synchronize()
        assert torch.equal(_bits(xf.half()), _bits(xh)), mode
