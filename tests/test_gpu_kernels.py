"""-m gpu: kernel-level parity of libmp_engine.so (called through the C-ABI) against the CPU oracle.

Tolerances (stated per test): rasteriser bit-exact (integer coverage + explicit fmaf contract);
roi_align / pose math <= 1e-5 abs (same fp32 formulas, different contraction); every convolution kernel <= 2e-5 of the output scale
against the float64 sum of the fp32 operands (CONV_TOL); backbone features <= 1e-4 abs
relative to the activation scale (fp32 MFMA k-ordered fmaf chain vs MKL-DNN's blocked summation).
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
    n_cu, lds, arch = engine.device_info()
    assert arch.startswith("gfx950")
    return engine


def _to_padded(eng, x_nchw, cp, border):
    n, c, h, w = x_nchw.shape
    buf = eng.padded_nhwc(n, h, w, cp, border, "cuda")
    v = eng.padded_view(buf, n, h, w, cp, border)
    v[..., :c] = x_nchw.permute(0, 2, 3, 1).cuda()
    return buf


def _from_padded(eng, buf, n, h, w, c, border):
    return eng.padded_view(buf, n, h, w, c, border).permute(0, 3, 1, 2).contiguous().cpu()


CONV_CASES = [
    # N, Cin, H, W, Cout, K, stride, pad, in_border
    (2, 9, 48, 64, 64, 7, 2, 3, 3),     # coarse stem (C padded 9 -> 12, run 84 -> 96)
    (1, 27, 30, 40, 64, 7, 2, 3, 3),    # refiner stem (27 -> 28)
    (1, 32, 30, 40, 64, 5, 2, 2, 2),    # wide-resnet RGBD stem
    (3, 64, 15, 20, 64, 3, 1, 1, 1),    # layer1
    (2, 64, 15, 21, 128, 3, 2, 1, 1),   # layer2.0.conv1 (odd width)
    (2, 64, 15, 21, 128, 1, 2, 0, 1),   # downsample
    (2, 128, 8, 10, 256, 3, 2, 1, 1),
    (5, 256, 4, 5, 512, 3, 1, 1, 1),    # M = 100: partial tile
    (1, 512, 8, 10, 512, 3, 1, 1, 2),   # border larger than pad
]


# What the hardware achieves (profiles/r04_wino_bf16_native_check.txt, r04_stem_native_check_v1.txt: <= 7e-6 of the output scale for every
# convolution kernel against the direct fp32 sum) with a margin of 3: a kernel that loses a piece product or a bit of an operand (2^-16
# relative and up) fails this; the round-4 bound of 2e-4 would have let a 20x regression pass.  The reference sum is formed in float64
# from the fp32 operands (the folded weight w * scale rounded to fp32 first, as the packers do), so the bound measures OUR error only.
CONV_TOL = 2e-5


def _conv_ref_f64(x, w, scale, bias, stride, pad):
    wf = (w.double() * scale.double().view(-1, 1, 1, 1)).float() if scale is not None else w
    y = F.conv2d(x.double(), wf.double(), None, stride=stride, padding=pad)
    if bias is not None:
        y = y + bias.double().view(1, -1, 1, 1)
    return y.float()


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("epi", ["plain", "bias_relu", "res_relu", "dual"])
@pytest.mark.parametrize("splitk", [False, True])
def test_conv_matches_torch_fp32(eng, case, epi, splitk):
    """splitk=True hands the launch a scratch buffer: all of these cases have too few tiles to fill the chip, so they take the
    split-K kernel + deterministic reduce (csrc/conv.hip) instead of the single-pass kernel"""
    N, Cin, H, W, Cout, K, s, p, ib = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (Cin * K * K)) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    res = torch.randn(N, Cout, Ho, Wo, generator=g)
    cp = (Cin + 3) // 4 * 4
    xb = _to_padded(eng, x, cp, ib)
    use_scale = epi != "plain"
    wp = torch.from_numpy(eng.conv_pack_weights(w.numpy(), cp, scale.numpy() if use_scale else None)).cuda()
    ob = 1
    yb = eng.padded_nhwc(N, Ho, Wo, Cout, ob, "cuda")
    yb += 7.0  # poison: interior must be fully overwritten
    ya = eng.padded_nhwc(N, Ho, Wo, Cout, ob, "cuda") if epi == "dual" else None
    sc2, sh2 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    rb = _to_padded(eng, res, Cout, ob) if epi in ("res_relu", "dual") else None
    eng.conv2d_nhwc(xb, N, H, W, cp, ib, wp, bias.cuda() if use_scale else None, Cout, K, s, p, yb, ob,
                    residual=rb, relu=epi in ("bias_relu", "res_relu"), y_act=ya,
                    act_scale=sc2.cuda() if ya is not None else None, act_shift=sh2.cuda() if ya is not None else None,
                    splitk_ws=torch.empty(4 << 20, device="cuda") if splitk else None)
    torch.cuda.synchronize()
    ref = _conv_ref_f64(x, w, scale if use_scale else None, bias if use_scale else None, s, p)
    if epi in ("res_relu", "dual"):
        ref = ref + res
    if epi in ("bias_relu", "res_relu"):
        ref = F.relu(ref)
    got = _from_padded(eng, yb, N, Ho, Wo, Cout, ob)
    tol = CONV_TOL * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < tol
    # border untouched (still the poison value)
    full = yb[: N * (Ho + 2) * (Wo + 2) * Cout].view(N, Ho + 2, Wo + 2, Cout)
    assert torch.all(full[:, 0] == 7.0) and torch.all(full[:, :, 0] == 7.0)
    if epi == "dual":
        ref_a = F.relu(ref * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1))
        got_a = _from_padded(eng, ya, N, Ho, Wo, Cout, ob)
        assert (got_a - ref_a).abs().max().item() < tol * 2


WINO_CASES = [
    # N, Cin, H, W, Cout, in_border
    (3, 64, 12, 16, 64, 1),
    (5, 64, 15, 20, 128, 1),    # odd height: the last tile row is half empty
    (2, 128, 13, 11, 64, 1),    # odd height and width
    (4, 256, 8, 10, 256, 1),
    (70, 16, 6, 6, 64, 2),      # 630 tiles: ten workgroups, the last one partial; border larger than the pad
    (2, 512, 8, 10, 512, 1),
    (64, 64, 30, 38, 128, 1),   # 18 240 tiles x 2 channel blocks = 570 units > 256 CUs: the PERSISTENT launch form (round 6), ragged: 58 CUs walk 3 units, 198 walk 2
]


@pytest.mark.parametrize("case", WINO_CASES)
@pytest.mark.parametrize("epi", ["plain", "bias_relu", "res_relu", "dual"])
@pytest.mark.parametrize("kernel", ["bf16x9", "fp32"])
def test_winograd_conv_matches_torch_fp32(eng, case, epi, kernel):
    """the fused Winograd F(2x2, 3x3) kernels -- mp_conv3x3_wino_bf16_nhwc (exact bf16 pieces, csrc/conv_wino_bf16.hip: the default of the
    backbone) and mp_conv3x3_wino_nhwc (fp32 MFMA, csrc/conv_wino.hip) -- against torch's fp32 convolution, every fused epilogue;
    the slack the kernel may read behind an odd-sized input is poisoned with NaN (nothing read there may reach an output).
    Reference layers: models/torchvision_resnet.py:74-120, models/wide_resnet.py:29-56."""
    N, Cin, H, W, Cout, ib = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, Cout, H, W, generator=g)
    xb0 = _to_padded(eng, x, Cin, ib)
    n_x = N * (H + 2 * ib) * (W + 2 * ib) * Cin
    xb = torch.full((n_x + (W + 2 * ib + 1) * Cin + 64,), float("nan"), device="cuda")
    xb[:n_x] = xb0.flatten()[:n_x]
    use_scale = epi != "plain"
    pack = eng.conv_wino_bf16_pack_weights if kernel == "bf16x9" else eng.conv_wino_pack_weights
    up = torch.from_numpy(pack(w.numpy(), Cin, scale.numpy() if use_scale else None)).cuda()
    ob = 1
    yb = eng.padded_nhwc(N, H, W, Cout, ob, "cuda")
    yb += 7.0  # poison: interior must be fully overwritten, the border untouched
    ya = eng.padded_nhwc(N, H, W, Cout, ob, "cuda") if epi == "dual" else None
    sc2, sh2 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    rb = _to_padded(eng, res, Cout, ob) if epi in ("res_relu", "dual") else None
    eng.conv3x3_wino_nhwc(xb, N, H, W, Cin, ib, up, bias.cuda() if use_scale else None, Cout, yb, ob, residual=rb,
                          relu=epi in ("bias_relu", "res_relu"), y_act=ya, act_scale=sc2.cuda() if ya is not None else None,
                          act_shift=sh2.cuda() if ya is not None else None)
    torch.cuda.synchronize()
    ref = _conv_ref_f64(x, w, scale if use_scale else None, bias if use_scale else None, 1, 1)
    if epi in ("res_relu", "dual"):
        ref = ref + res
    if epi in ("bias_relu", "res_relu"):
        ref = F.relu(ref)
    got = _from_padded(eng, yb, N, H, W, Cout, ob)
    assert torch.isfinite(got).all()
    tol = CONV_TOL * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < tol
    full = yb[: N * (H + 2) * (W + 2) * Cout].view(N, H + 2, W + 2, Cout)
    assert torch.all(full[:, 0] == 7.0) and torch.all(full[:, :, 0] == 7.0) and torch.all(full[:, -1] == 7.0) and torch.all(full[:, :, -1] == 7.0)
    if epi == "dual":
        ref_a = F.relu(ref * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1))
        got_a = _from_padded(eng, ya, N, H, W, Cout, ob)
        assert (got_a - ref_a).abs().max().item() < tol * 2


EDGE_SCALES = [
    # (name, operand scale, weight scale): what the exact-piece split meets at the ends of the fp32 range
    ("tiny_1e-30", 1e-30, 1.0),            # every piece still a normal number (2^-100 .. 2^-124)
    ("huge_1e30", 1e30, 1e-2),             # products ~1e28, sums stay finite
    ("third_piece_subnormal", 2.0 ** -115, 1.0),   # x1 ~ 2^-115, x2 ~ 2^-123 normal, x3 ~ 2^-131: a SUBNORMAL bf16
    ("tiny_weights_1e-30", 1.0, 1e-30),    # the same on the weight side (split on the host)
]


@pytest.mark.parametrize("name,sx,sw", EDGE_SCALES)
@pytest.mark.parametrize("kernel", ["bf16x9", "fp32_wino", "direct"])
def test_exact_piece_kernels_at_the_ends_of_the_fp32_range(eng, name, sx, sw, kernel):
    """The truncation split x = x1 + x2 + x3 (csrc/conv_wino_bf16.hip WB_SP*, host: split3) is exact for every finite fp32 value -- the
    pieces are differences of fp32 numbers -- but the bf16 MFMA treats a SUBNORMAL operand piece as zero.  That can only touch a piece
    below 2^-126, i.e. a contribution below 2^-126 x |weight| per product: the documented behaviour is "equal to the float64 sum of the
    fp32 operands to CONV_TOL of the output scale, plus at most 1e-37 absolute" (torch's CPU fp32 path keeps subnormals; so does the
    reference).  Mixed signs throughout (randn operands, truncation keeps the sign of every piece).  256 -> 64 channels, 3x3."""
    N, Cin, H, W, Cout = 2, 256, 8, 10, 64
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(N, Cin, H, W, generator=g).double() * sx).float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g).double() * (2.0 / (Cin * 9)) ** 0.5 * sw).float()
    assert torch.isfinite(x).all() and torch.isfinite(w).all() and x.abs().max() > 0 and (x < 0).any() and (w < 0).any()
    xb = _to_padded(eng, x, Cin, 1)
    yb = eng.padded_nhwc(N, H, W, Cout, 1, "cuda")
    if kernel == "direct":
        wp = torch.from_numpy(eng.conv_pack_weights(w.numpy(), Cin, None)).cuda()
        eng.conv2d_nhwc(xb, N, H, W, Cin, 1, wp, None, Cout, 3, 1, 1, yb, 1)
    else:
        pack = eng.conv_wino_bf16_pack_weights if kernel == "bf16x9" else eng.conv_wino_pack_weights
        xs = torch.zeros(xb.numel() + (W + 3) * Cin + 64, device="cuda")
        xs[: xb.numel()] = xb.flatten()
        eng.conv3x3_wino_nhwc(xs, N, H, W, Cin, 1, torch.from_numpy(pack(w.numpy(), Cin, None)).cuda(), None, Cout, yb, 1)
    torch.cuda.synchronize()
    got = _from_padded(eng, yb, N, H, W, Cout, 1).double()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert scale > 0 and torch.isfinite(got).all()
    # (Winograd: the transform forms sums of up to four inputs / 0.25 x sums of nine weights before the product: their own fp32 rounding is
    #  part of CONV_TOL; nothing here may lose more than a subnormal piece)
    assert err < CONV_TOL * scale + 1e-37, (name, kernel, err, scale)


def test_backbone_takes_the_winograd_path_at_full_batch(eng):
    """a 576-row forward runs its 3x3 / stride-1 layers on the Winograd kernel (profiler row present), a 2-row forward stays on the direct
    kernel's split-K path (grid too small), and both agree with each other on the rows they share"""
    from tests.support import synthetic as syn

    sd = syn.make_state_dict("vanilla_resnet34", 9, "logits", 1, seed=5)
    bb = eng.Backbone("vanilla_resnet34", 9, "logits", 1, sd)
    g = torch.Generator().manual_seed(0)
    x2 = torch.rand(2, 9, 240, 320, generator=g)
    x = x2.repeat(48, 1, 1, 1)   # 96 rows
    outs = {}
    for name, xx in (("small", x2), ("large", x)):
        b = xx.shape[0]
        xb = _to_padded(eng, xx, bb.c_in_p, bb.in_border)
        out, feat = torch.empty(b, 1, device="cuda"), torch.empty(b, 512, device="cuda")
        eng.profile_begin()
        bb.forward(xb, b, 240, 320, out, None, feat)
        prof = eng.profile_end()
        outs[name] = (feat.cpu(), prof)
    assert not any(k.startswith("conv3x3_wino") for k in outs["small"][1])
    assert any(k.startswith("conv3x3_wino") for k in outs["large"][1])
    assert (outs["large"][0][:2] - outs["small"][0]).abs().max().item() < 2e-5
    assert torch.equal(outs["large"][0][:2], outs["large"][0][2:4])


def test_conv_splitk_is_taken_and_deterministic(eng):
    """layer4-sized conv at batch 1 (M = 80 -> 4 tiles): the split-K path runs (profiler sees its kernels) and is bit-reproducible"""
    g = torch.Generator().manual_seed(1)
    N, Cin, H, W, Cout = 1, 512, 8, 10, 512
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02
    xb = _to_padded(eng, x, Cin, 1)
    wp = torch.from_numpy(eng.conv_pack_weights(w.numpy(), Cin, None)).cuda()
    ws = torch.empty(4 << 20, device="cuda")
    outs = []
    eng.profile_begin()
    for _ in range(2):
        yb = eng.padded_nhwc(N, H, W, Cout, 1, "cuda")
        eng.conv2d_nhwc(xb, N, H, W, Cin, 1, wp, None, Cout, 3, 1, 1, yb, 1, splitk_ws=ws)
        outs.append(yb.clone())
    prof = eng.profile_end()
    assert "conv_splitk_reduce" in prof and any(k.endswith("/splitk") for k in prof)
    assert torch.equal(outs[0], outs[1])
    y0 = eng.padded_nhwc(N, H, W, Cout, 1, "cuda")
    eng.conv2d_nhwc(xb, N, H, W, Cin, 1, wp, None, Cout, 3, 1, 1, y0, 1)          # single-pass kernel
    ref = _conv_ref_f64(x, w, None, None, 1, 1)
    assert (_from_padded(eng, outs[0], N, H, W, Cout, 1) - ref).abs().max() < CONV_TOL * ref.abs().max()
    assert (outs[0] - y0).abs().max() < CONV_TOL * ref.abs().max()


def test_maxpool_and_tail(eng):
    g = torch.Generator().manual_seed(3)
    N, C, H, W = 3, 64, 30, 41
    x = torch.rand(N, C, H, W, generator=g)
    xb = _to_padded(eng, x, C, 1)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    yb = eng.padded_nhwc(N, Ho, Wo, C, 1, "cuda")
    eng.maxpool3x3s2(xb, N, H, W, C, 1, yb, 1)
    ref = F.max_pool2d(x, 3, 2, 1)
    assert torch.equal(_from_padded(eng, yb, N, Ho, Wo, C, 1), ref)
    # tail: avgpool + fc + head + sigmoid
    C2 = 512
    x2 = torch.randn(4, C2, 8, 10, generator=g)
    fcw, fcb = torch.randn(512, 512, generator=g) * 0.05, torch.randn(512, generator=g)
    hw, hb = torch.randn(9, 512, generator=g) * 0.05, torch.randn(9, generator=g)
    out = torch.empty(4, 9, device="cuda")
    sig = torch.empty(4, 9, device="cuda")
    feat = torch.empty(4, 512, device="cuda")
    eng.pool_fc_heads(_to_padded(eng, x2, C2, 1), 4, 8, 10, C2, 1, fcw.cuda(), fcb.cuda(), 512, hw.cuda(), hb.cuda(), 9, feat, out, sig)
    f_ref = F.linear(x2.mean(dim=(2, 3)), fcw, fcb)
    o_ref = F.linear(f_ref, hw, hb)
    assert (feat.cpu() - f_ref).abs().max() < 1e-4
    assert (out.cpu() - o_ref).abs().max() < 1e-4
    assert (sig.cpu() - torch.sigmoid(o_ref)).abs().max() < 1e-5


@pytest.mark.parametrize("kind,c_in,head,n_out", [("vanilla_resnet34", 9, "logits", 1), ("vanilla_resnet34", 27, "pose", 9),
                                                  ("resnet34", 27, "pose", 9), ("resnet18", 9, "logits", 1),
                                                  ("vanilla_resnet34", 32, "pose", 9), ("resnet34_width=2", 9, "logits", 1)])
def test_backbone_matches_oracle(eng, kind, c_in, head, n_out):
    from tests.support import synthetic as syn
    from oracle import backbones as ob

    sd = syn.make_state_dict(kind, c_in, head, n_out, seed=1)
    bb = eng.Backbone(kind, c_in, head, n_out, sd)
    b, h, w = 3, 240, 320
    g = torch.Generator().manual_seed(0)
    x = torch.rand(b, c_in, h, w, generator=g)
    xb = _to_padded(eng, x, bb.c_in_p, bb.in_border)
    out = torch.empty(b, n_out, device="cuda")
    feat = torch.empty(b, 512 * getattr(bb, "width", 1), device="cuda")
    bb.forward(xb, b, h, w, out, None, feat)
    # run twice: the second call must reuse the workspace (no re-zeroing) and give identical results
    out2 = torch.empty_like(out)
    bb.forward(xb, b, h, w, out2, None, None)
    torch.cuda.synchronize()
    ref = ob.net_forward(sd, kind, x)
    key = "pose" if head == "pose" else "renderings_logits"
    assert ref["features"].abs().max().item() < 10.0  # O(1) features: absolute bounds
    assert (feat.cpu() - ref["features"]).abs().max().item() < 1e-4
    assert (out.cpu() - ref[key]).abs().max().item() < 1e-4
    assert torch.equal(out, out2)
    assert abs(bb.flops(1, 240, 320) / 1e9 - {9: 12.068, 27: 14.236, 32: 14.838}[c_in]) < 0.01 or kind != "vanilla_resnet34"


def _mesh_db(eng, engine_meshes):
    return eng.MeshDB(engine_meshes)


@pytest.mark.parametrize("flags,lit", [(1, False), (3, False), (1 | 4, False), (0, True), (16 | 3, False), (16 | 1 | 4, False), (16, True)])
def test_raster_bit_exact_vs_oracle(eng, engine_meshes, oracle_meshes, flags, lit):
    """flags: 1 normals, 2 depth, 4 GL eye axes, 16 = 4x MSAA (the reference's configuration); lit = ambient + 6 point lights"""
    from tests.support import synthetic as syn
    from oracle import raster as orr

    db = _mesh_db(eng, engine_meshes)
    rng = np.random.RandomState(5)
    n = 6
    mesh_ids = np.array([0, 1, 2, 0, 1, 2], np.int32)
    T = np.stack([syn.random_pose(rng, z_range=(0.25, 0.6)) for _ in range(n)])
    K = np.repeat(syn.K_EXAMPLE[None].astype(np.float32), n, 0)
    K[:, 0, 0] *= 0.5; K[:, 1, 1] *= 0.5; K[:, 0, 2] = 160; K[:, 1, 2] = 120
    T[3, 0, 3] = np.nan  # invalid pose -> zeros (panda3d_batch_renderer.py:109-135)
    h, w = 240, 320
    out = torch.full((n, h, w, 8), -1.0, device="cuda")
    if lit:
        dirs = orr.POINT_DIRS
        cols = [(0.4, 0.4, 0.4)] * 6
        offs = [(0.0, 0.0, 0.01 * k) for k in range(6)]
        L = eng.make_lights((0.1, 0.1, 0.1), dirs, cols, offs)
        Lo = orr.lights_struct((0.1, 0.1, 0.1), dirs, cols, offs)
    else:
        L, Lo = eng.make_lights(), orr.lights_struct()
    eng.raster_render(db, torch.from_numpy(mesh_ids).cuda(), torch.from_numpy(T).cuda(), torch.from_numpy(K).cuda(), h, w, flags,
                      L, out, h * w * 8, w * 8, 8, 0, 3, 6)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(n):
        rgb, nrm, dep = orr.render(oracle_meshes[mesh_ids[i]], T[i : i + 1], K[i : i + 1], h, w, flags, Lo)
        if i != 3:
            assert (rgb[0].sum(-1) > 0).mean() > 0.02, "object should be visible"
        if lit:  # point lights use sqrt/div chains: allow 1 LSB of the uint8 quantisation on a few pixels
            d = np.abs(got[i, :, :, 0:3] - rgb[0])
            assert d.max() <= 1.0 / 255 + 1e-7 and (d > 0).mean() < 1e-3
        else:
            assert np.array_equal(got[i, :, :, 0:3], rgb[0])
        if flags & 1:
            assert np.array_equal(got[i, :, :, 3:6], nrm[0])
        if flags & 2:
            assert np.array_equal(got[i, :, :, 6], dep[0])


@pytest.mark.parametrize("variant", ["closed", "open_and_misoriented"])
def test_raster_occlusion_bound_is_exact_on_many_views(eng, engine_meshes, oracle_meshes, variant):
    """The block visits skip a piece that provably cannot win a sample (every sample of the 4x4 block already holds a piece nearer than
    the piece's nearest vertex; raster.hip cover_batch_blocks).  That is a depth bound, not a back-face rule: it must not change a single
    bit for closed meshes (where it removes most visits of the faces that look away), for meshes with holes, and for meshes whose
    triangles are wound at random (the phase order is only a hint).  48 views per variant incl. close-ups, 4x MSAA, colours + normals +
    depth against the independent C rasteriser."""
    from tests.support import synthetic as syn
    from oracle import raster as orr

    rng = np.random.RandomState(11 if variant == "closed" else 12)
    meshes, ref_meshes = [], []          # the engine's input (product loader) | the C rasteriser's (oracle/mesh_loader.py)
    for m, mo in zip(engine_meshes, oracle_meshes):
        m, mo = dict(m), dict(mo)
        if variant != "closed":
            n_f = m["faces"].shape[0]
            assert mo["faces"].shape[0] == n_f
            keep = rng.rand(n_f) > 0.3                   # holes: a third of the faces is gone
            flip = rng.rand(int(keep.sum())) < 0.5       # and half of the rest is wound the other way
            for d in (m, mo):
                f = d["faces"].copy()[keep]
                f[flip] = f[flip][:, ::-1]
                d["faces"] = np.ascontiguousarray(f)
        meshes.append(m)
        ref_meshes.append(mo)
    db = eng.MeshDB(meshes)
    n = 48
    mesh_ids = (np.arange(n) % len(meshes)).astype(np.int32)
    T = np.stack([syn.random_pose(rng, z_range=(0.22, 0.33) if i % 4 == 0 else (0.4, 0.8), xy_frac=0.25) for i in range(n)])
    K = np.repeat(syn.K_EXAMPLE[None].astype(np.float32), n, 0)
    K[:, 0, 0] *= 0.5; K[:, 1, 1] *= 0.5; K[:, 0, 2] = 160; K[:, 1, 2] = 120
    h, w, flags = 240, 320, 16 | 3
    out = torch.full((n, h, w, 8), -1.0, device="cuda")
    eng.raster_render(db, torch.from_numpy(mesh_ids).cuda(), torch.from_numpy(T).cuda(), torch.from_numpy(K).cuda(), h, w, flags,
                      eng.make_lights(), out, h * w * 8, w * 8, 8, 0, 3, 6)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    covered = 0
    for i in range(n):
        rgb, nrm, dep = orr.render(ref_meshes[mesh_ids[i]], T[i : i + 1], K[i : i + 1], h, w, flags, orr.lights_struct())
        assert np.array_equal(got[i, :, :, 0:3], rgb[0]), i
        assert np.array_equal(got[i, :, :, 3:6], nrm[0]), i
        assert np.array_equal(got[i, :, :, 6], dep[0]), i
        covered += int((dep[0] > 0).sum())
    assert covered > 0.05 * n * h * w


def test_raster_large_triangles_and_close_camera(eng):
    """low-poly mesh (huge triangles -> block-cooperative path) and a camera inside the guard band."""
    from oracle import raster as orr

    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], [0, 0, 0.5]], np.float32) * 0.1
    f = np.array([[0, 1, 2], [0, 2, 3], [0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]], np.int32)
    nrm = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-6)
    col = np.random.RandomState(0).rand(5, 3).astype(np.float32)
    mesh = {"vertices": v, "normals": nrm.astype(np.float32), "colors": col, "faces": f}
    db = eng.MeshDB([mesh])
    T = np.tile(np.eye(4, dtype=np.float32), (5, 1, 1))
    T[:, 2, 3] = [0.3, 0.15, 0.11, 0.07, 0.09]          # the last two put part of the pyramid behind the near plane (clipping)
    T[1, :3, :3] = np.array([[0.8, 0, 0.6], [0, 1, 0], [-0.6, 0, 0.8]], np.float32)
    T[4, :3, :3] = np.array([[0.8, 0, 0.6], [0, 1, 0], [-0.6, 0, 0.8]], np.float32)
    K = np.tile(np.array([[300, 0, 160], [0, 300, 120], [0, 0, 1]], np.float32), (5, 1, 1))
    for flags in (3, 16 | 3):
        out = torch.zeros(5, 240, 320, 8, device="cuda")
        eng.raster_render(db, torch.zeros(5, dtype=torch.int32, device="cuda"), torch.from_numpy(T).cuda(), torch.from_numpy(K).cuda(),
                          240, 320, flags, eng.make_lights(), out, 240 * 320 * 8, 320 * 8, 8, 0, 3, 6)
        got = out.cpu().numpy()
        rgb, nr, dep = orr.render(mesh, T, K, 240, 320, flags)
        assert np.array_equal(got[..., 0:3], rgb) and np.array_equal(got[..., 3:6], nr) and np.array_equal(got[..., 6], dep)
        assert (dep[0] > 0).mean() > 0.1 and (dep[3] > 0).mean() > 0.1 and dep[3][dep[3] > 0].min() >= 0.1 - 1e-6


def test_raster_mid_size_triangles_take_block_visits_bit_exact(eng):
    """Fan triangles of 30 .. 120 pixels (the caps of a lathe mesh): they touch more than 16 tiles, so they are not replicated into the
    tile lists; up to 81 pixels they take the block visits (re-derived from the mesh, no record slot), beyond that the 64-bit sweep.
    Two discs a little apart with opposite orientation + a tessellated strip in front: depth interplay between pieces with and
    without a record, both occlusion-bound phases.  Bit-exact vs the oracle, with and without multisampling."""
    from oracle import raster as orr

    n = 24
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    ring = np.stack([np.cos(ang), np.sin(ang), np.zeros(n)], 1) * 0.05
    v = [np.zeros((1, 3)), ring, np.array([[0.004, -0.003, 0.012]]), ring * 0.9 + np.array([0, 0, 0.012])]
    f = [[0, 1 + i, 1 + (i + 1) % n] for i in range(n)] + [[n + 1, n + 2 + (i + 1) % n, n + 2 + i] for i in range(n)]   # second disc: flipped
    base = 2 * n + 2
    gx, gy = np.meshgrid(np.linspace(-0.03, 0.03, 13), np.linspace(-0.008, 0.008, 4))
    strip = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, -0.01)], 1)
    v.append(strip)
    for r in range(3):
        for c in range(12):
            a = base + r * 13 + c
            f += [[a, a + 1, a + 14], [a, a + 14, a + 13]]
    v = np.concatenate(v).astype(np.float32)
    f = np.asarray(f, np.int32)
    rs = np.random.RandomState(3)
    nrm = rs.randn(*v.shape).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    mesh = {"vertices": v, "normals": nrm, "colors": rs.rand(len(v), 3).astype(np.float32), "faces": f}
    db = eng.MeshDB([mesh])
    zs = [0.5, 0.3, 0.2, 0.15, 0.25, 0.3]
    T = np.tile(np.eye(4, dtype=np.float32), (len(zs), 1, 1))
    T[:, 2, 3] = zs
    T[:, 0, 3] = [0.0, 0.01, -0.02, 0.0, 0.03, -0.05]
    c, s_ = np.cos(0.7), np.sin(0.7)
    T[4, :3, :3] = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], np.float32)
    T[5, :3, :3] = np.array([[1, 0, 0], [0, -c, s_], [0, -s_, -c]], np.float32)      # seen from behind, tilted
    K = np.tile(np.array([[300, 0, 160], [0, 300, 120], [0, 0, 1]], np.float32), (len(zs), 1, 1))
    for flags in (3, 16 | 3):
        out = torch.zeros(len(zs), 240, 320, 8, device="cuda")
        eng.raster_render(db, torch.zeros(len(zs), dtype=torch.int32, device="cuda"), torch.from_numpy(T).cuda(), torch.from_numpy(K).cuda(),
                          240, 320, flags, eng.make_lights(), out, 240 * 320 * 8, 320 * 8, 8, 0, 3, 6)
        got = out.cpu().numpy()
        rgb, nr, dep = orr.render(mesh, T, K, 240, 320, flags)
        assert np.array_equal(got[..., 6], dep) and np.array_equal(got[..., 0:3], rgb) and np.array_equal(got[..., 3:6], nr)
        assert all((dep[i] > 0).mean() > 0.01 for i in range(len(zs)))


@pytest.mark.parametrize("C", [3, 4])
def test_crop_roi_align_vs_oracle(eng, C):
    from oracle import thirdparty as tp

    g = torch.Generator().manual_seed(C)
    imgs = torch.rand(2, C, 120, 160, generator=g)
    if C == 4:
        imgs[:, 3] = imgs[:, 3] * 2
        imgs[:, 3][torch.rand(2, 120, 160, generator=g) < 0.05] = 0.0
    boxes = torch.tensor([[10.3, 5.2, 90.7, 65.5], [-20.0, -10.0, 100.0, 80.0], [100.0, 60.0, 200.0, 135.0], [50.0, 50.0, 50.5, 50.2]])
    im_ids = torch.tensor([0, 1, 1, 0], dtype=torch.int32)
    oh, ow = 60, 80
    out = torch.zeros(4, oh, ow, 8, device="cuda")
    eng.crop_roi_align(imgs.cuda(), im_ids.cuda(), boxes.cuda(), oh, ow, out, oh * ow * 8, ow * 8, 8, 1)
    torch.cuda.synchronize()
    rois = torch.cat([im_ids.float()[:, None], boxes], 1)
    ref = tp.roi_align(imgs, rois, (oh, ow), sampling_ratio=4)
    if C == 4:  # cropping.py:131-142
        valid = (imgs[:, 3:4] > 0).float()
        vc = tp.roi_align(valid, rois, (oh, ow), sampling_ratio=4)
        ref[:, 3:4] = ref[:, 3:4] * (vc >= 0.99).float()
    got = out[..., 1 : 1 + C].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 1e-5
    assert torch.all(out[..., 0] == 0)


MV_CASES = [("TCO+front_1view", False, False), ("sphere_26views", False, False), ("TCO+front_3views", True, False),
            ("TCO+front_1view", True, False), ("TCO+front_3views", True, True), ("sphere_26views", True, True)]


@pytest.mark.parametrize("mvt,remove,inplane", MV_CASES)
def test_pose_prepare_every_multiview_mode_vs_oracle(eng, engine_meshes, mvt, remove, inplane):
    """mp_pose_prepare_ex for the view lists of lib3d/multiview.py:197-246 beyond the released recipe: cameras vs the oracle restatement
    (itself pinned to the reference function's outputs, tests/test_oracle_golden.py), per-view crop intrinsics from 200 points, and the
    main crop (K_crop / boxes of crop_inputs, 2000 points) whether or not the TCO view is in the list (models/pose_rigid.py:540-552)."""
    from tests.support import synthetic as syn
    from oracle import geometry as og

    rng = np.random.RandomState(13)
    b = 5
    pts = og.pad_stack_points([torch.from_numpy(m["points"]) for m in engine_meshes])
    pts_s = pts[:, og.sample_point_ids(pts.shape[1], 2000)]
    mesh_ids = torch.tensor(rng.randint(0, 3, size=b), dtype=torch.int32)
    T = torch.from_numpy(np.stack([syn.random_pose(rng) for _ in range(b)]))
    K = torch.from_numpy(np.repeat(syn.K_EXAMPLE[None], b, 0)).float()
    code = eng.MV_MODES[mvt] | (eng.MV_REMOVE_TCO if remove else 0) | (eng.MV_INPLANE if inplane else 0)
    V = eng.multiview_n_views(code)
    TCO_n, tCR, TCV, KV, brend, bcrop, K_main = [t.cpu() for t in eng.pose_prepare(T.cuda(), K.cuda(), mesh_ids.cuda(), pts_s.cuda(), 2000, 200, V, code,
                                                                                     (480, 640), (240, 320), with_K_main=True)]
    Tn = og.normalize_T(T)
    tcr = Tn[:, :3, 3]
    TV = og.make_TCO_multiview(Tn, tcr, mvt, V, remove_TCO_rendering=remove, views_inplane_rotations=inplane)
    assert TV.shape == TCV.shape and (TCV - TV).abs().max() < 3e-6
    P = pts_s[mesh_ids.long()]
    br = og.boxes_from_uv(og.project_points_robust(P, K, Tn))
    bc = og.crop_boxes_robust(br, K, Tn, tcr, P, (480, 640))
    Kc = og.get_K_crop_resize(K, bc, (240, 320))
    assert (brend - br).abs().max() < 2e-3 and (bcrop - bc).abs().max() < 2e-3
    assert ((K_main - Kc).abs() / Kc.abs().clamp(min=1.0)).max() < 1e-4
    Pv = P[:, :200].unsqueeze(1).repeat(1, V, 1, 1).flatten(0, 1)
    TVf, Kf = TV.flatten(0, 1), K.unsqueeze(1).repeat(1, V, 1, 1).flatten(0, 1)
    bcv = og.crop_boxes_robust(og.boxes_from_uv(og.project_points_robust(Pv, Kf, TVf)), Kf, TVf, TVf[:, :3, 3], Pv, (480, 640))
    Kcv = og.get_K_crop_resize(Kf, bcv, (240, 320)).view(b, V, 3, 3)
    if not remove:
        Kcv[:, 0] = Kc
    assert ((KV - Kcv).abs() / Kcv.abs().clamp(min=1.0)).max() < 2e-4


@pytest.mark.parametrize("mvt,remove", [("TCO+front_1view", False), ("TCO+front_3views", True), ("sphere_26views", False)])
def test_pose_predictor_forward_other_multiview_modes_vs_oracle(object_dataset, mvt, remove):
    """PosePredictor.forward with the view lists the released recipes do not use (2 / 3 / 27 rendered views; the 27-view input needs
    several rasteriser launches per step): raw network output and updated pose against the oracle predictor, 2 iterations."""
    from types import SimpleNamespace

    from megapose6d_amd import engine as eng
    from megapose6d_amd.load_model import build_pose_model
    from megapose6d_amd.mesh_db import MeshDataBase
    from megapose6d_amd.renderer import Panda3dBatchRenderer
    from oracle import pipeline as op
    from oracle import raster as orr
    from tests.support import synthetic as syn

    code = eng.MV_MODES[mvt] | (eng.MV_REMOVE_TCO if remove else 0)
    V = eng.multiview_n_views(code)
    cfg = syn.make_cfg("refiner")
    cfg.multiview_type, cfg.n_rendered_views, cfg.remove_TCO_rendering = mvt, V, remove
    sd = syn.make_state_dict("vanilla_resnet34", syn.n_inputs_for(cfg), "pose", 9, seed=21)
    renderer = Panda3dBatchRenderer(object_dataset, n_workers=1, preload_cache=True)
    from oracle import mesh_loader
    meshes, db = mesh_loader.load_dataset(object_dataset)        # the oracle's inputs through the oracle's own reader
    model = build_pose_model(cfg, sd, renderer, MeshDataBase.from_object_ds(object_dataset).batched().cuda())
    labels = [object_dataset[0].label, object_dataset[1].label]
    rng = np.random.RandomState(4)
    T0 = torch.from_numpy(np.stack([syn.random_pose(rng, (0.45, 0.6), 0.1) for _ in labels]))
    K = torch.from_numpy(np.repeat(syn.K_EXAMPLE[None], 2, 0)).float()
    images = torch.rand(2, 3, 480, 640, generator=torch.Generator().manual_seed(2))
    images = torch.round(images * 255) / 255
    opred = op.OraclePosePredictor(cfg, sd, db.labels.tolist(), db.points, orr.OracleBatchRenderer(meshes))
    ref = opred.forward(images, torch.arange(2), K, labels, T0, 2)
    got = model(images=images.cuda(), K=K.cuda(), labels=labels, TCO=T0.cuda(), n_iterations=2)
    for n in range(2):
        o = got[f"iteration={n + 1}"]
        assert (o.network_outputs["pose"].cpu() - ref[n]["net"]["pose"]).abs().max().item() < 1e-4, n
        assert (o.TCO_output.cpu() - ref[n]["TCO_output"]).abs().max().item() < 1e-4, n
        assert (o.KV_crop.cpu() - ref[n]["KV_crop"]).abs().div(ref[n]["KV_crop"].abs().clamp(min=1)).max().item() < 2e-4


def test_pose_ops_vs_oracle(eng, engine_meshes):
    from tests.support import synthetic as syn
    from oracle import geometry as og

    rng = np.random.RandomState(11)
    b = 7
    pts_list = [torch.from_numpy(m["points"]) for m in engine_meshes]
    pts = og.pad_stack_points(pts_list)  # [3, Nmax, 3]
    ids2000 = og.sample_point_ids(pts.shape[1], 2000)
    pts_s = pts[:, ids2000]
    mesh_ids = torch.tensor(rng.randint(0, 3, size=b), dtype=torch.int32)
    T = torch.from_numpy(np.stack([syn.random_pose(rng) for _ in range(b)]))
    T[:, :3, :3] += 0.01 * torch.randn(b, 3, 3, generator=torch.Generator().manual_seed(77))  # not orthonormal on purpose (seeded: independent of test order)
    K = torch.from_numpy(np.repeat(syn.K_EXAMPLE[None], b, 0)).float()
    # normalize_T
    got = eng.normalize_T(T.cuda()).cpu()
    assert (got - og.normalize_T(T)).abs().max() < 1e-6
    # prepare, single view
    for V, mv, mvt in ((1, 0, "TCO"), (4, 1, "TCO+front_3views")):
        TCO_n, tCR, TCV, KV, brend, bcrop = [t.cpu() for t in eng.pose_prepare(T.cuda(), K.cuda(), mesh_ids.cuda(), pts_s.cuda(), 2000, 200, V, mv, (480, 640), (240, 320))]
        Tn = og.normalize_T(T)
        tcr = Tn[:, :3, 3]
        P = pts_s[mesh_ids.long()]
        uv = og.project_points_robust(P, K, Tn)
        br = og.boxes_from_uv(uv)
        bc = og.crop_boxes_robust(br, K, Tn, tcr, P, (480, 640))
        Kc = og.get_K_crop_resize(K, bc, (240, 320))
        assert (TCO_n - Tn).abs().max() < 1e-6 and (tCR - tcr).abs().max() < 1e-6
        assert (brend - br).abs().max() < 2e-3 and (bcrop - bc).abs().max() < 2e-3  # pixels, values ~ 1e2..1e3
        assert ((KV[:, 0] - Kc).abs() / Kc.abs().clamp(min=1.0)).max() < 1e-4   # f / box-size: the fp32 round-off of the box is amplified
        TV = og.make_TCO_multiview(Tn, tcr, mvt, V)
        assert (TCV - TV).abs().max() < 2e-6
        if V == 4:
            Pv = P[:, :200].unsqueeze(1).repeat(1, V, 1, 1).flatten(0, 1)
            TVf, Kf = TV.flatten(0, 1), K.unsqueeze(1).repeat(1, V, 1, 1).flatten(0, 1)
            brv = og.boxes_from_uv(og.project_points_robust(Pv, Kf, TVf))
            bcv = og.crop_boxes_robust(brv, Kf, TVf, TVf[:, :3, 3], Pv, (480, 640))
            Kcv = og.get_K_crop_resize(Kf, bcv, (240, 320)).view(b, V, 3, 3)
            Kcv[:, 0] = Kc
            assert ((KV - Kcv).abs() / Kcv.abs().clamp(min=1.0)).max() < 1e-4
    # pose update
    out9 = torch.randn(b, 9) * 0.05 + torch.tensor([1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0])
    Tn = og.normalize_T(T)
    got = eng.pose_update(Tn.cuda(), Kc.cuda().contiguous(), out9.cuda(), Tn[:, :3, 3].contiguous().cuda()).cpu()
    assert (got - og.update_pose(Tn, Kc, out9, Tn[:, :3, 3])).abs().max() < 1e-6
    # SO(3)-grid init
    quats = torch.randn(16, 4); quats = quats / quats.norm(dim=1, keepdim=True)
    R = og.load_SO3_grid_from_quats(quats)
    ext = eng.init_extents(pts.cuda(), R.cuda())
    rot_ids = torch.tensor(rng.randint(0, 16, size=b), dtype=torch.int32)
    boxes = torch.tensor([[200.0, 150, 330, 300]]).repeat(b, 1) + torch.rand(b, 4) * 20
    got = eng.init_poses_from_boxes(boxes.cuda(), K.cuda(), mesh_ids.cuda(), rot_ids.cuda(), R.cuda(), ext).cpu()
    ref = og.TCO_init_from_boxes_autodepth_with_R(boxes, pts[mesh_ids.long()], K, R[rot_ids.long()])
    assert (got - ref).abs().max() < 1e-5


@pytest.mark.parametrize("shape", [(64, 64, 60, 80), (128, 128, 30, 40), (256, 256, 15, 20), (128, 256, 15, 20)])
def test_conv_full_rounds_plus_splitk_tail(eng, shape):
    """a grid of ~600 tiles on 512 resident workgroups: the first 512 tiles run single-pass, the ~88 tail tiles split K and are
    reduced deterministically ("mode 2" of plan_conv); checked against torch fp32 for the Cout = 64 tile AND the 128x128 tile
    (Cout = 128 / 256: the launch mode the 576-row layer-2/3 convs of BASELINE config 2 take)"""
    n_cu = eng.device_info()[0]
    g = torch.Generator().manual_seed(5)
    Cin, Cout, H, W = shape
    n_nb = max(1, Cout // 128)
    N = -(-((2 * n_cu + 88) // n_nb) * 128 // (H * W))      # enough rows for 2*n_cu + ~88 tiles of 128 pixels
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    res = torch.randn(N, Cout, H, W, generator=g)
    bias = torch.randn(Cout, generator=g) * 0.1
    xb = _to_padded(eng, x, Cin, 1)
    rb = _to_padded(eng, res, Cout, 1)
    wp = torch.from_numpy(eng.conv_pack_weights(w.numpy(), Cin, None)).cuda()
    ws = torch.empty(24 << 20, device="cuda")
    plan = eng.conv2d_plan(N, H, W, Cin, 1, Cout, 3, 1, 1, n_cu, ws.numel())
    assert plan["mode"] == 2 and plan["k_split"] >= 2, plan
    outs = []
    for scratch in (None, ws, ws):
        yb = eng.padded_nhwc(N, H, W, Cout, 1, "cuda")
        eng.profile_begin()
        eng.conv2d_nhwc(xb, N, H, W, Cin, 1, wp, bias.cuda(), Cout, 3, 1, 1, yb, 1, residual=rb, relu=True, splitk_ws=scratch)
        prof = eng.profile_end()
        outs.append((yb.clone(), prof))
    assert not any(k.endswith("/splitk") for k in outs[0][1])
    assert any(k.endswith("/splitk") for k in outs[1][1]) and "conv_splitk_reduce" in outs[1][1]
    assert torch.equal(outs[1][0], outs[2][0])                                   # deterministic
    ref = F.relu(_conv_ref_f64(x, w, None, bias, 1, 1) + res)
    for yb, _ in outs[:2]:
        assert (_from_padded(eng, yb, N, H, W, Cout, 1) - ref).abs().max() < CONV_TOL * ref.abs().max()
    assert (outs[0][0] - outs[1][0]).abs().max() < CONV_TOL * ref.abs().max()


@pytest.mark.parametrize("flags", [1, 17])
def test_raster_items_of_four_views_bit_exact_vs_oracle(eng, engine_meshes, oracle_meshes, flags):
    """One launch, 3 items x 4 views (the refiner's layout: a wave walks the 4 views of its item) against the oracle, view by view, and
    run to run.  Regression test: the per-view list headers used to be kept in per-lane registers and read back with v_readlane; a
    register spill under a partial exec mask inside the view loop lost them for the later views (a few wrong pixels, nondeterministic)."""
    from tests.support import synthetic as syn
    from oracle import raster as orr

    db = _mesh_db(eng, engine_meshes)
    rng = np.random.RandomState(7)
    n_items, V, h, w, Cp = 3, 4, 240, 320, 32
    Tn = np.stack([syn.random_pose(rng, z_range=(0.3, 0.6)) for _ in range(n_items * V)])
    Kn = np.repeat(syn.K_EXAMPLE[None].astype(np.float32), n_items * V, 0)
    Kn[:, :2] *= 0.5
    ids = torch.tensor([0, 1, 2], dtype=torch.int32).repeat_interleave(V).cuda()
    ref = [orr.render(oracle_meshes[v // V], Tn[v:v + 1], Kn[v:v + 1], h, w, flags | 1) for v in range(n_items * V)]
    outs = []
    for rep in range(3):
        x = torch.full((n_items, h, w, Cp), -3.0, device="cuda")
        eng.raster_render(db, ids, torch.from_numpy(Tn).cuda(), torch.from_numpy(Kn).cuda(), h, w, flags | 1, eng.make_lights(), x,
                          h * w * Cp, w * Cp, Cp, 3, 6, -1, views_per_item=V, stride_view=6)
        outs.append(x.cpu().numpy())
    for v in range(n_items * V):
        got = outs[0][v // V, :, :, 3 + 6 * (v % V): 9 + 6 * (v % V)]
        assert np.array_equal(got[..., :3], ref[v][0][0]), f"view {v}: rgb differs from the oracle"
        assert np.array_equal(got[..., 3:], ref[v][1][0]), f"view {v}: normals differ from the oracle"
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("C", [3, 4])
def test_fused_crop_in_raster_launch_equals_standalone_crop(eng, engine_meshes, C):
    """mp_raster_render_crop: the crop role of the band kernel writes exactly what mp_crop_roi_align writes, and leaves the views alone"""
    from tests.support import synthetic as syn

    db = _mesh_db(eng, engine_meshes)
    rng = np.random.RandomState(7)
    n_items, V, h, w, Cp = 3, 4, 240, 320, 32
    g = torch.Generator().manual_seed(0)
    images = torch.rand(2, C, 480, 640, generator=g).cuda()
    if C == 4:
        images[:, 3] = torch.where(images[:, 3] < 0.1, torch.zeros_like(images[:, 3]), images[:, 3])   # invalid depth pixels
    im_ids = torch.tensor([1, 0, 1], dtype=torch.int32).cuda()
    boxes = torch.tensor([[100.0, 80, 400, 305], [-40.0, -30, 300, 225], [500.0, 300, 700, 450]]).cuda()
    T = torch.from_numpy(np.stack([syn.random_pose(rng, z_range=(0.3, 0.6)) for _ in range(n_items * V)])).cuda()
    K = torch.from_numpy(np.repeat(syn.K_EXAMPLE[None].astype(np.float32), n_items * V, 0)).cuda()
    K[:, :2] *= 0.5
    ids = torch.tensor([0, 1, 2], dtype=torch.int32).repeat_interleave(V).cuda()
    nin = C
    outs = []
    for fused in (False, True):
        x = torch.full((n_items, h, w, Cp), -3.0, device="cuda")
        if not fused:
            eng.crop_roi_align(images, im_ids, boxes, h, w, x, h * w * Cp, w * Cp, Cp, 0)
        eng.raster_render(db, ids, T, K, h, w, 1, eng.make_lights(), x, h * w * Cp, w * Cp, Cp, nin, nin + 3, -1, views_per_item=V,
                          stride_view=6, crop=(images, im_ids, boxes, 0) if fused else None)
        outs.append(x)
    assert torch.equal(outs[0], outs[1])
    assert (outs[1][..., :nin] != -3.0).all() and (outs[1][..., nin + 6 * V:] == -3.0).all()   # crop written, padding untouched
    # NHWC4-packed observation (what the pipeline passes): same values
    x = torch.full((n_items, h, w, Cp), -3.0, device="cuda")
    eng.raster_render(db, ids, T, K, h, w, 1, eng.make_lights(), x, h * w * Cp, w * Cp, Cp, nin, nin + 3, -1, views_per_item=V,
                      stride_view=6, crop=(eng.PackedObservation(images), im_ids, boxes, 0))
    assert torch.equal(x, outs[0])


@pytest.mark.parametrize("mode", ["fp32_crop", "fp32_plain", "f16_crop", "records_rgb", "records_rgbd"])
def test_raster_compacted_launch_equals_the_direct_form(eng, engine_meshes, monkeypatch, mode):
    """Round 5: raster_classify marks the (item, tile) pairs no view reaches; raster_tiles' waves of those pairs leave after one byte load
    and raster_tiles_light writes them (background + crop).  Which kernel writes a tile must not change a bit: the same launch with
    MP_RASTER_COMPACT=0 (every pair through raster_tiles, the form of rounds 2-4) and in the compacted form, NaN / sentinel-poisoned
    outputs (every pixel of every written channel must be written by exactly the same values), for the fp32, binary16 and record outputs,
    with and without the fused crop, small objects (most tiles empty), an item whose views are ALL empty (non-finite pose) and an object
    filling its crop."""
    from tests.support import synthetic as syn

    db = _mesh_db(eng, engine_meshes)
    rng = np.random.RandomState(17)
    n_items, V, h, w = 4, 4, 240, 320
    rgbd = mode == "records_rgbd"
    C = 4 if rgbd else 3
    g = torch.Generator().manual_seed(3)
    images = torch.rand(2, C, 480, 640, generator=g).cuda()
    if rgbd:
        images[:, 3] = torch.where(images[:, 3] < 0.1, torch.zeros_like(images[:, 3]), 0.3 + images[:, 3])
    im_ids = torch.tensor([1, 0, 1, 0], dtype=torch.int32).cuda()
    boxes = torch.tensor([[100.0, 80, 400, 305], [-40.0, -30, 300, 225], [500.0, 300, 700, 450], [200.0, 100, 420, 265]]).cuda()
    T = np.stack([syn.random_pose(rng, z_range=(0.9, 1.6) if i // V != 3 else (0.2, 0.25)) for i in range(n_items * V)])   # far = small; item 3 fills the crop
    T[1 * V:2 * V, 0, 3] = np.nan                                                                                     # item 1: every view empty
    T = torch.from_numpy(T).cuda()
    K = torch.from_numpy(np.repeat(syn.K_EXAMPLE[None].astype(np.float32), n_items * V, 0)).cuda()
    K[:, :2] *= 0.5
    ids = torch.tensor([0, 1, 2, 0], dtype=torch.int32).repeat_interleave(V).cuda()
    tCR = torch.tensor([[0.0, 0.0, 1.2], [0.0, 0.0, 1.0], [0.1, 0.0, 0.9], [0.0, 0.0, 0.22]]).cuda()
    nper = 7 if rgbd else 6
    n_in = C + nper * V
    outs = {}
    for compact in ("0", "1"):
        monkeypatch.setenv("MP_RASTER_COMPACT", compact)
        if mode.startswith("records"):
            mask = (1 << C) - 1
            if rgbd:
                for v in range(V):
                    mask |= 1 << (C + 6 + nper * v)
            R = eng.xrec_elements(bin(mask).count("1"), n_in - bin(mask).count("1"))
            x = torch.full((n_items, h, w, R), 7.0, device="cuda", dtype=torch.bfloat16)
            eng.raster_render(db, ids, T, K, h, w, 1 | 16 | (2 if rgbd else 0), eng.make_lights(), x, h * w * R, w * R, R, C, C + 3, C + 6 if rgbd else -1,
                              views_per_item=V, stride_view=nper, crop=(eng.PackedObservation(images), im_ids, boxes, 0),
                              xrec=(mask, tCR, 2) if rgbd else None)
            assert not (x == 7.0).all(dim=-1).any()   # every pixel record written
        else:
            Cp = 32
            dt = torch.float16 if mode == "f16_crop" else torch.float32
            x = torch.full((n_items, h, w, Cp), float("nan"), device="cuda", dtype=dt)
            crop = (eng.PackedObservation(images), im_ids, boxes, 0) if mode != "fp32_plain" else None
            eng.raster_render(db, ids, T, K, h, w, 1 | 16, eng.make_lights(), x, h * w * Cp, w * Cp, Cp, 3, 6, -1, views_per_item=V, stride_view=6, crop=crop)
            c_first = 0 if crop is not None else 3
            assert torch.isfinite(x[..., c_first:3 + 6 * V].float()).all()     # every written channel of every pixel written
            assert torch.isnan(x[..., 3 + 6 * V:].float()).all()                # nothing else touched
            x = torch.nan_to_num(x.float(), nan=-5.0)
        outs[compact] = x
    torch.cuda.synchronize()
    assert torch.equal(outs["0"], outs["1"])
    if not mode.startswith("records"):
        img = outs["1"][..., 3:3 + 6 * V]
        assert (img[1] == 0).all() and (img[0] > 0).float().mean() < 0.1 and (img[3] > 0).float().mean() > 0.15   # empty item, small object, object filling its crop
