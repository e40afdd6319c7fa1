#!/usr/bin/env python
"""bench.py -- pose-hypotheses/sec of the hot path (render + coarse + refine + score) on N MI355X.

One "step" = one `PoseEstimator.run_inference_pipeline` call over synthetic 640x480 frame(s) resident in HBM.
Default workload = BASELINE.json configs[1]: megapose-1.0-RGB structure (coarse 9-ch + refiner 27-ch vanilla ResNet-34, fp32),
1 object x 576 SO(3)-grid hypotheses, ALL 576 refined for 5 iterations (n_pose_hypotheses=576), then re-scored:
576 coarse + 2880 refine + 576 score CNN rows and 12 672 rendered views per object.
`--config 3|4|5` run the other BASELINE.json configurations (their lines are informational; the driver reads config 2).

N > 1: `python bench.py --gpus N` re-launches itself through `python -m torch.distributed.run` (one rank per GPU, backend
"nccl" = RCCL) unless it already runs under torchrun (WORLD_SIZE set).  Weak scaling: config 2 puts one object x 576
hypotheses per GPU; rows are sharded rank::world and the packed logits/poses are all-gathered once per stage.  Beside it the line then
carries `strong_scaling_config4` (BASELINE configs[3]: 64 detections, K = 5, the same total work at every N), `rccl` (world size, 3
gathers per step, final poses bit-identical across ranks: asserted) and `per_rank` diagnostics.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     the dominant kernel (conv3x3_wino_bf16x9: the fused Winograd 3x3 convolution on the bf16 MFMA through exact operand
                  pieces, csrc/conv_wino_bf16.hip) measured live with HIP events on its launch stream: executed bf16 rate / 2.5 PFLOP/s
  "cpu_baseline": the oracle ("port" of the reference's CPU path) timed on this box's host cores on a bounded sample
  "parity":       the rows the cpu_baseline leg computed, compared with the same rows of the timed GPU call (N = 1).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

# dmabuf IPC: RCCL across processes needs it on this driver, and the HSA runtime reads it when it starts (first GPU call of the process)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

METRIC = json.loads((ROOT / "BASELINE.json").read_text())["metric"] if (ROOT / "BASELINE.json").is_file() else \
    "pose-hypotheses/sec (render+coarse+refine), 640×480, megapose-1.0-RGB"
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0
N_HYP, N_ITERS = 576, 5
PARITY_TOL = 1e-4  # BASELINE.json north_star: poses within 1e-4 (absolute); logits 1e-4 x max(1, |logit|)


def _best_cpu_threads() -> int:
    """thread count of the committed host sweep (profiles/r02_cpu_thread_sweep.json) if there is one for this core count"""
    f = ROOT / "profiles" / "r02_cpu_thread_sweep.json"
    try:
        j = json.loads(f.read_text())
        if j.get("host_cores") == os.cpu_count():
            return int(j["best_threads"])
    except Exception:
        pass
    return min(os.cpu_count() or 1, 32)


def cpu_baseline(ds, obs_images: torch.Tensor, K: torch.Tensor, bboxes: torch.Tensor, budget_s: float = 20.0, threads: int = 0,
                 n_max: int = 48) -> dict:
    """Oracle (port of the reference CPU path: reference orchestration + torch-CPU fp32 CNN + C software rasteriser standing in
    for Panda3D, which cannot be installed offline) on a bounded sample of the config-2 workload: the first `n_coarse` coarse rows
    and the 5-iteration refiner chains + re-score of the first `n_refine` hypotheses of object 0.  The sample is sized from a
    4-row probe so that it stays within `budget_s` seconds of CPU work.  Returns the timing AND the computed values (the parity
    leg compares them with the GPU's rows)."""
    from oracle import geometry as og
    from oracle import harness

    threads = threads or _best_cpu_threads()
    torch.set_num_threads(threads)
    oest, db = harness.make_oracle_estimator(ds, N_HYP)
    cpred, rpred = oest.coarse, oest.refiner
    grid = oest.grid
    images, Kc = obs_images.cpu(), K.cpu()
    label = ds[0].label
    with torch.no_grad():
        T = og.TCO_init_from_boxes_autodepth_with_R(bboxes[:1].cpu().float().repeat(n_max, 1), db.points[:1].repeat(n_max, 1, 1),
                                                   Kc[:1].repeat(n_max, 1, 1), grid[:n_max])
        im = torch.zeros(n_max, dtype=torch.long)

        def coarse(n, poses):
            t0 = time.perf_counter()
            o = cpred.forward_coarse(images, im[:n], Kc[:1].repeat(n, 1, 1), [label] * n, poses[:n])
            return time.perf_counter() - t0, o["logits"].flatten(), o["net"]["features"].abs().max().item()

        coarse(2, T)  # warm-up
        t_row = coarse(4, T)[0] / 4  # probe
        n_coarse = int(max(4, min(n_max, 0.35 * budget_s / max(t_row, 1e-4))))
        n_refine = int(max(1, min(8, 0.65 * budget_s / max(t_row * 4.5 * (N_ITERS + 1), 1e-4))))
        t_coarse, coarse_logits, fs1 = coarse(n_coarse, T)
        t0 = time.perf_counter()
        outs = rpred.forward(images, im[:n_refine], Kc[:1].repeat(n_refine, 1, 1), [label] * n_refine, T[:n_refine], N_ITERS)
        t_refine = time.perf_counter() - t0
        t_score, score_logits, fs2 = coarse(n_refine, outs[-1]["TCO_output"])
    t_full = t_coarse * (N_HYP / n_coarse) + (t_refine + t_score) * (N_HYP / n_refine)
    return {"value": N_HYP / t_full, "unit": "pose-hypotheses/s", "cores": threads, "kind": "port",
            "sample": f"{n_coarse} of 576 coarse rows + {n_refine} of 576 hypotheses x {N_ITERS} refine iters + {n_refine} score rows, "
                      f"{t_coarse + t_refine + t_score:.1f} s of CPU work on {threads} threads ({os.cpu_count()} host cores), extrapolated "
                      "linearly per stage; Panda3D replaced by the oracle's C rasteriser",
            "_values": {"coarse_TCO": T[:n_coarse], "coarse_logits": coarse_logits, "refine_poses": [o["TCO_output"] for o in outs],
                        "refine_pose_out": [o["net"]["pose"] for o in outs], "score_logits": score_logits, "feature_max": max(fs1, fs2)}}


def parity_block(vals: dict, extra: dict) -> dict:
    """oracle rows of the cpu_baseline leg vs the SAME rows of the timed 576-row GPU call (object 0: hypotheses 0..n-1)"""
    cd = extra["coarse"]
    n_c = vals["coarse_logits"].numel()
    lo = vals["coarse_logits"]
    scale = max(1.0, lo.abs().max().item())  # the seeded nets' features are O(1) (tests/support/synthetic.py): no feature-scale factor
    out = {"rows": {"coarse": n_c, "refine_chains": len(vals["score_logits"]), "iterations": len(vals["refine_poses"])},
           "tolerance": PARITY_TOL, "logit_scale": scale, "feature_max": float(vals.get("feature_max", 0.0))}
    out["coarse_TCO_max_err"] = (cd["preds"].poses[:n_c].cpu() - vals["coarse_TCO"]).abs().max().item()
    ce = (cd["data"]["logits"].flatten()[:n_c].cpu() - lo).abs()
    out["coarse_logit_max_err"] = ce.max().item()
    # the GPU call refined the hypotheses in top-K order: find hypotheses 0..n-1 of detection 0 in its filtered table
    dff = extra["coarse_filter"]["preds"].infos.reset_index(drop=True)
    n_r = len(vals["score_logits"])
    det0 = dff["bbox_id"].values == dff["bbox_id"].values.min()
    pos = [int(np.nonzero((dff["hypothesis_id"].values == h) & det0)[0][0]) for h in range(n_r)]
    preds = extra["refiner_all_hypotheses"]["preds"]
    pouts = extra["refiner_all_hypotheses"]["data"]["pose_outputs"]
    out["pose_max_err_per_iter"] = [(preds[f"iteration={n + 1}"].poses[pos].cpu() - vals["refine_poses"][n]).abs().max().item()
                                    for n in range(len(vals["refine_poses"]))]
    out["pose_out_max_err_per_iter"] = [(pouts[f"iteration={n + 1}"][pos].cpu() - vals["refine_pose_out"][n]).abs().max().item()
                                        for n in range(len(vals["refine_pose_out"]))]
    sl = vals["score_logits"]
    scale = max(scale, sl.abs().max().item())
    out["logit_scale"] = scale
    se = (extra["scoring"]["data"]["logits"].flatten()[pos].cpu() - sl).abs()
    out["score_logit_max_err"] = se.max().item()
    # poses: 1e-4 absolute.  logits: oracle.harness.logit_flip_rule -- every row within 1e-4 x scale except COUNTED flipped-silhouette rows
    # (at most one per 64 rows, each within 2e-4 x scale); the counts are part of the line
    from oracle.harness import logit_flip_rule

    rc_, rs_ = logit_flip_rule(ce.numpy(), scale, PARITY_TOL), logit_flip_rule(se.numpy(), scale, PARITY_TOL)
    out["logit_rule"] = "every row < tol*scale, except <= 1 row per 64 (counted) < 2*tol*scale; the score logits are compared CHAINED (each side scores its own final pose) under this strict rule"
    out["coarse_logit_rows_over_tol"], out["score_logit_rows_over_tol"] = rc_["rows_over_tol"], rs_["rows_over_tol"]
    out["ok"] = bool(out["coarse_TCO_max_err"] < PARITY_TOL and rc_["ok"] and rs_["ok"] and all(e < PARITY_TOL for e in out["pose_max_err_per_iter"]))
    return out


def extras(est, obs, det, steps: int) -> dict:
    """Secondary numbers (NOT `value`): the released inference parameters (n_pose_hypotheses = 1 / 5) and the
    other modes on the headline workload.  Same timing discipline, 1 warm-up + `steps` timed calls."""

    def timed(k_hyp: int) -> float:
        est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=N_ITERS, n_pose_hypotheses=k_hyp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=N_ITERS, n_pose_hypotheses=k_hyp)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    out = {}

    def guarded(name: str, fn) -> None:
        # an optional mode that fails must not take the headline line down with it: record the error under its own key
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.synchronize()

    def hyp_line(note: str, k_hyp: int = N_HYP) -> dict:
        dt = timed(k_hyp)
        return {"ms_per_step": dt * 1e3, "pose_hypotheses_per_s": N_HYP / dt, "note": note}

    for k in (1, 5):
        def released(k=k):
            dt = timed(k)
            return {"ms_per_call": dt * 1e3, "coarse_hypotheses_per_s": N_HYP / dt,
                    "note": "576 coarse rows + K x 5 refine rows + K score rows (megapose-1.0-RGB[-multi-hypothesis] defaults)"}
        guarded(f"n_pose_hypotheses={k}", released)
    guarded("without_event_profiler", lambda: hyp_line(
        "the `value` workload without the per-launch HIP events the roofline figures need (two event records per kernel launch inside "
        "the timed region cost the difference)"))
    rend = est.coarse_model.renderer   # one renderer serves both models

    def single_sample():
        rend.msaa = 1
        try:
            return hyp_line("Panda3dBatchRenderer(msaa=1): one centre sample per pixel instead of the reference's 4x multisampling "
                            "(panda3d_scene_renderer.py:73-74); what the rasteriser's multisampling costs end to end; never `value`")
        finally:
            rend.msaa = 4
    guarded("single_sample_renders", single_sample)

    def streams(n):
        def run():
            est.n_streams = n  # chunk interleave on n HIP streams: fills the tails of the conv grids (a 576-row Winograd launch is 5.6 .. 42.2 rounds
            try:               # of 256 workgroups) and overlaps raster with MFMA work; not the default because overlapping kernels distort the per-kernel
                               # event timing the roofline figures rest on (each kernel's duration then includes the other's share of the chip)
                return hyp_line(f"same fp32 path, PoseEstimator.n_streams={n} (MP_N_STREAMS); not used for `value` (round 6, same box: 260.7 / 262.5 "
                                "ms on one stream, 255.5 / 257.2 on two, 256.6 on three)")
            finally:
                est.n_streams = 1
        return run
    guarded("two_stream_interleave", streams(2))
    guarded("three_stream_interleave", streams(3))

    def fp16_renders():
        est.render_dtype = torch.float16
        try:
            return hyp_line("optional mode (BASELINE.json configs[4] \"fp16 renders\"): the rasteriser launch stores the CNN input (renders + "
                            "observation crop) as binary16 (MP_RASTER_F16), the stem convolution widens it on its way into LDS "
                            "(mp_backbone_forward_f16); narrower than the reference's fp32 input -- never `value`")
        finally:
            est.render_dtype = torch.float32
    guarded("fp16_renders", fp16_renders)
    def detector_line():
        # the 2D detector front-end: Mask R-CNN ResNet-50-FPN, one 640x480 frame, 22 classes (YCB-V sized head),
        # random weights of that architecture; NOT part of `value` (detections are supplied to the pose pipeline, as BASELINE.json defines)
        from megapose6d_amd import engine as eng_
        from megapose6d_amd.mask_rcnn import DetectorMaskRCNN

        g = torch.Generator().manual_seed(0)
        sd = {}
        for name, shape in eng_.DetectorNet.state_spec(22):
            t = torch.rand(shape, generator=g) * 2 - 1
            if name.endswith("running_var") or ((".bn" in name or "downsample.1." in name) and name.endswith(".weight")):
                t = t * 0.5 + 1.0
            elif name.endswith(".weight"):
                t = t * (3.0 / (t.numel() // shape[0])) ** 0.5
            else:
                t = t * 0.1
            sd[name] = t
        m = DetectorMaskRCNN(input_resize=(480, 640), n_classes=22)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        frame = [obs.images[0, :3]]
        m(frame)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out_ = m(frame)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        m._engine.close()
        return {"ms_per_frame": dt * 1e3, "detections": int(len(out_[0]["boxes"])),
                "note": "Detector front-end: DetectorMaskRCNN (ResNet-50 + FPN Mask R-CNN, 22 classes, masks included) on one 640x480 frame, "
                        "one native call per frame (mp_detector_forward) + the host read of the detection count; informational"}
    guarded("detector_front_end", detector_line)
    def other_workload(cfg_id: int, backbone: str, k_hyp: int, note: str):
        """a second, driver-visible line on another BASELINE configuration / backbone: own estimator, 1 warm-up + 2 timed calls, with the
        dominant conv kernel's rate from the in-library event profiler (never `value`)"""
        import shutil
        import tempfile

        from megapose6d_amd import engine as eng_

        tmp2 = tempfile.mkdtemp(prefix="mp_bench_extra_")
        try:
            est2, obs2, det2, _, desc2, n_obj2, run2 = build_workload(cfg_id, 1, backbone, tmp2, k_hyp)
            est2.run_inference_pipeline(obs2, detections=det2, **run2)
            torch.cuda.synchronize()
            eng_.profile_begin()
            n_t = 2
            t0 = time.perf_counter()
            for _ in range(n_t):
                est2.run_inference_pipeline(obs2, detections=det2, **run2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_t
            prof2 = eng_.profile_end()
            conv2 = {k: v for k, v in prof2.items() if k.startswith("conv") and k != "conv_splitk_reduce"}
            dom = max(conv2, key=lambda k: conv2[k]["ms"])
            rast = sum(v["ms"] for k, v in prof2.items() if k.startswith("raster_")) / n_t
            line = {"workload": desc2, "ms_per_call": dt * 1e3, "pose_hypotheses_per_s": n_obj2 * N_HYP / dt, "dominant_conv_kernel": dom,
                    "dominant_conv_tflops_algorithmic": conv2[dom]["flops"] / (conv2[dom]["ms"] * 1e-3) / 1e12,
                    "dominant_conv_mfma_utilisation": conv2[dom]["executed"] / (conv2[dom]["ms"] * 1e-3) / 1e12 / conv2[dom]["peak_tflops"],
                    "all_conv_kernels_tflops_algorithmic": sum(v["flops"] for v in conv2.values()) / (sum(v["ms"] for v in conv2.values()) * 1e-3) / 1e12,
                    "raster_ms_per_call": rast, "note": note}
            del est2, obs2, det2
            torch.cuda.empty_cache()
            return line
        finally:
            shutil.rmtree(tmp2, ignore_errors=True)

    def activation_statistics():
        # the seeded networks' post-ReLU activations are about half zeros; a trained network's are what they are.  bf16 MFMA work is
        # power-limited, so the clock -- and with it every rate of this line -- depends on how much the operands toggle.  This entry makes
        # that visible to the driver: one 576-row refiner backbone forward on the CNN input of the last step, (a) with the bench's network,
        # (b) with every BatchNorm shift raised by 0.75 (dense activations: few zeros, N(0,1)-like operand statistics).
        from megapose6d_amd import engine as eng_

        m = est.refiner_model
        x = m._x[0]
        h, w = m.render_size
        rows = min(576, m._x_rows[0])
        res = {}
        for name, shift in (("bench_network", 0.0), ("dense_activations", 0.75)):
            sd = {k: (v.clone() + shift if (".bn" in k or k.endswith("bn1.bias") or "downsample.1" in k) and k.endswith(".bias") else v)
                  for k, v in m.state_dict().items()}
            bb = eng_.Backbone(m.backbone.backbone_str, m.backbone.n_inputs, "pose", 9, sd)
            out_ = torch.empty(rows, 9, device="cuda")
            kw = dict(n_f32=3) if x.dtype == torch.bfloat16 else {}
            bb.forward(x, rows, h, w, out_, None, **kw)
            torch.cuda.synchronize()
            eng_.conv_wino_bf16_clock(reset=True)
            t0 = time.perf_counter()
            for _ in range(3):
                bb.forward(x, rows, h, w, out_, None, **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            mhz, cps = eng_.conv_wino_bf16_clock(reset=True)
            res[name] = {"ms_per_576_row_forward": dt * 1e3, "bf16_winograd_k_loop_mhz": mhz, "k_loop_cycles_per_step": cps,
                         "finite": bool(torch.isfinite(out_).all())}
            bb.close()
        res["note"] = ("conv stack of one refiner forward under two operand statistics; the ratio says how much slower a network with dense "
                       "activations runs on the same kernels (clock give-back under bf16 MFMA load); informational")
        return res
    guarded("conv_stack_activation_statistics", activation_statistics)
    guarded("released_recipe_64_detections_K5", lambda: other_workload(
        4, "vanilla_resnet34", 5, "BASELINE configs[3] on one GPU with the RELEASED inference parameters (n_pose_hypotheses = 5, utils/load_model.py:31-34): "
        "64 detections over 8 frames, 36 864 coarse rows + 320 x 5 refiner rows + 320 score rows per call"))
    # the 8-GPU claim, checkable on one GPU (north_star: ">= 6x at 8 GPUs"): one rank's share of BASELINE configs[3] (64 detections,
    # released K = 5: strong scaling) and of the weak-scaled `value` workload
    guarded("emulated_rank_of_8_config4_K5", lambda: emulate_rank_share(4, 8, "vanilla_resnet34", 5, 2))
    guarded("emulated_rank_of_8_config2_weak", lambda: emulate_rank_share(2, 8, "vanilla_resnet34", N_HYP, 1))
    guarded("wide_resnet34_backbone", lambda: other_workload(
        2, "resnet34", N_HYP, "the `value` workload on the OTHER backbone the reference can ship (backbone_str 'resnet34' = WideResNet-34, "
        "training/pose_models_cfg.py:110-111; the released config.yaml is not available offline to tell which one it is)"))
    return out


def emulate_rank_share(cfg_id: int, world: int, backbone: str, k_hyp: int, steps: int = 2, rank: int = 0) -> dict:
    """`--emulate-rank-of N` (never `value`): what ONE rank of an N-GPU run does, measured on this one GPU.

    The N-GPU workload of configuration `cfg_id` is built exactly as `bench.py --gpus N` builds it (config 2: weak-scaled, N objects;
    configs 4 / 5: the fixed 64 detections = strong scaling), and run twice on this GPU: (a) whole, undistributed -- the time ONE GPU
    needs for all of it; (b) as rank `rank` of N through megapose6d_amd.distributed.emulate (rows rank::N of every stage table, the three
    all-gathers replaced by local copies).  T(a) / T(b) projects the speed-up of N GPUs over one on that workload, minus the RCCL time
    (3 latency-bound all-gathers per call) and inter-rank skew.  The per-stage split (HIP events, one extra fenced call each) says which
    stage keeps a rank busy -- SURVEY.md 8e expects the refiner's small per-rank batch at K = 5 (320 rows / 8 = 40) to be the limiter."""
    import shutil

    from megapose6d_amd import distributed as mpd_
    from megapose6d_amd import engine as eng_

    tmp2 = tempfile.mkdtemp(prefix="mp_bench_emul_")
    try:
        est2, obs2, det2, _, desc2, n_obj2, run2 = build_workload(cfg_id, world, backbone, tmp2, k_hyp)
        est2.distributed = True   # (build_workload sets it for world > 1; without a process group it is inert until emulate() is on)

        def timed():
            est2.run_inference_pipeline(obs2, detections=det2, **run2)
            torch.cuda.synchronize()
            eng_.profile_begin()
            t0 = time.perf_counter()
            for _ in range(steps):
                est2.run_inference_pipeline(obs2, detections=det2, **run2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            prof_ = eng_.profile_end()
            _, ex = est2.run_inference_pipeline(obs2, detections=det2, cuda_timer=True, **run2)
            stages = {s_: ex[s_]["data"]["time"] * 1e3 for s_ in ("coarse", "refiner", "scoring")}
            top = {k: round(v["ms"] / steps, 3) for k, v in sorted(prof_.items(), key=lambda kv: -kv[1]["ms"])[:6]}
            return dt * 1e3, stages, top

        mpd_.emulate(None)
        full_ms, full_st, full_top = timed()
        mpd_.emulate(rank, world)
        mpd_.stats.reset()
        try:
            share_ms, share_st, share_top = timed()
            gathers = mpd_.stats.calls / (steps + 2)
        finally:
            mpd_.emulate(None)
        rows_per_obj = N_HYP + k_hyp * N_ITERS + k_hyp
        line = {"workload": desc2, "emulated": f"rank {rank} of {world}", "objects": n_obj2,
                "one_gpu_whole_workload_ms": full_ms, "one_rank_share_ms": share_ms,
                "projected_speedup": full_ms / share_ms, "projected_efficiency": full_ms / share_ms / world,
                "projected_pose_hypotheses_per_s_at_N": n_obj2 * N_HYP / (share_ms * 1e-3),
                "rows_whole": n_obj2 * rows_per_obj, "rows_rank_share": -(-n_obj2 * rows_per_obj // world),
                "stage_ms_whole": full_st, "stage_ms_rank_share": share_st,
                "stage_speedup": {k: (full_st[k] / share_st[k] if share_st[k] > 0 else None) for k in full_st},
                "limiting_stage": min(full_st, key=lambda k: full_st[k] / max(share_st[k], 1e-9)),
                "top_kernels_ms_rank_share": share_top, "all_gathers_stubbed_per_call": gathers,
                "note": "projection from ONE GPU: one rank's share of the N-GPU call (rows rank::N, gathers replaced by local copies) against the "
                        "whole workload on one GPU; excludes RCCL time (3 all-gathers of <= 8.5 MB per call) and rank skew; never `value`"}
        del est2, obs2, det2
        torch.cuda.empty_cache()
        return line
    finally:
        shutil.rmtree(tmp2, ignore_errors=True)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _relaunch(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: spawn one rank per GPU through torch.distributed.run (RCCL rendezvous on 127.0.0.1)"""
    n_dev = torch.cuda.device_count()
    if n_dev < n:
        raise SystemExit(f"bench.py --gpus {n}: only {n_dev} GPU(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes needs it on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_workload(cfg_id: int, world: int, backbone: str, tmp: str, k_hyp: int):
    """-> (estimator, observation, detections, object dataset, description dict, n_objects, run kwargs)"""
    from tests.support.scene import make_multi_frame_scene, make_scene
    from tests.support import synthetic as syn

    common = dict(SO3_grid_size=N_HYP, tmp_dir=tmp, distributed=world > 1, n_streams=int(os.environ.get("MP_N_STREAMS", "1")))
    run = dict(n_refiner_iterations=N_ITERS, n_pose_hypotheses=k_hyp)
    if cfg_id == 2:
        n_obj = world  # weak scaling: one object x 576 hypotheses per GPU
        est, obs, det, _ = make_scene(n_objects=n_obj, seed=0, backbone=backbone, **common)
        ds = syn.make_object_dataset(tmp, n_objects=n_obj, seed=0)
        desc = f"configs[1]: megapose-1.0-RGB structure ({backbone} coarse 9ch + refiner 27ch), {n_obj} object(s) x 576 hypotheses"
    elif cfg_id == 3:
        n_obj = 8 * world
        est, obs, det, _ = make_scene(n_objects=n_obj, seed=7, backbone=backbone, rgbd=True, **common)
        ds = syn.make_object_dataset(tmp, n_objects=n_obj, seed=7)
        desc = f"configs[2]: megapose-1.0-RGBD structure ({backbone} coarse 9ch + RGBD refiner 32ch), {n_obj} objects x 576 hypotheses"
    elif cfg_id in (4, 5):
        n_obj = 64
        # config 5 = the "...-RGB-multi-hypothesis-icp" recipe: RGB coarse + refiner, the frames' depth channel only feeds the depth refiner
        est, obs, det, ds = make_multi_frame_scene(8, 8, 16, backbone=backbone, depth_obs=(cfg_id == 5), **common)
        desc = (f"configs[{cfg_id - 1}]: megapose-1.0-RGB-multi-hypothesis structure ({backbone}), 64 detections over 8 frames / 16 meshes x 576 "
                "hypotheses" + (", + depth refiner (ICP) on the frames' depth channel" if cfg_id == 5 else ""))
        if cfg_id == 5:
            from megapose6d_amd.icp_refiner import ICPRefiner

            est.depth_refiner = ICPRefiner(est.mesh_db, est.coarse_model.renderer)
            run["run_depth_refiner"] = True
    else:
        raise SystemExit(f"unknown --config {cfg_id}")
    desc += f" x {N_ITERS} refine iters (n_pose_hypotheses={k_hyp}) + re-score, 640x480 frames, 240x320 crops, 10k-triangle meshes"
    return est, obs, det, ds, desc, n_obj, run


def strong_scaling_companion(world: int, rank: int, backbone: str, steps: int) -> dict:
    """BASELINE.json configs[3] beside the weak-scaled headline line when N > 1 (never `value`): 64 detections over 8 frames x 576 hypotheses,
    the released K = 5, the SAME total work at every N (strong scaling) -- rows sharded rank::world, three all-gathers per call.  Called by
    every rank; rank 0 reports.  The driver computes efficiency itself from `value` of its runs at N = 1, 2, 4, 8: this block gives it the
    strong-scaled counterpart measured in the same launch (whole-job detections x 576 / time, max over ranks)."""
    from megapose6d_amd import distributed as mpd

    tmp = tempfile.mkdtemp(prefix=f"mp_bench_strong_r{rank}_")
    try:
        est, obs, det, ds, desc, n_obj, run = build_workload(4, world, backbone, tmp, 5)

        def fence():
            torch.cuda.synchronize()
            torch.distributed.barrier()
            torch.cuda.synchronize()

        est.run_inference_pipeline(obs, detections=det, **run)
        fence()
        mpd.stats.reset()
        mpd.stats.timing = True
        t0 = time.perf_counter()
        for _ in range(steps):
            final, _ = est.run_inference_pipeline(obs, detections=det, **run)
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device="cpu" if torch.distributed.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        assert len(final) == n_obj and torch.isfinite(final.poses).all()
        return {"workload": desc, "scaling": "strong", "n_gpus": world, "steps": steps, "ms_per_step": t.item() / steps * 1e3,
                "pose_hypotheses_per_s": n_obj * N_HYP * steps / t.item(), "refiner_rows_per_rank": (n_obj * 5 + world - 1) // world,
                "all_gathers_per_step": mpd.stats.calls / steps, "all_gather_ms_per_step": mpd.stats.ms() / steps,
                "host_topk_ms_per_step": mpd.stats.host_s * 1e3 / steps}
    except Exception as e:  # noqa: BLE001  (a companion measurement must never take the headline line down)
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5), help="BASELINE.json configuration (1-based, as SURVEY.md 8d numbers them); 2 = headline")
    ap.add_argument("--k-hyp", type=int, default=0, help="n_pose_hypotheses (default: 576 for configs 2/3, 5 for configs 4/5)")
    ap.add_argument("--backbone", default="vanilla_resnet34")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-thread-sweep", default="", help="comma list of thread counts: time the cpu_baseline sample at each, write "
                                                          "gpurun_out/cpu_thread_sweep.json and exit")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (faithful K=1/K=5 configs, other backbone / modes)")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="N",
                    help="one GPU only, never `value`: run ONE rank's share of the N-GPU workload of --config (rows rank::N, gathers stubbed) and "
                         "the whole workload, print projected_speedup = T(one GPU, whole) / T(rank share) with the per-stage split, and exit")
    ap.add_argument("--emulate-rank", type=int, default=0, help="which rank --emulate-rank-of emulates (default 0)")
    a = ap.parse_args()

    if a.emulate_rank_of:
        if a.gpus != 1 or "WORLD_SIZE" in os.environ:
            raise SystemExit("--emulate-rank-of is a single-process, single-GPU measurement")
        torch.cuda.set_device(0)
        k_e = a.k_hyp or (N_HYP if a.config in (2, 3) else 5)
        print(json.dumps({"emulate_rank_of": a.emulate_rank_of, "config": a.config,
                          **emulate_rank_share(a.config, a.emulate_rank_of, a.backbone, k_e, max(1, min(a.steps, 3)), a.emulate_rank)}))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_relaunch(a.gpus))

    from megapose6d_amd import distributed as mpd
    from megapose6d_amd import engine as eng

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # MP_BENCH_SHARED_GPU=1 (test rig only, never a measurement): all ranks share GPU 0 and talk over gloo -- exercises the world > 1
    # code path of this script (sharding, gathers, per-rank diagnostics) on a 1-GPU box
    shared_gpu = os.environ.get("MP_BENCH_SHARED_GPU", "0") == "1"
    torch.cuda.set_device(0 if shared_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        mpd.init_from_env("gloo" if shared_gpu else "nccl")
        mpd.stats.timing = True
    n_cu, lds, arch = eng.device_info()
    eng.conv_wino_bf16_telemetry(True)   # the K-loop clock / cycles-per-step figures of the roofline block (off in the product path)

    tmp = tempfile.mkdtemp(prefix=f"mp_bench_r{rank}_")
    k_hyp = a.k_hyp or (N_HYP if a.config in (2, 3) else 5)
    est, obs, det, ds, desc, n_obj, run = build_workload(a.config, world, a.backbone, tmp, k_hyp)

    if a.cpu_thread_sweep:
        res = {"host_cores": os.cpu_count(), "runs": []}
        for t in [int(v) for v in a.cpu_thread_sweep.split(",")]:
            r = cpu_baseline(ds, obs.images, obs.K, det.bboxes, budget_s=12.0, threads=t)
            r.pop("_values")
            res["runs"].append(r)
            print(f"[sweep] {t} threads: {r['value']:.3f} hyp/s", file=sys.stderr)
        best = max(res["runs"], key=lambda r: r["value"])
        res["best_threads"], res["best_value"] = best["cores"], best["value"]
        out_dir = ROOT / "gpurun_out"
        out_dir.mkdir(exist_ok=True)
        (out_dir / "cpu_thread_sweep.json").write_text(json.dumps(res, indent=1))
        print(json.dumps(res))
        return

    def step(**kw):
        return est.run_inference_pipeline(obs, detections=det, **run, **kw)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    mpd.stats.reset()
    mpd.stats.timing = world > 1
    eng.conv_clock(reset=True)
    eng.conv_wino_bf16_clock(reset=True)
    eng.conv_stem_bg_stats(reset=True)
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        final, extra = step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    stem_bg, stem_wgs = eng.conv_stem_bg_stats(reset=True)   # stem workgroups that took the background-tile walk / all, of the refiner-stem launches
    conv_mhz = eng.conv_clock(reset=True)   # shader clock INSIDE the (direct fp32) conv kernels of the timed steps
    wb_mhz, wb_cps = eng.conv_wino_bf16_clock(reset=True)   # ... and inside the K loops of the bf16x9 Winograd launches (+ cycles per step)
    gather_ms = mpd.stats.ms() if world > 1 else 0.0
    gather_calls, gather_bytes = mpd.stats.calls, mpd.stats.bytes   # (of the timed steps only: the extra stage-timing call below gathers too)
    host_topk_ms = mpd.stats.host_s * 1e3 / a.steps   # the replicated pandas top-K / arg-max of a step (does not shrink with the world size)
    dt_local = dt
    if world > 1:
        t = torch.tensor([dt], device="cpu" if torch.distributed.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    assert len(final) == n_obj and torch.isfinite(final.poses).all()
    poses_identical = None
    if world > 1:   # every rank must hold the SAME final table: identical gathered inputs -> identical top-K / arg-max -> identical poses, bit for bit
        assert torch.distributed.get_world_size() == a.gpus, (torch.distributed.get_world_size(), a.gpus)
        mine_p = final.poses.detach().to(torch.float32).contiguous()
        if torch.distributed.get_backend() == "gloo":
            mine_p = mine_p.cpu()
        all_p = [torch.empty_like(mine_p) for _ in range(world)]
        torch.distributed.all_gather(all_p, mine_p)
        poses_identical = bool(all(torch.equal(all_p[0], q) for q in all_p[1:]))
        assert poses_identical, "final poses differ between ranks"
    per_rank = None
    if world > 1:   # one diagnostic row per rank, so that the first multi-GPU run says where each rank's time went
        mine = {"rank": rank, "timed_region_ms_per_step": dt_local / a.steps * 1e3, "all_gather_ms_per_step": gather_ms / a.steps,
                "host_topk_ms_per_step": host_topk_ms,
                "kernel_ms_per_step": {k: round(v["ms"] / a.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:6]}}
        per_rank = [None] * world
        try:
            torch.distributed.all_gather_object(per_rank, mine)
        except Exception as e:  # noqa: BLE001  (diagnostics must never take the measurement down)
            per_rank = [{"error": f"{type(e).__name__}: {e}"[:200]}]
    # stage times from HIP events: one extra, untimed call with cuda_timer=True (each stage fenced, DEVICE render/model times)
    _, extra_t = step(cuda_timer=True)
    # sustained shader clock of THIS box under fp32-MFMA load (boxes of one pool differ by ~10 %): context for `roofline.frac`, not part of it
    clk = eng.clock_probe(30.0)

    strong = None
    if world > 1 and a.config == 2 and not a.no_extras:
        strong = strong_scaling_companion(world, rank, a.backbone, min(a.steps, 3))

    rc = 0
    if rank == 0:
        conv = {k: v for k, v in prof.items() if k.startswith("conv")}
        conv.pop("conv_splitk_reduce", None)
        dom_name = max(conv, key=lambda k: conv[k]["ms"])
        dom = conv[dom_name]
        # `frac` = the rate the dominant kernel EXECUTES on its matrix pipe / that pipe's dense peak (<= 1 by construction).  The algorithmic
        # (direct-convolution) rate sits next to it: Winograd executes 16/36 of the algorithmic multiplications, the
        # exact-piece kernels execute 9 (stem: 3 | 9) bf16 products per fp32 multiplication on the 16x faster bf16 pipe.
        achieved = dom["executed"] / (dom["ms"] * 1e-3) / 1e12
        alg_rate = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        per_kernel = {k: {"ms_per_step": round(v["ms"] / a.steps, 3), "algorithmic_tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                          "executed_tflops": round(v["executed"] / (v["ms"] * 1e-3) / 1e12, 2), "peak_tflops": v["peak_tflops"],
                          "mfma_utilisation": round(v["executed"] / (v["ms"] * 1e-3) / 1e12 / v["peak_tflops"], 4)}
                      for k, v in sorted(conv.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}
        all_conv_tf = sum(v["flops"] for v in conv.values()) / (sum(v["ms"] for v in conv.values()) * 1e-3) / 1e12
        kernel_ms = {k: round(v["ms"] / a.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        # the rasteriser's tile pass = raster_classify + raster_tiles (the pairs some view reaches) + raster_tiles_light (the others): one
        # row, the algorithmic bytes of the launch (attached to raster_tiles) over the summed time of the three kernels
        rparts = {k: v for k, v in prof.items() if k.startswith(("raster_tiles", "raster_classify"))}
        rb, rb_name = None, None
        if rparts:
            rb_name = max((k for k in rparts if not k.startswith(("raster_tiles_light", "raster_classify"))), key=lambda k: rparts[k]["ms"])
            rb = {"ms": sum(v["ms"] for v in rparts.values()), "bytes": sum(v["bytes"] for k, v in rparts.items() if k.startswith("raster_tiles")),
                  "launches": rparts[rb_name]["launches"], "parts_ms_per_step": {k: round(v["ms"] / a.steps, 3) for k, v in rparts.items()}}
        traffic, traffic_src, r_traffic = None, None, None
        tfile = next((f for f in (ROOT / "profiles" / "r06_traffic.json", ROOT / "profiles" / "r05_traffic.json", ROOT / "profiles" / "r04_traffic.json", ROOT / "profiles" / "r03_traffic.json")
                      if f.is_file()), None)
        if tfile is not None:  # PMC cannot be sampled from inside the process: committed rocprofv3 --pmc summary of the same command
            tj = json.loads(tfile.read_text())
            k = tj["kernels"].get(dom_name.replace(" ", ""))
            if k:
                traffic, traffic_src = k["hbm_bytes_per_launch_corrected"], tj["source"]
            k = tj["kernels"].get((rb_name or "").replace(" ", ""))
            if k:   # the tile pass = raster_tiles + (compacted form) raster_tiles_light + raster_classify, one launch each per raster call
                r_traffic = k["hbm_bytes_per_launch_corrected"] + sum(
                    tj["kernels"][n]["hbm_bytes_per_launch_corrected"] for n in ("raster_tiles_light", "raster_classify") if n in tj["kernels"])
        bf16 = dom["peak_tflops"] > 1000.0
        dom_mhz = wb_mhz if dom_name.startswith("conv3x3_wino_bf16") and wb_mhz > 0 else conv_mhz
        rows_per_obj = N_HYP + k_hyp * N_ITERS + k_hyp
        views_per_obj = N_HYP + 4 * k_hyp * N_ITERS + k_hyp
        sd = {s: extra_t[s]["data"] for s in ("coarse", "refiner", "scoring")}
        out = {
            "metric": METRIC, "value": n_obj * N_HYP * a.steps / dt, "unit": "pose-hypotheses/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (results of fp32 arithmetic; the convolutions' multiplications run on the bf16 MFMA as EXACT piece products of the fp32 operands, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": desc, "baseline_config": a.config, "objects": n_obj, "n_pose_hypotheses": k_hyp,
                       "rows_per_step": n_obj * rows_per_obj, "evals_per_s": n_obj * rows_per_obj * a.steps / dt,
                       "views_per_step": n_obj * views_per_obj,
                       "parallelism": f"rows sharded rank::world over {world} GPU(s)", "arch": arch, "cus": n_cu},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": dom["peak_tflops"], "unit": "TFLOP/s",
                         "frac": achieved / dom["peak_tflops"],
                         "frac_convention": "EXECUTED flops of the dominant kernel on its matrix pipe / that pipe's dense peak (MFMA utilisation, "
                                            "what north_star asks for); `frac_algorithmic` = SURVEY.md 8(d)'s algorithmic work (2 x MACs of the "
                                            "direct convolution) / the same time / the same peak -- the kernel executes 9 x 16/36 x tile padding "
                                            "= 4.1 x that",
                         "frac_algorithmic": alg_rate / dom["peak_tflops"],
                         "executed_arithmetic": ("bf16 MFMA on exact pieces of the fp32 operands (each fp32 value = three bf16 pieces, all nine piece "
                                                 "products, fp32 accumulate): every product exact" if bf16 else "fp32 MFMA"),
                         "algorithmic_equiv": {"tflops": alg_rate, "over_fp32_mfma_peak": alg_rate / PEAK_FP32_MFMA_TFLOPS,
                                               "note": "2 x MACs of the direct convolution (SURVEY.md 8d) / the same HIP-event time; may exceed the "
                                                       "fp32 matrix peak because the kernel does not execute those FLOPs"},
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src, "alg_bytes_per_launch": dom["bytes"] / dom["launches"], "launches": dom["launches"],
                         "avg_launch_ms": dom["ms"] / dom["launches"], "avg_launch_gflop_algorithmic": dom["flops"] / dom["launches"] / 1e9,
                         "avg_launch_gflop_executed": dom["executed"] / dom["launches"] / 1e9,
                         "all_conv_kernels_algorithmic_tflops": all_conv_tf, "per_kernel": per_kernel,
                         "shader_clock_mhz": dom_mhz,
                         "frac_at_measured_clock": achieved / (dom["peak_tflops"] * dom_mhz / 2400.0) if dom_mhz > 0 else None,
                         "k_loop_cycles_per_16_channel_step": wb_cps if dom_name.startswith("conv3x3_wino_bf16") else None,
                         "clock": {"bf16_winograd_k_loops_mhz": wb_mhz, "fp32_direct_kernels_mhz": conv_mhz, "probe_fp32_mfma_mhz": clk["shader_mhz"],
                                   "probe_fp32_mfma_tflops": clk["mfma_tflops"],
                                   "note": "s_memtime / s_memrealtime accumulated INSIDE the kernels of the timed steps (every 64th workgroup): the chip "
                                           "clocks to its power budget, and bf16 MFMA work on real operands runs well below the 2400 MHz `peak` is "
                                           "priced at; probe_* = a register-only fp32 MFMA loop right after the timed region (mp_clock_probe)"}},
            "raster": None if rb is None else {"bound": "hbm", "kernel": rb_name, "achieved": rb["bytes"] / (rb["ms"] * 1e-3) / 1e9,
                                               "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": rb["bytes"] / (rb["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                                               "traffic": r_traffic, "alg_bytes_per_launch": rb["bytes"] / rb["launches"],
                                               "avg_launch_ms": rb["ms"] / rb["launches"], "ms_per_step": rb["ms"] / a.steps,
                                               "parts_ms_per_step": rb["parts_ms_per_step"]},
            "kernel_ms_per_step": kernel_ms,
            "stage_s": {"source": "HIP events, one extra call with cuda_timer=True (stages fenced)",
                        **{s: {"time": sd[s]["time"], "render_time": sd[s]["render_time"], "model_time": sd[s]["model_time"]} for s in sd},
                        "total": extra_t["time"]},
        }
        if stem_wgs > 0:
            # K steps of the stem's slice walk (csrc/conv_stem.hip stem::n_steps): short walk over the dense walk, for this workload's
            # refiner stem (7x7, 3 fp32-kind + 24 integer channels: 5 record chunks, 2 of them hold the observation crop's pieces)
            n_steps = lambda ks, q: ((ks * ks * q + 3) // 4 + 1) // 2 * 2   # noqa: E731
            STEM_SHORT_WALK = n_steps(7, 2) / n_steps(7, 5)
            f_bg = stem_bg / stem_wgs
            out["stem_background"] = {
                "workgroups_short_walk_fraction": f_bg, "workgroups_per_step": stem_wgs / a.steps,
                "executed_flops_factor": 1.0 - f_bg * (1.0 - STEM_SHORT_WALK),
                "note": "refiner-stem workgroups (8 x 16 output pixels) whose input patch no rendered view reaches walk n_steps(7, 2) = 26 of the n_steps(7, 5) = 62 K steps "
                        "(only the observation crop's record chunks; the skipped products are exact zeros).  The per_kernel `executed_tflops` / "
                        "`mfma_utilisation` of that stem row are computed from the DENSE step count: multiply them by executed_flops_factor"}
        out["host"] = {"replicated_topk_ms_per_step": host_topk_ms,
                       "share_of_step": host_topk_ms / (dt / a.steps * 1e3),
                       "projected_share_at_8_gpus_same_total_work": host_topk_ms / (dt / a.steps * 1e3 / 8.0 + host_topk_ms * 7.0 / 8.0),
                       "note": "every rank repeats the pandas top-K / arg-max on the gathered table (DESIGN.md 5); if the projected share at 8 "
                               "GPUs exceeds 0.05 move it to rank 0 + broadcast (measured 0.044 in round 5: left replicated)"}
        if world > 1:
            out["per_rank"] = per_rank
            out["rccl"] = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                           "all_gathers_per_step": gather_calls / a.steps, "all_gather_bytes_per_step": gather_bytes / a.steps,
                           "all_gather_ms_per_step": gather_ms / a.steps}
            # one all-gather per stage (coarse | refiner, all iterations packed | scoring); config 5 adds none (ICP shards by object)
            assert abs(out["rccl"]["all_gathers_per_step"] - 3.0) < 1e-9, out["rccl"]
            out["rccl"]["final_poses_identical_across_ranks"] = poses_identical
        if strong is not None:
            out["strong_scaling_config4"] = strong
        if world == 1 and a.config == 2 and not a.no_extras:
            out["extras"] = extras(est, obs, det, min(a.steps, 3))   # (secondary lines: at most 3 timed calls each)
        if world == 1 and a.config == 5 and not a.no_extras:   # the "fp16 renders" variant BASELINE.json names for this configuration
            try:
                est.render_dtype = torch.float16
                step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                torch.cuda.synchronize()
                dt16 = (time.perf_counter() - t0) / a.steps
                out["extras"] = {"fp16_renders": {"ms_per_step": dt16 * 1e3, "pose_hypotheses_per_s": n_obj * N_HYP / dt16,
                                                  "note": "same workload with PoseEstimator.render_dtype = float16 (binary16 CNN input); never `value`"}}
            except Exception as e:  # noqa: BLE001
                out["extras"] = {"fp16_renders": {"error": f"{type(e).__name__}: {e}"[:300]}}
            finally:
                est.render_dtype = torch.float32
        if not a.no_cpu_baseline and world == 1 and a.config == 2 and k_hyp == N_HYP:
            cb = cpu_baseline(ds, obs.images, obs.K, det.bboxes)
            out["parity"] = parity_block(cb.pop("_values"), extra)
            # The port is the SLOWER stand-in: the reference's own orchestration (imported from /root/reference, same CNN / rasteriser) ran the
            # same rows 1.16-1.21x faster in the build container (profiles/r02_reference_vs_port_cpu.json; it cannot travel to the GPU box).
            # A slower baseline would inflate the ratio, so `vs_cpu_baseline` is quoted against the port's rate x 1.21.
            cb["reference_orchestration_factor"] = 1.21
            cb["value_reference_orchestration_estimate"] = cb["value"] * 1.21
            out["cpu_baseline"] = cb
            out["vs_cpu_baseline"] = out["value"] / cb["value_reference_orchestration_estimate"]
            # the thread setting `import megapose` itself enforces (reference src/megapose/__init__.py:39-40), smaller sample
            cb1 = cpu_baseline(ds, obs.images, obs.K, det.bboxes, budget_s=8.0, threads=1)
            cb1.pop("_values")
            out["cpu_baseline_1thread"] = cb1
            if not out["parity"]["ok"]:
                rc = 3
        print(json.dumps(out))
        if rc:
            print(f"bench.py: PARITY FAILED (tolerance {PARITY_TOL}): {out['parity']}", file=sys.stderr)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
