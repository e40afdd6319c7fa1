#!/usr/bin/env python
"""bench.py -- pose-hypotheses/sec of the hot path (render + coarse + refine + score) on N MI355X.

One "step" = one `PoseEstimator.run_inference_pipeline` call over a synthetic 640x480 frame resident in HBM:
BASELINE.json configs[1] = megapose-1.0-RGB structure (coarse 9-ch + refiner 27-ch vanilla ResNet-34, fp32),
1 object x 576 SO(3)-grid hypotheses, ALL 576 refined for 5 iterations (n_pose_hypotheses=576), then re-scored:
576 coarse + 2880 refine + 576 score CNN rows and 12 672 rendered views per object.
With N > 1 (torchrun, one rank per GPU, RCCL): N objects in the frame, rows sharded rank::world, all-gathers of
the packed logits/poses per stage (weak scaling: 1 object x 576 hypotheses per GPU).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     the dominant kernel (fp32-MFMA implicit-GEMM conv) measured live with HIP events on its launch stream
  "cpu_baseline": the oracle ("port" of the reference's CPU path) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

METRIC = "pose-hypotheses/sec (render+coarse+refine), 640x480, megapose-1.0-RGB"
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
N_HYP, N_ITERS = 576, 5


def cpu_baseline(tmp_dir: str, obs_images: torch.Tensor, K: torch.Tensor, bboxes: torch.Tensor, budget_s: float = 20.0,
                 threads: int = 0) -> dict:
    """Oracle (port of the reference CPU path: reference orchestration + torch-CPU fp32 CNN + C software rasteriser standing in
    for Panda3D, which cannot be installed offline) on a bounded sample of the same workload.  Threads: min(host cores, 32)
    -- more threads only add OpenMP overhead at these batch sizes (the reference itself pins 1 thread, __init__.py:39-40).
    The sample is sized adaptively from a 4-row probe so that it stays within `budget_s` seconds of CPU work."""
    from megapose6d_amd import mesh_io
    from megapose6d_amd import synthetic as syn
    from megapose6d_amd.mesh_db import MeshDataBase
    from megapose6d_amd.pose_estimator import load_SO3_grid
    from oracle import geometry as og
    from oracle import pipeline as op
    from oracle import raster as orr

    threads = threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    ds = syn.make_object_dataset(tmp_dir, n_objects=1, seed=0)
    meshes = {o.label: mesh_io.load_rigid_object(o) for o in ds.list_objects}
    db = MeshDataBase.from_object_ds(ds).batched()
    rend = orr.OracleBatchRenderer(meshes)
    preds = {}
    for role, seed in (("coarse", 11), ("refiner", 12)):
        cfg = syn.make_cfg(role)
        head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
        preds[role] = op.OraclePosePredictor(cfg, syn.make_state_dict("vanilla_resnet34", syn.n_inputs_for(cfg), head, n_out, seed=seed),
                                             db.labels.tolist(), db.points, rend)
    grid = load_SO3_grid(N_HYP)
    images, Kc = obs_images.cpu(), K.cpu()
    label = ds[0].label
    n_max = 48
    with torch.no_grad():
        T = og.TCO_init_from_boxes_autodepth_with_R(bboxes[:1].cpu().float().repeat(n_max, 1), db.points[:1].repeat(n_max, 1, 1),
                                                   Kc.repeat(n_max, 1, 1), grid[:n_max])
        im = torch.zeros(n_max, dtype=torch.long)

        def coarse(n, poses):
            t0 = time.perf_counter()
            preds["coarse"].forward_coarse(images, im[:n], Kc.repeat(n, 1, 1), [label] * n, poses[:n])
            return time.perf_counter() - t0

        coarse(2, T)  # warm-up
        t_row = coarse(4, T) / 4  # probe
        n_coarse = int(max(4, min(n_max, 0.35 * budget_s / max(t_row, 1e-4))))
        n_refine = int(max(1, min(8, 0.65 * budget_s / max(t_row * 4.5 * (N_ITERS + 1), 1e-4))))
        t_coarse = coarse(n_coarse, T)
        t0 = time.perf_counter()
        outs = preds["refiner"].forward(images, im[:n_refine], Kc.repeat(n_refine, 1, 1), [label] * n_refine, T[:n_refine], N_ITERS)
        t_refine = time.perf_counter() - t0
        t_score = coarse(n_refine, outs[-1]["TCO_output"])
    t_full = t_coarse * (N_HYP / n_coarse) + (t_refine + t_score) * (N_HYP / n_refine)
    return {"value": N_HYP / t_full, "unit": "pose-hypotheses/s", "cores": threads, "kind": "port",
            "sample": f"{n_coarse} of 576 coarse rows + {n_refine} of 576 hypotheses x {N_ITERS} refine iters + {n_refine} score rows, "
                      f"{t_coarse + t_refine + t_score:.1f} s of CPU work on {threads} threads ({os.cpu_count()} host cores), extrapolated "
                      "linearly per stage; Panda3D replaced by the oracle's C rasteriser"}


def extras(est, obs, det, steps: int) -> dict:
    """Secondary numbers (NOT `value`): the released inference parameters (n_pose_hypotheses = 1 / 5) and the
    optional split-precision conv modes on the headline workload.  Same timing discipline, 1 warm-up + `steps` timed calls."""

    def timed(k_hyp: int) -> float:
        est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=N_ITERS, n_pose_hypotheses=k_hyp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=N_ITERS, n_pose_hypotheses=k_hyp)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    out = {}
    for k in (1, 5):
        dt = timed(k)
        out[f"n_pose_hypotheses={k}"] = {"ms_per_call": dt * 1e3, "coarse_hypotheses_per_s": N_HYP / dt,
                                         "note": "576 coarse rows + K x 5 refine rows + K score rows (megapose-1.0-RGB[-multi-hypothesis] defaults)"}
    est.n_streams = 3  # chunk interleave on 3 HIP streams: fills the tails of the conv grids and overlaps raster with MFMA work; not the
    dt = timed(N_HYP)  # default because overlapping kernels distort the per-kernel event timing the roofline figures rest on
    est.n_streams = 1
    out["three_stream_interleave"] = {"ms_per_step": dt * 1e3, "pose_hypotheses_per_s": N_HYP / dt,
                                      "note": "same fp32 path, PoseEstimator.n_streams=3 (MP_N_STREAMS); not used for `value`"}
    for prec in (9, 6):
        for m in (est.coarse_model, est.refiner_model):
            m.conv_precision = prec
            m._engine_bb = None
        dt = timed(N_HYP)
        out[f"conv_bf16x{prec}_split"] = {"ms_per_step": dt * 1e3, "pose_hypotheses_per_s": N_HYP / dt,
                                          "note": "optional mode: fp32 operands split exactly into 3 bf16 pieces, bf16 MFMA, fp32 accumulate; "
                                                  "meets the same parity bounds (tests/test_gpu_pipeline.py), not used for `value`"}
    for m in (est.coarse_model, est.refiner_model):
        m.conv_precision = 0
        m._engine_bb = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--backbone", default="vanilla_resnet34")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (faithful K=1/K=5 configs, optional split-precision modes)")
    ap.add_argument("--precision", type=int, default=0, help="0 = native fp32 MFMA (default, what `value` is quoted on); 9 / 6 = optional bf16 split modes")
    a = ap.parse_args()

    from megapose6d_amd import distributed as mpd
    from megapose6d_amd import engine as eng
    from megapose6d_amd.scene import make_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs `python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py ...` (WORLD_SIZE={world})")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        mpd.init_from_env("nccl")
    n_cu, lds, arch = eng.device_info()

    tmp = tempfile.mkdtemp(prefix=f"mp_bench_r{rank}_")
    n_obj = world  # weak scaling: one object x 576 hypotheses per GPU
    est, obs, det, _ = make_scene(n_objects=n_obj, seed=0, backbone=a.backbone, SO3_grid_size=N_HYP, tmp_dir=tmp, distributed=world > 1,
                                   n_streams=int(os.environ.get("MP_N_STREAMS", "1")), precision=a.precision)

    def step():
        return est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=N_ITERS, n_pose_hypotheses=N_HYP)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        final, extra = step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    assert len(final) == n_obj and torch.isfinite(final.poses).all()

    if rank == 0:
        conv = {k: v for k, v in prof.items() if k.startswith("conv_nhwc_f32")}
        dom_name = max(conv, key=lambda k: conv[k]["ms"])
        dom = conv[dom_name]
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        all_conv_tf = sum(v["flops"] for v in conv.values()) / (sum(v["ms"] for v in conv.values()) * 1e-3) / 1e12
        kernel_ms = {k: round(v["ms"] / a.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        rb = prof.get("raster_bands")
        traffic, traffic_src = None, None
        tfile = ROOT / "profiles" / "r01_conv_traffic.json"  # PMC cannot be sampled from inside the process: committed rocprofv3 summary
        if tfile.is_file():
            tj = json.loads(tfile.read_text())
            k = tj["kernels"].get(dom_name.replace(" ", ""))
            if k:
                traffic, traffic_src = k["hbm_bytes_per_launch_corrected"], tj["source"]
        out = {
            "metric": METRIC, "value": n_obj * N_HYP * a.steps / dt, "unit": "pose-hypotheses/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.precision == 0 else f"f32 via exact bf16x{a.precision} operand split (bf16 MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"megapose-1.0-RGB structure ({a.backbone} coarse 9ch + refiner 27ch), {n_obj} object(s) x 576 hypotheses x 5 refine "
                                   "iters (n_pose_hypotheses=576) + re-score, 640x480 frame, 240x320 crops, 10k-triangle meshes",
                       "rows_per_step": n_obj * (2 * N_HYP + N_HYP * N_ITERS),
                       "evals_per_s": n_obj * (2 * N_HYP + N_HYP * N_ITERS) * a.steps / dt, "views_per_step": n_obj * (2 * N_HYP + 4 * N_HYP * N_ITERS),
                       "parallelism": f"rows sharded rank::world over {world} GPU(s)", "arch": arch, "cus": n_cu},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src, "alg_bytes_per_launch": dom["bytes"] / dom["launches"], "launches": dom["launches"],
                         "avg_launch_ms": dom["ms"] / dom["launches"], "avg_launch_gflop": dom["flops"] / dom["launches"] / 1e9,
                         "all_conv_kernels_tflops": all_conv_tf},
            "raster": None if rb is None else {"bound": "hbm", "kernel": "raster_bands", "achieved_GBps": rb["bytes"] / (rb["ms"] * 1e-3) / 1e9,
                                               "peak_GBps": 8000.0, "avg_launch_ms": rb["ms"] / rb["launches"]},
            "kernel_ms_per_step": kernel_ms,
            "stage_s": {"coarse": extra["coarse"]["data"]["time"], "refiner": extra["refiner"]["data"]["time"],
                        "scoring": extra["scoring"]["data"]["time"], "total": extra["time"]},
        }
        if world == 1 and not a.no_extras:
            out["extras"] = extras(est, obs, det, a.steps)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(tmp, obs.images, obs.K, det.bboxes)
            out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            # the thread setting `import megapose` itself enforces (reference src/megapose/__init__.py:39-40), smaller sample
            out["cpu_baseline_1thread"] = cpu_baseline(tmp, obs.images, obs.K, det.bboxes, budget_s=8.0, threads=1)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
