#!/bin/bash
# round 5: L2 prefetch of the successor workgroup's step-0 patch lines from the bf16x9 Winograd epilogue, MP_WINO_PREFETCH=1 (default) vs 0,
# alternating on one box: native_wino_check (parity vs the direct kernel + CLK telemetry incl. prologue / epilogue cycles), then the bench.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5pf
mkdir -p $O
B=scripts/microbench/_build
for rep in 1 2; do
  MP_WINO_PREFETCH=1 timeout 200 $B/native_wino_check > $O/pf1_$rep.log 2>&1; echo "rc=$?" >> $O/pf1_$rep.log
  MP_WINO_PREFETCH=0 timeout 200 $B/native_wino_check > $O/pf0_$rep.log 2>&1; echo "rc=$?" >> $O/pf0_$rep.log
  echo "== prefetch on ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|rc=" $O/pf1_$rep.log | cut -c1-230
  echo "== prefetch off ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|rc=" $O/pf0_$rep.log | cut -c1-230
done
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "winograd or backbone or exact_piece" > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
for v in 1 0 1 0; do
  MP_WINO_PREFETCH=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_pf${v}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5pf/bench_pf*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"))
    except Exception as e: print(f, "error", e)
PY
