#!/bin/bash
# round 6, call 15: persistent form as the default launch: the whole GPU test suite, phase stamps of both forms, bench with both.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c15
mkdir -p $O
B=scripts/microbench/_build
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 4 $O/pytest_gpu.log
for v in 1 0; do
  MP_WINO_PERSIST=$v LD_LIBRARY_PATH=$B/phases timeout 200 $B/native_wino_check > $O/phases_persist$v.log 2>&1
  echo "== phases, MP_WINO_PERSIST=$v"; grep -E "PHASE|ALL|FAIL" $O/phases_persist$v.log | cut -c1-330
done
for v in 1 0 1 0; do
  MP_WINO_PERSIST=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_persist${v}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c15/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"))
    except Exception as e: print(f, "error", e)
PY
tail -n 3 $O/bench.err
