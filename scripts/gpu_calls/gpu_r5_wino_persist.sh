#!/bin/bash
# round 5: persistent form of the bf16x9 Winograd kernel with the K loop untouched (next unit's step-0 patch + first weights requested under the
# epilogue's second half), MP_WINO_PERSIST=1 vs 0, alternating on one box (native_wino_check: parity vs the direct kernel + CLK telemetry),
# then the bench with both.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5wp
mkdir -p $O
B=scripts/microbench/_build
for rep in 1 2; do
  MP_WINO_PERSIST=1 timeout 200 $B/native_wino_check > $O/persist1_$rep.log 2>&1; echo "rc=$?" >> $O/persist1_$rep.log
  MP_WINO_PERSIST=0 timeout 200 $B/native_wino_check > $O/persist0_$rep.log 2>&1; echo "rc=$?" >> $O/persist0_$rep.log
  echo "== persistent ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|rc=" $O/persist1_$rep.log | cut -c1-230
  echo "== one workgroup per unit ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|rc=" $O/persist0_$rep.log | cut -c1-230
done
MP_WINO_PERSIST=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "winograd or backbone" > $O/pytest_persist.log 2>&1; tail -n 2 $O/pytest_persist.log
MP_WINO_PERSIST=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_persist1.json 2> $O/bench_persist1.err
MP_WINO_PERSIST=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_persist0.json 2> $O/bench_persist0.err
MP_WINO_PERSIST=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_persist1b.json 2> $O/bench_persist1b.err
python - <<'PY'
import json
O="gpurun_out/r5wp"
for n in ("bench_persist1","bench_persist0","bench_persist1b"):
    try:
        b=json.loads(open(f"{O}/{n}.json").read().strip().splitlines()[-1]); print(n, b["value"], b["ms_per_step"], b["roofline"]["frac"], b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"), b["parity"] if "parity" in b else "")
    except Exception as e: print(n, "error", e)
PY
