#!/bin/bash
# round 6, call 21: the step before the last peeled (request-free points 2 / 3: 16 memory instructions fewer per unit) + the store loop's LDS
# offset opaque per unit (8 hoisted address registers gone); base = HEAD.  Alternating on one box; parity tests; bench with both.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c21
mkdir -p $O
B=scripts/microbench/_build
for rep in 1 2; do
  timeout 200 $B/native_wino_check > $O/persist_$rep.log 2>&1; echo "rc=$?" >> $O/persist_$rep.log
  LD_LIBRARY_PATH=$B/base timeout 200 $B/native_wino_check > $O/plain_$rep.log 2>&1; echo "rc=$?" >> $O/plain_$rep.log
  echo "== new ($rep)"; grep -E "CLK|TIME.*bf16x9|ALL|FAIL|MISMATCH|rc=" $O/persist_$rep.log | cut -c1-230
  echo "== base = HEAD ($rep)"; grep -E "CLK|TIME.*bf16x9|ALL|FAIL|MISMATCH|rc=" $O/plain_$rep.log | cut -c1-230
done
MP_WINO_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "winograd or backbone or exact_piece" > $O/pytest_persist.log 2>&1; echo "== pytest persistent"; tail -n 3 $O/pytest_persist.log
for v in 1 0 1 0; do
  L=$PWD/megapose6d_amd/libmp_engine.so; [ $v = 0 ] && L=$PWD/$B/base/libmp_engine.so
  MP_ENGINE_LIB=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_persist${v}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c21/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"))
    except Exception as e: print(f, "error", e)
PY
tail -n 3 $O/bench.err
