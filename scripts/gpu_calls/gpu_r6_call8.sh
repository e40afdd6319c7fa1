#!/bin/bash
# round 6, call 8: the epilogue store loop as straight-line code (second output = template parameter, ReLU = max with 0 or -inf) vs the tree one commit earlier (base).
# v_perm pairs in the ds_write gaps) against the tables of rounds 4/5 (base = HEAD before the edit), alternating on one box; parity tests.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c8
mkdir -p $O
B=scripts/microbench/_build
for rep in 1 2; do
  timeout 200 $B/native_wino_check > $O/new_$rep.log 2>&1; echo "rc=$?" >> $O/new_$rep.log
  LD_LIBRARY_PATH=$B/base timeout 200 $B/native_wino_check > $O/base_$rep.log 2>&1; echo "rc=$?" >> $O/base_$rep.log
  echo "== new ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|MISMATCH|rc=" $O/new_$rep.log | cut -c1-230
  echo "== base ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|MISMATCH|rc=" $O/base_$rep.log | cut -c1-230
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wino_permuted.py -m gpu -q -p no:cacheprovider -k "winograd or backbone or exact_piece" > $O/pytest_product.log 2>&1; echo "== pytest product"; tail -n 3 $O/pytest_product.log
for v in product base product base; do
  L=$PWD/megapose6d_amd/libmp_engine.so; [ $v = base ] && L=$PWD/$B/base/libmp_engine.so
  MP_ENGINE_LIB=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_${v}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c8/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"))
    except Exception as e: print(f, "error", e)
PY
tail -n 3 $O/bench.err
