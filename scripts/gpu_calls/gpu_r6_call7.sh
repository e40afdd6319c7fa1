#!/bin/bash
# round 6, call 7: the rebalanced slot tables of the bf16x9 Winograd K loop (conv_wino_bf16_sched.h: one 4-instruction group per MFMA gap,
# v_perm pairs in the ds_write gaps) against the tables of rounds 4/5 (base = HEAD before the edit), alternating on one box; parity tests.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c7
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
for f in sorted(glob.glob("gpurun_out/r6c7/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"))
    except Exception as e: print(f, "error", e)
PY
tail -n 3 $O/bench.err
# the tests of the other round-6 changes: WideResNet fused pool, strict / pose-aware logit rules, strong pose head, bench --gpus 2 on one GPU
timeout 1500 python -m pytest tests/test_gpu_stem_records.py tests/test_gpu_parity_full_size.py tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider -s > $O/pytest_more.log 2>&1; echo "rc=$?" >> $O/pytest_more.log
grep -E "strong pose head|chained score logits" $O/pytest_more.log | cut -c1-400; tail -n 4 $O/pytest_more.log
for bbk in resnet34; do
  timeout 300 python bench.py --config 3 --backbone $bbk --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_c3_$bbk.json 2>> $O/bench.err
  MP_STEM_POOL=0 timeout 300 python bench.py --config 3 --backbone $bbk --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_c3_${bbk}_nopool.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c7/bench_c3*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), {k:v for k,v in list(b["kernel_ms_per_step"].items())[:8]})
    except Exception as e: print(f, "error", e)
PY
