#!/bin/bash
# round 6, call 18 (exchange layout [wave][tile][cout][2]: the pair of a (tile, cout) in one ds_write_b64; base = HEAD) (next unit's index arithmetic under the MFMAs of the last point; base = HEAD) (step-1 patch requests of the prologue spread between the statements of transform rows 1..3; base = HEAD) (the next unit's patch requests moved into the exchange phase, four behind each accumulator block) (as call 12 + magic-number tile arithmetic in both forms, accumulator resets spread over the store loop in the persistent form) (as call 11; the next unit's patch requests now four at a time behind the stores of each store-loop pass, weight requests at the loop top): the persistent form of the bf16x9 Winograd kernel (MP_WINO_PERSIST=1: one workgroup per CU walks the units; the next
# unit's first 22 requests, tile table and accumulator reset are issued at the start of the current unit's store loop) vs the plain launch,
# alternating on one box; the kernel's parity tests with the persistent form; bench with both.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c18
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
for f in sorted(glob.glob("gpurun_out/r6c18/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"))
    except Exception as e: print(f, "error", e)
PY
tail -n 3 $O/bench.err
