#!/bin/bash
# round 6, call 5 (as call 4, now with `base` really built + request reordering in the prologue + successor L2 prefetch; MP_WINO_PREFETCH=0 A/B):
# round 6, call 4: the restructured bf16x9 Winograd kernel (store offsets computed after the prologue barrier, residual requested under a
# peeled, lean last K step, accumulators read out 64 at a time, padded exchange pitch) against the kernel before the restructuring (base),
# alternating on one box; phase stamps of the new kernel; the kernel's parity tests incl. the permuted builds; bench with both.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c5
mkdir -p $O
B=scripts/microbench/_build
for rep in 1 2; do
  timeout 200 $B/native_wino_check > $O/new_$rep.log 2>&1; echo "rc=$?" >> $O/new_$rep.log
  LD_LIBRARY_PATH=$B/base timeout 200 $B/native_wino_check > $O/base_$rep.log 2>&1; echo "rc=$?" >> $O/base_$rep.log
  echo "== new ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|MISMATCH|rc=" $O/new_$rep.log | cut -c1-230
  echo "== base ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|MISMATCH|rc=" $O/base_$rep.log | cut -c1-230
done
for rep in 1 2; do
  MP_WINO_PREFETCH=0 timeout 200 $B/native_wino_check > $O/nopf_$rep.log 2>&1; echo "rc=$?" >> $O/nopf_$rep.log
  echo "== new, prefetch off ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|MISMATCH|rc=" $O/nopf_$rep.log | cut -c1-230
done
LD_LIBRARY_PATH=$B/phases timeout 200 $B/native_wino_check > $O/phases.log 2>&1; echo "rc=$?" >> $O/phases.log
echo "== phases"; grep -E "PHASE|ALL|FAIL|rc=" $O/phases.log | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wino_permuted.py -m gpu -q -p no:cacheprovider -k "winograd or backbone or exact_piece" > $O/pytest_product.log 2>&1; echo "== pytest product"; tail -n 3 $O/pytest_product.log
for v in product base product base; do
  L=$PWD/megapose6d_amd/libmp_engine.so; [ $v = base ] && L=$PWD/$B/base/libmp_engine.so
  MP_ENGINE_LIB=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_${v}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c5/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"), b.get("parity",{}).get("ok"))
    except Exception as e: print(f, "error", e)
PY
tail -n 5 $O/bench.err
