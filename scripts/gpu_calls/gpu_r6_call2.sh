#!/bin/bash
# round 6, call 2: (a) packed-fp32 transform / split arithmetic in the bf16x9 Winograd K loop (product, 653 instructions per step) against the
# scalar form (pk0 = the round-5 loop, 781), alternating on one box; (b) cycle stamps inside prologue and epilogue (phases build);
# (c) the DIAG sweep with the finer switches (exp build); (d) the kernel's parity tests on the product; (e) bench with both.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c2
mkdir -p $O
B=scripts/microbench/_build
for rep in 1 2; do
  timeout 200 $B/native_wino_check > $O/pk1_$rep.log 2>&1; echo "rc=$?" >> $O/pk1_$rep.log
  LD_LIBRARY_PATH=$B/pk0 timeout 200 $B/native_wino_check > $O/pk0_$rep.log 2>&1; echo "rc=$?" >> $O/pk0_$rep.log
  echo "== packed ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|MISMATCH|rc=" $O/pk1_$rep.log | cut -c1-230
  echo "== scalar ($rep)"; grep -E "bf16x9 wino|CLK|ALL|FAIL|MISMATCH|rc=" $O/pk0_$rep.log | cut -c1-230
done
LD_LIBRARY_PATH=$B/phases timeout 200 $B/native_wino_check > $O/phases.log 2>&1; echo "rc=$?" >> $O/phases.log
echo "== phases"; grep -E "PHASE|CLK|ALL|FAIL|rc=" $O/phases.log | cut -c1-330
MP_WINO_DIAG_SWEEP=1 LD_LIBRARY_PATH=$B/exp timeout 300 $B/native_wino_check > $O/diag.log 2>&1; echo "rc=$?" >> $O/diag.log
echo "== diag sweep"; grep -E "DIAG|rc=" $O/diag.log | cut -c1-60,195-300
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wino_permuted.py -m gpu -q -p no:cacheprovider -k "winograd or backbone or exact_piece" > $O/pytest_product.log 2>&1; echo "== pytest product"; tail -n 3 $O/pytest_product.log
for v in product pk0 product pk0; do
  L=$PWD/megapose6d_amd/libmp_engine.so; [ $v = pk0 ] && L=$PWD/$B/pk0/libmp_engine.so
  MP_ENGINE_LIB=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_${v}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c2/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), b["kernel_ms_per_step"].get("conv3x3_wino_bf16x9<64t,64c>"), b["roofline"].get("k_loop_cycles_per_16_channel_step"), b["parity"]["ok"] if "parity" in b else None)
    except Exception as e: print(f, "error", e)
PY
