#!/bin/bash
# round 6, call 1: root cause of the round-5 "inf / NaN under another register assignment" of conv3x3_wino_bf16x9.
# Hypothesis (scripts/isa_hazards.py, rule R1): the accumulator-reset MFMAs are inline asm; the compiler writes their zero operand with
# v_mov directly in front of the statement; without two wait states the first MFMA reads the registers' previous contents.
#   r5bad    = the archived failing source, unchanged                     -> expected: MISMATCH / NaN
#   r5badfix = the same source + `s_nop 1` in front of the reset MFMAs    -> expected: ALL OK
#   wperm3/8 = today's source, deliberately permuted register assignment  -> expected: ALL OK
#   product  = today's source                                             -> expected: ALL OK
# then the GPU parity tests of the kernel on the permuted builds, and the round's baseline bench.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c1
mkdir -p $O
B=scripts/microbench/_build
for v in r5bad r5badfix wperm3 wperm8; do
  LD_LIBRARY_PATH=$B/$v timeout 200 $B/native_wino_check > $O/check_$v.log 2>&1; echo "rc=$?" >> $O/check_$v.log
  echo "== $v"; grep -E "CASE|ALL|FAIL|MISMATCH|rc=" $O/check_$v.log | cut -c1-200 | tail -n 12
done
MP_WINO_PREFETCH=0 LD_LIBRARY_PATH=$B/r5bad timeout 200 $B/native_wino_check > $O/check_r5bad_pf0.log 2>&1; echo "rc=$?" >> $O/check_r5bad_pf0.log
echo "== r5bad, prefetch off at run time"; grep -E "CASE|ALL|FAIL|MISMATCH|rc=" $O/check_r5bad_pf0.log | cut -c1-200 | tail -n 6
timeout 200 $B/native_wino_check > $O/check_product.log 2>&1; echo "rc=$?" >> $O/check_product.log
echo "== product"; grep -E "CASE|CLK|TIME.*bf16x9|ALL|FAIL|MISMATCH|rc=" $O/check_product.log | cut -c1-230 | tail -n 20
for v in wperm3 wperm8 r5badfix; do
  MP_ENGINE_LIB=$PWD/$B/$v/libmp_engine.so timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "winograd or backbone or exact_piece" > $O/pytest_$v.log 2>&1
  echo "== pytest $v"; tail -n 2 $O/pytest_$v.log
done
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "winograd or backbone or exact_piece" > $O/pytest_product.log 2>&1; echo "== pytest product"; tail -n 2 $O/pytest_product.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_baseline.json 2> $O/bench_baseline.err; echo "rc=$?" >> $O/bench_baseline.err
tail -c 600 $O/bench_baseline.json; tail -n 2 $O/bench_baseline.err
