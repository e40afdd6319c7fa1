#!/bin/bash
# GPU call 1 of round 2: full -m gpu suite, bench line, microbenchmarks, PMC passes, host thread sweep
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2c1
O=gpurun_out/r2c1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
timeout 300 python scripts/conv_slope.py > $O/slope.log 2>&1
timeout 120 scripts/microbench/unaligned_ld > $O/unaligned.log 2>&1
MP_PROF_DETAIL=1 timeout 300 python scripts/profile_layers.py 27 576 > $O/layers27.log 2>&1
MP_PROF_DETAIL=1 timeout 300 python scripts/profile_layers.py 9 576 > $O/layers9.log 2>&1
timeout 120 python scripts/bench_raster.py 1 3 > $O/raster.log 2>&1
timeout 600 python bench.py --cpu-thread-sweep 1,16,32,64,128,256 > $O/sweep.log 2>&1
cp gpurun_out/cpu_thread_sweep.json $O/ 2>/dev/null
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/$O/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/$O/pmc_$c.log 2>&1
done
cd "$R"
find $O -name "*.csv" -size +30M -delete
ls -la $O > $O/ls.txt
