#!/bin/bash
# Final measurement call of a round (run through gpurun): the whole GPU test suite, smoke, the driver's bench command, the other configs,
# rocprofv3 kernel stats of the bench command and the three PMC passes (ONE hardware counter group per pass: FETCH_SIZE | WRITE_SIZE |
# SQ_*) that feed profiles/rNN_traffic.json (scripts/pmc_summary.py).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
mkdir -p $O
R="$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
T0=$(date +%s); timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$? wall_s=$(( $(date +%s) - T0 ))" >> $O/bench_n1.err
for c in 3 4 5; do timeout 200 python bench.py --config $c --steps 1 --warmup 1 > $O/bench_c$c.json 2> $O/bench_c$c.err; done
timeout 200 python scripts/icp_timing.py > $O/icp_timing.txt 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $R/$O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace -d $R/$O/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/$O/pmc_$c.log 2>&1
done
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/$O/pmc_SQ -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/$O/pmc_SQ.log 2>&1
cd $R
# summaries for profiles/: per-kernel stats of the bench command, HBM traffic / SQ counters per launch (scripts/pmc_summary.py)
cp $(find $O/stats -name "*kernel_stats.csv" | head -n 1) $O/bench_kernel_stats.csv 2>/dev/null
python scripts/pmc_summary.py $O/traffic.json "gpurun_out/final (round 6 final code, scripts/gpu_final.sh): rocprofv3 --pmc, one counter group per pass (FETCH_SIZE | WRITE_SIZE | SQ_*), python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline" $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.csv" -size +30M -delete
tail -n 3 $O/pytest_gpu.log; tail -n 2 $O/smoke.log; tail -c 400 $O/bench_n1.json; tail -n 2 $O/bench_n1.err
