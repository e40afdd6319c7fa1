#!/bin/bash
# refresh of the bench line + the PMC passes behind profiles/r03_traffic.json after a kernel change (subset of scripts/gpu_final.sh)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
mkdir -p $O
R="$GRAFT_REPO_ROOT"
rm -rf $O/pmc_* $O/stats
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $R/$O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace -d $R/$O/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/$O/pmc_$c.log 2>&1
done
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/$O/pmc_SQ -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/$O/pmc_SQ.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.csv" -size +30M -delete
python scripts/pmc_summary.py profiles/r03_traffic.json "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_*} --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline (scripts/gpu_calls/gpu_pmc_refresh.sh = the PMC part of scripts/gpu_final.sh, round 3 final code, MI355X)" $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $O/pmc_SQ/p_counter_collection.csv | head -4
cp profiles/r03_traffic.json $O/r03_traffic.json
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$?" >> $O/bench_n1.err
head -c 260 $O/bench_n1.json; tail -n 1 $O/bench_n1.err
