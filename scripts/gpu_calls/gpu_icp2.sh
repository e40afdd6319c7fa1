#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/icp
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_prediction_runner.py -m gpu -q -p no:cacheprovider > gpurun_out/icp/pytest_icp_all.log 2>&1; echo "rc=$?" >> gpurun_out/icp/pytest_icp_all.log
tail -n 6 gpurun_out/icp/pytest_icp_all.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/icp/prof -o icp -- python $R/scripts/icp_timing.py > $R/gpurun_out/icp/prof.log 2>&1
f=$(find $R/gpurun_out/icp/prof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
