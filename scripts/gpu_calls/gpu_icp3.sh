#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/icp
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_zz_detector.py tests/test_gpu_prediction_runner.py -m gpu -q -p no:cacheprovider -s > gpurun_out/icp/pytest_icp3.log 2>&1; echo "rc=$?" >> gpurun_out/icp/pytest_icp3.log
grep -n "points, iterations\|passed\|failed\|Error\|assert" gpurun_out/icp/pytest_icp3.log | tail -20
timeout 200 python scripts/icp_timing.py 2>&1 | tail -2
