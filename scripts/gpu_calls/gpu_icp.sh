#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/icp
timeout 600 python -m pytest tests/test_gpu_icp.py -m gpu -q -p no:cacheprovider -x -s -k "step_for_step or user_masks or pipeline" > gpurun_out/icp/pytest_icp.log 2>&1; echo "rc=$?" >> gpurun_out/icp/pytest_icp.log
grep -n "^(\|passed\|failed\|Error\|max |"  gpurun_out/icp/pytest_icp.log
timeout 300 python scripts/icp_timing.py 2>&1 | tail -4 | tee gpurun_out/icp/timing.txt
