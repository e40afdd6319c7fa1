#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/icp
timeout 600 python scripts/icp_debug.py 5 2 11 > gpurun_out/icp/debug.txt 2>&1
tail -n 12 gpurun_out/icp/debug.txt
bash scripts/gpu_icp.sh
