#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/graph
timeout 600 python -m pytest tests/test_gpu_refiner_graph.py tests/test_gpu_icp.py -m gpu -q -p no:cacheprovider -x -s > gpurun_out/graph/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/graph/pytest.log
grep -v "^  File\|^Extension" gpurun_out/graph/pytest.log | tail -n 30
timeout 300 python scripts/icp_timing.py 2>&1 | tail -2 | tee gpurun_out/graph/icp_timing.txt
