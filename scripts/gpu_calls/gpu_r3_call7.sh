#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c7
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "multiview or backbone or pose_ops" > $O/pytest_mv.log 2>&1; echo "rc=$?" >> $O/pytest_mv.log
tail -n 25 $O/pytest_mv.log
