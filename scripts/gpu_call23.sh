#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c23
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or backbone" > $O/pytest_conv.log 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
