#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c21
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or backbone" > $O/pytest_conv.log 2>&1
timeout 200 python scripts/bench_backbone.py --cin 27 --batch 576 > $O/bb.log 2>&1
MP_PROF_DETAIL=1 timeout 200 python scripts/profile_layers.py > $O/layers.log 2>&1
timeout 200 python scripts/conv_slope.py > $O/slope.log 2>&1
