#!/bin/bash
# conv variant 4097 (two-chunks-ahead pipeline): correctness + A/B timing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c12
mkdir -p $O
MP_CONV_VARIANT=4097 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or backbone" > $O/pytest_conv_4097.log 2>&1
for v in 257 4097; do
  MP_CONV_VARIANT=$v timeout 200 python scripts/bench_backbone.py --cin 27 --batch 576 > $O/bb_$v.log 2>&1
  MP_CONV_VARIANT=$v timeout 200 python scripts/conv_one_wg.py > $O/conv_2wg_$v.log 2>&1
  MP_CONV_VARIANT=$v MP_CONV_LDS_PAD_KB=20 timeout 200 python scripts/conv_one_wg.py > $O/conv_1wg_$v.log 2>&1
done
MP_CONV_VARIANT=4097 timeout 200 python scripts/conv_slope.py > $O/slope_4097.log 2>&1
