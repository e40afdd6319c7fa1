#!/bin/bash
# conv with buffer loads (8449 = default schedule, 12289 = two-chunks-ahead pipeline): correctness + A/B timing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c15
mkdir -p $O
for v in 8449 12289; do
  MP_CONV_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or backbone" > $O/pytest_conv_$v.log 2>&1
done
for v in 257 8449 12289 257 8449; do
  MP_CONV_VARIANT=$v timeout 200 python scripts/bench_backbone.py --cin 27 --batch 576 >> $O/bb_$v.log 2>&1
done
for v in 257 8449 12289; do
  MP_CONV_VARIANT=$v timeout 200 python scripts/conv_one_wg.py > $O/conv_2wg_$v.log 2>&1
done
