#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c18
mkdir -p $O
for st in 0 2400 4800 9600 0 4800; do
  echo "stagger $st" >> $O/bb.log
  MP_CONV_STAGGER=$st timeout 200 python scripts/bench_backbone.py --cin 27 --batch 576 >> $O/bb.log 2>&1
done
MP_CONV_STAGGER=4800 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or backbone" > $O/pytest_conv.log 2>&1
