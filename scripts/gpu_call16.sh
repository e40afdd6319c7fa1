#!/bin/bash
# buffer loads as the default in every conv instantiation: correctness + timing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c16
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or backbone" > $O/pytest_conv.log 2>&1
for cin in 27 9; do timeout 200 python scripts/bench_backbone.py --cin $cin --batch 576 >> $O/bb.log 2>&1; done
MP_CONV_VARIANT=257 timeout 200 python scripts/bench_backbone.py --cin 27 --batch 576 >> $O/bb_257.log 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
