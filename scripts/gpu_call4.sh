#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c4
mkdir -p $O
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/scripts/microbench/_build/libmp_engine_prof.so timeout 300 python scripts/raster_phases.py > $O/phases.log 2>&1
timeout 300 python scripts/bench_backbone.py --batch 576 --iters 3 > $O/bb_default.log 2>&1
MP_CONV_VARIANT=2305 timeout 300 python scripts/bench_backbone.py --batch 576 --iters 3 > $O/bb_2305.log 2>&1
MP_CONV_VARIANT=2817 timeout 300 python scripts/bench_backbone.py --batch 576 --iters 3 > $O/bb_2817.log 2>&1
MP_CONV_VARIANT=2305 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or backbone" > $O/pytest_conv2305.log 2>&1
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_parity_full_size.py -x -q > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
