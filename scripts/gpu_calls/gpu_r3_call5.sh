#!/bin/bash
# Round 3, GPU call 5: rasteriser with per-tile large lists + batched headers + pipelined visits
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c5
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_zz_fp16_renders.py tests/test_gpu_edge_cases.py -m gpu -q -p no:cacheprovider -k "raster or texture or fp16 or crop or overflow or large or clip" > $O/pytest_raster.log 2>&1; echo "rc=$?" >> $O/pytest_raster.log
timeout 120 python scripts/bench_raster.py 1 17 > $O/bench_raster.log 2>&1; echo "rc=$?" >> $O/bench_raster.log
MP_ENGINE_LIB=scripts/microbench/_build/libmp_engine_prof.so timeout 150 python scripts/raster_phases.py > $O/raster_phases.log 2>&1; echo "rc=$?" >> $O/raster_phases.log
tail -n 6 $O/pytest_raster.log; cat $O/bench_raster.log $O/raster_phases.log
