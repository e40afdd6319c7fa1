#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c8
mkdir -p $O
B=$GRAFT_REPO_ROOT/scripts/microbench/_build
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py -x -q -k "raster or textured or crop" > $O/pytest_raster.log 2>&1; echo "rc=$?" >> $O/pytest_raster.log
timeout 200 python scripts/bench_raster.py 1 17 > $O/raster.log 2>&1
MP_ENGINE_LIB=$B/libmp_engine_prof.so timeout 300 python scripts/raster_phases.py > $O/phases.log 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 2 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
