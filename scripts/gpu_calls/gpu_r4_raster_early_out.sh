#!/bin/bash
# round 4: second-phase early-out of the block visits: raster bit-exact tests on the new library, then A/B against the previous raster
# (scripts/microbench/_build/libmp_engine_base.so) on the same box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_early
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_stem_records.py -k "raster or textur" -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
bash scripts/gpu_calls/gpu_raster_ab3.sh base 2>&1 | tee $O/ab.txt
