#!/bin/bash
# round 4: "large"-list pieces that fit the 32-bit edge functions take the block visits: every raster bit-exact test + the new mid-size
# test, A/B against the previous raster (scripts/microbench/_build/libmp_engine_base.so) on the same box, phase probe
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_mid
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_stem_records.py tests/test_gpu_pipeline.py -k "raster or textur or golden or cnn_input" -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
bash scripts/gpu_calls/gpu_raster_ab3.sh base 2>&1 | tee $O/ab.txt
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/scripts/microbench/_build/libmp_engine_prof.so timeout 300 python scripts/raster_phases.py > $O/phases.txt 2>&1
grep -v "^  *per tile\|inside the block" $O/phases.txt | cut -c1-400
