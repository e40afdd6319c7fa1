#!/bin/bash
# Round 3, GPU call 2: block-visit rasteriser -- bit-exactness tests, micro-benchmark (1 / 4 samples, 4 vs 3 waves per SIMD), the
# config-3 parity test with the flip-tolerant logit rule, a short bench line.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c2
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_zz_fp16_renders.py -m gpu -q -p no:cacheprovider -k "raster or texture or fp16 or crop" > $O/pytest_raster.log 2>&1; echo "rc=$?" >> $O/pytest_raster.log
timeout 120 python scripts/bench_raster.py 1 17 > $O/bench_raster.log 2>&1; echo "rc=$?" >> $O/bench_raster.log
MP_ENGINE_LIB=scripts/microbench/_build/libmp_engine_w3.so timeout 120 python scripts/bench_raster.py 1 17 > $O/bench_raster_w3.log 2>&1; echo "rc=$?" >> $O/bench_raster_w3.log
timeout 300 python -m pytest tests/test_gpu_parity_full_size.py tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
timeout 200 python bench.py --steps 3 --warmup 1 --no-extras > $O/bench_quick.json 2> $O/bench_quick.err; echo "rc=$?" >> $O/bench_quick.err
tail -n 6 $O/pytest_raster.log $O/bench_raster.log $O/bench_raster_w3.log $O/pytest_parity.log
tail -c 1500 $O/bench_quick.json; tail -n 5 $O/bench_quick.err
