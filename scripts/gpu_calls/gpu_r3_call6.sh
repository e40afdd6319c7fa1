#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c6
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_zz_fp16_renders.py tests/test_gpu_edge_cases.py -m gpu -q -p no:cacheprovider -k "raster or texture or fp16 or crop or overflow or large or clip" > $O/pytest_raster.log 2>&1; echo "rc=$?" >> $O/pytest_raster.log
timeout 120 python scripts/bench_raster.py 1 17 > $O/bench_raster.log 2>&1; echo "rc=$?" >> $O/bench_raster.log
MP_ENGINE_LIB=scripts/microbench/_build/libmp_engine_prof.so timeout 150 python scripts/raster_phases.py > $O/raster_phases.log 2>&1; echo "rc=$?" >> $O/raster_phases.log
timeout 200 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; echo "rc=$?" >> $O/bench_quick.err
tail -n 6 $O/pytest_raster.log; cat $O/bench_raster.log; grep -v "^  coverage" $O/raster_phases.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c6/bench_quick.json'))
print(d['value'], d['ms_per_step']); print(d['kernel_ms_per_step']); print(d.get('parity'))
PY
