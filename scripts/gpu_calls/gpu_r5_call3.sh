#!/bin/bash
# round 5, call 3: the compacted rasteriser launch (classify + early exit + light kernel): every raster test + the new A/B equality test, the
# record tests, quick benches with the compaction on / off, and the config-3 parity figures with the record path on / off.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_stem_records.py tests/test_gpu_pipeline.py tests/test_gpu_zz_fp16_renders.py -m gpu -q -p no:cacheprovider -k "raster or crop or textur or golden or cnn_input or record or f16 or fp16 or multiview" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 8 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_compact.json 2> $O/bench_compact.err; echo "rc=$?" >> $O/bench_compact.err
MP_RASTER_COMPACT=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_direct.json 2> $O/bench_direct.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_compact2.json 2> $O/bench_compact2.err
timeout 600 python scripts/parity_config3_debug.py resnet34 > $O/parity_c3.txt 2>&1
python - <<'PY'
import json
O="gpurun_out/r5c3"
def load(f):
    try: return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: return {"error": str(e)}
for n in ("bench_compact","bench_direct","bench_compact2"):
    b=load(f"{O}/{n}.json"); print(n, b.get("value"), b.get("ms_per_step"), b.get("raster")); print("  ", {k:v for k,v in (b.get("kernel_ms_per_step") or {}).items() if "raster" in k})
PY
grep -E "^(RECORDS|FP32)" $O/parity_c3.txt | cut -c1-1800
tail -n 3 $O/parity_c3.txt | cut -c1-400
