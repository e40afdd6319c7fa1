#!/bin/bash
# round 5, call 2: RGBD on the stem-record path (48-element records, depth normalised in the rasteriser launch), the padded LDS pixel pitch of
# the even-record stem instances, the n_cu / 4 Winograd threshold: record tests, RGBD parity tests at full size, config-3 bench, quick bench.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c2
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stem_records.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "stem or record or backbone or winograd_path or raster" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
tail -n 15 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_parity_full_size.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider -x > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
tail -n 15 $O/pytest_b.log
timeout 300 python bench.py --config 3 --steps 2 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; echo "rc=$?" >> $O/bench_c3.err
MP_STEM_RECORDS=0 timeout 300 python bench.py --config 3 --steps 1 --warmup 1 > $O/bench_c3_fp32tensor.json 2> $O/bench_c3_fp32tensor.err
timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$?" >> $O/bench_n1.err
timeout 200 python bench.py --config 2 --k-hyp 5 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/k5.json 2> $O/k5.err
python - <<'PY'
import json
O="gpurun_out/r5c2"
def load(f):
    try: return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: return {"error": str(e)}
for n in ("bench_c3","bench_c3_fp32tensor","bench_n1","k5"):
    b=load(f"{O}/{n}.json"); print(n, b.get("value"), b.get("ms_per_step"), (b.get("roofline") or {}).get("frac")); print("  ", dict(list((b.get("kernel_ms_per_step") or {}).items())[:9]))
PY
tail -n 3 $O/bench_c3.err
