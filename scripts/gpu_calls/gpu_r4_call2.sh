#!/bin/bash
# round 4, call 2: stem records end to end (new tests + the pipeline / full-size parity tests on the record path) + a quick bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stem_records.py tests/test_gpu_refiner_graph.py tests/test_gpu_pipeline.py tests/test_gpu_parity_full_size.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 25 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c2/bench.json"))
print(d["value"], d["ms_per_step"]); print(d["kernel_ms_per_step"]); print(d["stage_s"])
PY
tail -n 3 $O/bench.err
