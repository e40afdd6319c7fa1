#!/bin/bash
# round 4, call 5: fused stem + max pool (tests, quick bench)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c5
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stem_records.py tests/test_gpu_pipeline.py::test_pipeline_matches_reference_golden -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c5/bench.json"))
print(d["value"], d["ms_per_step"]); print(d["kernel_ms_per_step"])
PY
tail -n 3 $O/bench.err
