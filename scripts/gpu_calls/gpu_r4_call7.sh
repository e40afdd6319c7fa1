#!/bin/bash
# round 4, call 7: occlusion bound in the block visits: every raster bit-exact test + the new stress test + pipeline golden + multi-rank rig + quick bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c7
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_pipeline.py -k "raster or crop or textur or golden or cnn_input or multi_rank" -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c7/bench.json"))
print(d["value"], d["ms_per_step"]); print(d["kernel_ms_per_step"]); print(d["raster"])
PY
tail -n 3 $O/bench.err
