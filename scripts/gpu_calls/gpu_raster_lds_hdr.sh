#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/raster_ab
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py tests/test_gpu_zz_fp16_renders.py -m gpu -q -p no:cacheprovider -k "raster or render or texture or fp16" 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/ldshdr.json 2> $O/ldshdr.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/raster_ab"
d = json.loads(open(f"{O}/ldshdr.json").read().strip().splitlines()[-1])
print("ldshdr", round(d["ms_per_step"], 1), {k: v for k, v in d["kernel_ms_per_step"].items() if "raster" in k}, d.get("parity", {}).get("ok"))
PY
