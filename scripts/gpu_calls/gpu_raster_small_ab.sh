#!/bin/bash
# A/B on ONE box: the small-footprint coverage form off / 2x2 (product build) / 3x2 / 3x3 pixels
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/raster_ab
mkdir -p $O
B=$GRAFT_REPO_ROOT/scripts/microbench/_build
for v in s00 prod s32 s33; do
  if [ $v = prod ]; then L=$GRAFT_REPO_ROOT/megapose6d_amd/libmp_engine.so; else L=$B/libmp_engine_$v.so; fi
  MP_ENGINE_LIB=$L timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/$v.json 2> $O/$v.err
done
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/raster_ab"
for n in ("s00", "prod", "s32", "s33"):
    try:
        d = json.loads(open(f"{O}/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"], 1), {k: v for k, v in d["kernel_ms_per_step"].items() if "raster" in k})
    except Exception as e:
        print(n, "failed", e, open(f"{O}/{n}.err").read()[-400:])
PY
