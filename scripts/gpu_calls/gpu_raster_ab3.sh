#!/bin/bash
# A/B on ONE box: product library vs a variant library (scripts/microbench/_build/libmp_engine_$1.so)
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/raster_ab
mkdir -p $O
B=$GRAFT_REPO_ROOT/scripts/microbench/_build
for v in prod $1 prod $1; do
  if [ $v = prod ]; then L=$GRAFT_REPO_ROOT/megapose6d_amd/libmp_engine.so; else L=$B/libmp_engine_$v.so; fi
  MP_ENGINE_LIB=$L timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/x.json 2> $O/x.err
  python - "$v" <<'PY'
import json, os, sys
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/raster_ab"
d = json.loads(open(f"{O}/x.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 1), {k: v for k, v in d["kernel_ms_per_step"].items() if "raster" in k})
PY
done
