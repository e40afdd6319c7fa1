#!/bin/bash
# round 5: the round's run-time switches all off (direct rasteriser launch, dense stem walk, one-workgroup-per-CU Winograd threshold = the
# round-4 behaviour of those parts) vs the defaults, alternating on ONE box: what the round's adopted changes are worth together on the
# `value` workload and on the released K = 5 recipe.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5ab
mkdir -p $O
for rep in 1 2; do
  MP_RASTER_COMPACT=0 MP_STEM_SPARSE=0 MP_WINO_MIN_WGS=256 timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/off_$rep.json 2>> $O/err.log
  timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/on_$rep.json 2>> $O/err.log
done
MP_RASTER_COMPACT=0 MP_STEM_SPARSE=0 MP_WINO_MIN_WGS=256 timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --no-extras > $O/c4_off.json 2>> $O/err.log
timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --no-extras > $O/c4_on.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5ab/*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); k=b["kernel_ms_per_step"]
        print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), {n:v for n,v in k.items() if n.startswith(("raster","conv_stem","pool_zero"))})
    except Exception as e: print(f, "error", e)
PY
