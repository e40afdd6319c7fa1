#!/bin/bash
# A/B on ONE box: raster_tiles compiled for 4 (product) vs 3 waves per SIMD
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/raster_ab
mkdir -p $O
timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/w4.json 2> $O/w4.err
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/scripts/microbench/_build/libmp_engine_w3.so timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/w3.json 2> $O/w3.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/raster_ab"
for n in ("w4", "w3"):
    try:
        d = json.loads(open(f"{O}/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"], 1), {k: v for k, v in d["kernel_ms_per_step"].items() if "raster" in k}, d["parity"]["ok"] if "parity" in d else None)
    except Exception as e:
        print(n, "failed", e, open(f"{O}/{n}.err").read()[-600:])
PY
