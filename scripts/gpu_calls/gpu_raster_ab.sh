#!/bin/bash
# A/B on ONE box: the round-2 tree (git worktree under _build/r02tree, its own library) against the current tree, same bench command
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/raster_ab
mkdir -p $O
timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/cur.json 2> $O/cur.err
(cd _build/r02tree && timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/r02.json 2> $O/r02.err)
timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/cur2.json 2> $O/cur2.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/raster_ab"
for n in ("cur", "r02", "cur2"):
    try:
        d = json.loads(open(f"{O}/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"], 1), {k: v for k, v in d["kernel_ms_per_step"].items() if "raster" in k or "maxpool" in k})
    except Exception as e:
        print(n, "failed", e, open(f"{O}/{n}.err").read()[-600:])
PY
