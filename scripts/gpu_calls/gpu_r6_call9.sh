#!/bin/bash
# round 6, call 9: the pose pipeline's chunks interleaved over 1 / 2 / 3 HIP streams (PoseEstimator.n_streams, MP_N_STREAMS) with the round-6
# kernels: do the tails of one chunk's launches (a 576-row Winograd launch is 5.6 .. 42.2 rounds of 256 workgroups) fill with the other's work?
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c9
mkdir -p $O
for ns in 1 2 3 1 2; do
  MP_N_STREAMS=$ns timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_ns${ns}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c9/bench_*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), round(b["roofline"]["frac"],4), {k:v for k,v in list(b["kernel_ms_per_step"].items())[:5]})
    except Exception as e: print(f, "error", e)
PY
tail -n 3 $O/bench.err
