#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c8
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "multiview or backbone or pose_ops" > $O/pytest_mv.log 2>&1; echo "rc=$?" >> $O/pytest_mv.log
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider -x > $O/pytest_pipe.log 2>&1; echo "rc=$?" >> $O/pytest_pipe.log
timeout 500 python bench.py --steps 3 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err; echo "rc=$?" >> $O/bench_full.err
tail -n 5 $O/pytest_mv.log $O/pytest_pipe.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c8/bench_full.json'))
print(d['value'], d['ms_per_step'])
for k,v in d['extras'].items(): print(k, {kk:(round(vv,2) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ('note','workload')})
print(d['cpu_baseline'])
PY
tail -n 3 $O/bench_full.err
