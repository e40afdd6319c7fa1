#!/bin/bash
# round 4, call 6: the multi-rank bench path on one GPU (gloo rig) + the default full bench line (extras, cpu baseline, parity)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c6
mkdir -p $O
timeout 600 python -m pytest "tests/test_gpu_pipeline.py::test_bench_multi_rank_code_path_with_two_ranks_on_one_gpu" -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 15 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c6/bench.json"))
print(d["value"], d["ms_per_step"], d["vs_cpu_baseline"] if "vs_cpu_baseline" in d else None)
print(d.get("parity")); print(d.get("cpu_baseline")); print(d.get("host"))
for k,v in d.get("extras",{}).items(): print(k, {kk:(round(vv,2) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk!="note" and kk!="workload"})
PY
tail -n 3 $O/bench.err
