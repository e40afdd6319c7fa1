#!/bin/bash
# the whole GPU suite + smoke, then the PMC / stats / bench refresh, on the final tree
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/last
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
tail -n 3 $O/pytest_gpu.log; tail -n 2 $O/smoke.log
bash scripts/gpu_calls/gpu_pmc_refresh.sh
for c in 3 4 5; do timeout 200 python bench.py --config $c --steps 1 --warmup 1 > gpurun_out/final/bench_c$c.json 2> gpurun_out/final/bench_c$c.err; done
