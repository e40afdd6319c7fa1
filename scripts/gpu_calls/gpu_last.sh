#!/bin/bash
# last call of round 3: the whole GPU suite and the driver's bench command on the final tree
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/last
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$?" >> $O/bench_n1.err
tail -n 3 $O/pytest_gpu.log; tail -n 2 $O/smoke.log; tail -n 1 $O/bench_n1.err; head -c 300 $O/bench_n1.json
