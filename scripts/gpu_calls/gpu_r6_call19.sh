#!/bin/bash
# Round 6, call 19: repeatability of the GPU tier on the final tree -- the driver's command (-x) three times on one box, smoke after each.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/call19
mkdir -p $O
for k in 1 2 3; do
  timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu_$k.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_$k.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_$k.log 2>&1; echo "rc=$?" >> $O/smoke_$k.log
  tail -n 2 $O/pytest_gpu_$k.log; tail -n 2 $O/smoke_$k.log
done
