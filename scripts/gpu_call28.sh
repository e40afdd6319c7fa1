#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c28
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
timeout 200 python scripts/bench_backbone.py --cin 27 --batch 576 > $O/bb.log 2>&1
