#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c24
mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
