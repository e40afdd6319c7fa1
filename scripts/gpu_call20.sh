#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c20
mkdir -p $O
timeout 200 python scripts/conv_slope.py > $O/slope.log 2>&1
