#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c14
mkdir -p $O
timeout 120 scripts/microbench/_build/conv_loop_parts 8 > $O/loop_parts_v2.log 2>&1
