#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c14
mkdir -p $O
for mb in 1 8 64; do timeout 120 scripts/microbench/_build/conv_loop_parts $mb > $O/loop_parts_$mb.log 2>&1; done
