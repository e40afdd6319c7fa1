#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c17
mkdir -p $O
timeout 120 scripts/microbench/_build/conv_loop_dma > $O/loop_dma.log 2>&1
