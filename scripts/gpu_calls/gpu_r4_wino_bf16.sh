#!/bin/bash
# round 4: the bf16x9 exact-piece Winograd kernel vs the fp32 Winograd kernel vs the direct kernel (parity + timing, 576 rows)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/wino
timeout 200 scripts/microbench/_build/native_wino_check > gpurun_out/wino/wino_bf16.log 2>&1; echo "rc=$?" >> gpurun_out/wino/wino_bf16.log
cat gpurun_out/wino/wino_bf16.log
