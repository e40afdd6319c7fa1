#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/wino
timeout 120 scripts/microbench/_build/native_wino_check > gpurun_out/wino/wino.log 2>&1; echo "rc=$?" >> gpurun_out/wino/wino.log
cat gpurun_out/wino/wino.log
