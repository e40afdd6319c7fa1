#!/bin/bash
# round 4, call 1: the exact-piece bf16 stem convolution against the fp32 direct kernel (parity on 6 small shapes, timing at 576 rows)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stem
timeout 200 scripts/microbench/_build/native_stem_check > gpurun_out/stem/stem.log 2>&1; echo "rc=$?" >> gpurun_out/stem/stem.log
cat gpurun_out/stem/stem.log
