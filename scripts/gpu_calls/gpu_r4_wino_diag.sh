#!/bin/bash
# round 4: where the bf16x9 Winograd K loop spends its time -- the same kernel with pieces of its filler work removed (wrong results, timing only)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/wino
B=scripts/microbench/_build
mkdir -p $B/explib && cp $B/libmp_engine_exp.so $B/explib/libmp_engine.so
MP_WINO_DIAG_SWEEP=1 LD_LIBRARY_PATH=$B/explib timeout 200 $B/native_wino_check > gpurun_out/wino/wino_diag.log 2>&1; echo "rc=$?" >> gpurun_out/wino/wino_diag.log
grep -E "DIAG|TIME|rc=|FAIL|ALL" gpurun_out/wino/wino_diag.log
