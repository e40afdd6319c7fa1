#!/bin/bash
# round 6, call 3: (a) what an instruction costs in the gap between two MFMAs of one wave per SIMD, by kind and count
# (scripts/microbench/mfma_gap_fillers.hip) + exactness of the candidate round-to-nearest split; (b) the DIAG sweep of the scalar K loop
# with the finer switches (32 no V writes, 64 no transform arithmetic, 128 no patch requests).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c3
mkdir -p $O
B=scripts/microbench/_build
timeout 300 $B/mfma_gap_fillers > $O/gap.log 2>&1; echo "rc=$?" >> $O/gap.log
cat $O/gap.log
MP_WINO_DIAG_SWEEP=1 LD_LIBRARY_PATH=$B/exp timeout 300 $B/native_wino_check > $O/diag.log 2>&1; echo "rc=$?" >> $O/diag.log
echo "== diag sweep (scalar loop)"; grep -E "DIAG|rc=" $O/diag.log | cut -c1-60,195-300
