#!/bin/bash
# Round 6, call 20: what a patch request of the Winograd K loop costs as a function of how its lanes spread over cache lines
# (scripts/microbench/patch_request_patterns.hip), three runs.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/call20
mkdir -p $O
for k in 1 2 3; do timeout 120 scripts/microbench/_build/patch_request_patterns > $O/patterns_$k.txt 2>&1; echo "rc=$?" >> $O/patterns_$k.txt; done
cat $O/patterns_1.txt
