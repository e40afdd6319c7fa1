#!/bin/bash
# Round 3, GPU call 1: conv epilogue / persistent A/B (native, ~40 s), then the whole -m gpu suite with the O(1) synthetic networks.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c1
mkdir -p $O
B=scripts/microbench/_build
timeout 90 $B/native_conv_bench --variants 8449,73985,24833,90369 > $O/conv_ab.log 2>&1; echo "rc=$?" >> $O/conv_ab.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -n 30 $O/conv_ab.log
tail -n 40 $O/pytest_gpu.log
