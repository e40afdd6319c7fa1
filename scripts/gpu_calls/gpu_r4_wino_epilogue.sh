#!/bin/bash
# round 4: bf16x9 Winograd epilogue with its runtime switches hoisted (four straight-line copies, ReLU as a maximum): parity + timing through
# native_wino_check, new library then the previous one (scripts/microbench/_build/libmp_engine_base.so) on the same box, twice
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wino_epi
mkdir -p $O
cp megapose6d_amd/libmp_engine.so /tmp/new.so
for r in 1 2; do
  for v in new base; do
    if [ $v = new ]; then cp /tmp/new.so megapose6d_amd/libmp_engine.so; else cp scripts/microbench/_build/libmp_engine_base.so megapose6d_amd/libmp_engine.so; fi
    timeout 200 scripts/microbench/_build/native_wino_check > $O/$v$r.log 2>&1; echo "rc=$?" >> $O/$v$r.log
    echo "== $v run $r"; grep -E "^TIME.*bf16x9|^PHASE|ALL OK|rc=|FAIL" $O/$v$r.log | cut -c1-200
  done
done
cp /tmp/new.so megapose6d_amd/libmp_engine.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "winograd" -q -p no:cacheprovider 2>&1 | tail -2
