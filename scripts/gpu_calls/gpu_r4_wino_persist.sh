#!/bin/bash
# round 4: persistent workgroups in the bf16x9 Winograd kernel (look-ahead runs on into the next unit) vs one workgroup per unit vs the
# previous (non-persistent, one-pass epilogue) kernel, all on the same box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/wino
B=scripts/microbench/_build
for rep in 1 2; do
timeout 200 $B/native_wino_check > gpurun_out/wino/wino_persist1.log 2>&1; echo "rc=$?" >> gpurun_out/wino/wino_persist1.log
MP_WINO_PERSIST=0 timeout 200 $B/native_wino_check > gpurun_out/wino/wino_persist0.log 2>&1; echo "rc=$?" >> gpurun_out/wino/wino_persist0.log
LD_LIBRARY_PATH=$B/oldlib timeout 200 $B/native_wino_check > gpurun_out/wino/wino_old.log 2>&1; echo "rc=$?" >> gpurun_out/wino/wino_old.log
echo "== persistent"; grep -E "bf16x9 wino|CLK|ALL|FAIL|rc=" gpurun_out/wino/wino_persist1.log | cut -c1-170
echo "== one workgroup per unit"; grep -E "bf16x9 wino|CLK|ALL|FAIL|rc=" gpurun_out/wino/wino_persist0.log | cut -c1-170
echo "== previous kernel"; grep -E "bf16x9 wino|CLK|ALL|FAIL|rc=" gpurun_out/wino/wino_old.log | cut -c1-170
done
