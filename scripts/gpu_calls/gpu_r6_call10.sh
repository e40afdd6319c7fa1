#!/bin/bash
# round 6, call 10: exchange pitch of the Winograd epilogue 72 (product) vs 64 floats (ws64), alternating; the texture tests with the
# anisotropic filter (contract v2.1) on the GPU.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c10
mkdir -p $O
B=scripts/microbench/_build
for rep in 1 2; do
  timeout 200 $B/native_wino_check > $O/ws72_$rep.log 2>&1; echo "rc=$?" >> $O/ws72_$rep.log
  LD_LIBRARY_PATH=$B/ws64 timeout 200 $B/native_wino_check > $O/ws64_$rep.log 2>&1; echo "rc=$?" >> $O/ws64_$rep.log
  echo "== pitch 72 ($rep)"; grep -E "CLK|ALL|FAIL|MISMATCH|rc=" $O/ws72_$rep.log | cut -c1-230
  echo "== pitch 64 ($rep)"; grep -E "CLK|ALL|FAIL|MISMATCH|rc=" $O/ws64_$rep.log | cut -c1-230
done
timeout 900 python -m pytest tests/test_gpu_textures.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "textur or raster" > $O/pytest_tex.log 2>&1; tail -n 3 $O/pytest_tex.log
