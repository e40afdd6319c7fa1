#!/bin/bash
# First GPU call of a round (run through gpurun, ~60 s of box time): everything the GPU-less session of round 2 left unmeasured.
#   1. native A/B of the conv variants on the config-2 layer shapes (default 8449 vs persistent workgroups 24833 vs the older ones)
#   2. the two oracle-heavy GPU tests that have not run on hardware yet
#   3. the detector / fp16 native parity runners again (they take seconds)
# Build the runners first (container):  see the "Build:" line at the top of each scripts/microbench/native_*.cpp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/first
mkdir -p $O
B=scripts/microbench/_build
timeout 40 $B/native_conv_bench --variants 8449,24833,257,12289 > $O/conv_ab.log 2>&1; echo "rc=$?" >> $O/conv_ab.log
# experimental build (make -C megapose6d_amd/csrc variant NAME=exp DEFS=-DMP_CONV_EXPERIMENTS; cp _build/libmp_engine_exp.so
# _build/exp_lib/libmp_engine.so): s_setprio around the MFMA groups, alone (41217) and with persistent workgroups (57601)
if [ -f $B/exp_lib/libmp_engine.so ]; then
  LD_LIBRARY_PATH=$B/exp_lib timeout 40 $B/native_conv_bench --variants 8449,41217,57601 > $O/conv_ab_exp.log 2>&1; echo "rc=$?" >> $O/conv_ab_exp.log
fi
timeout 20 $B/native_f16_check > $O/f16.log 2>&1; echo "rc=$?" >> $O/f16.log
timeout 30 $B/native_detector_check tests/_build/detector_fixture_native.bin tests/_build/detector_fixture_resized.bin tests/_build/detector_fixture_batch2.bin > $O/detector.log 2>&1; echo "rc=$?" >> $O/detector.log
timeout 150 python -m pytest tests/test_gpu_zzzz_oracle_heavy.py -q -m gpu -p no:cacheprovider > $O/oracle_heavy.log 2>&1; echo "rc=$?" >> $O/oracle_heavy.log
tail -n 12 $O/conv_ab.log $O/oracle_heavy.log
