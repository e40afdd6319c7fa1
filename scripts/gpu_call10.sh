#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c10
mkdir -p $O
P=$GRAFT_REPO_ROOT/scripts/microbench/_build/libmp_engine_prof.so
MP_ENGINE_LIB=$P MP_CONV_LDS_PAD_KB=20 timeout 200 python scripts/conv_one_wg.py > $O/seg_1wg.log 2>&1
MP_ENGINE_LIB=$P timeout 200 python scripts/conv_one_wg.py > $O/seg_2wg.log 2>&1
