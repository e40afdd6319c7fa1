#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c26
mkdir -p $O
MP_PROF_DETAIL=1 timeout 200 python scripts/profile_layers.py > $O/layers.log 2>&1
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/scripts/microbench/_build/libmp_engine_noepi.so MP_PROF_DETAIL=1 timeout 200 python scripts/profile_layers.py > $O/layers_noepi.log 2>&1
