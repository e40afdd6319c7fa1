#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c22
mkdir -p $O
P=$GRAFT_REPO_ROOT/scripts/microbench/_build/libmp_engine_prof.so
for cfg in "256 256" "256 64"; do MP_ENGINE_LIB=$P timeout 200 python scripts/conv_data_power.py $cfg >> $O/data_clk.log 2>&1; done
