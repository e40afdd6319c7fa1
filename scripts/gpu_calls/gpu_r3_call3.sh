#!/bin/bash
# Round 3, GPU call 3: Winograd kernel check + timing (native), raster phase breakdown (profiling build)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c3
mkdir -p $O
timeout 120 scripts/microbench/_build/native_wino_check > $O/wino.log 2>&1; echo "rc=$?" >> $O/wino.log
MP_ENGINE_LIB=scripts/microbench/_build/libmp_engine_prof.so timeout 150 python scripts/raster_phases.py > $O/raster_phases.log 2>&1; echo "rc=$?" >> $O/raster_phases.log
cat $O/wino.log; cat $O/raster_phases.log
