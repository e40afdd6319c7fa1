#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2c25
timeout 60 scripts/microbench/_build/occ_probe > gpurun_out/r2c25/occ.log 2>&1
