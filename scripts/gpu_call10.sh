#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c10
mkdir -p $O
timeout 200 python scripts/conv_one_wg.py > $O/conv_2wg.log 2>&1
MP_CONV_LDS_PAD_KB=90 timeout 200 python scripts/conv_one_wg.py > $O/conv_1wg.log 2>&1
