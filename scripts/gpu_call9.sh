#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c9
mkdir -p $O
R="$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_textures.py -x -q -k "raster or textured or crop" > $O/pytest_raster.log 2>&1; echo "rc=$?" >> $O/pytest_raster.log
timeout 200 python scripts/bench_raster.py 1 17 > $O/raster.log 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-trace -d $R/$O/pmc1 -o p --output-format csv -- python $R/scripts/bench_raster.py 17 > $R/$O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $R/$O/pmc2 -o p --output-format csv -- python $R/scripts/bench_raster.py 17 > $R/$O/pmc2.log 2>&1
cd $R
find $O -name "*.csv" -size +20M -delete
