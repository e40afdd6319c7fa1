#!/bin/bash
# round 4: PMC passes over the rasteriser micro-benchmark (pipeline-like poses, 4x MSAA): instruction mix, stall and LDS counters per wave
cd "$GRAFT_REPO_ROOT"
R="$GRAFT_REPO_ROOT"
O=gpurun_out/r4pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/$O/p1 -o p --output-format csv -- python $R/scripts/bench_raster.py 17 > $R/$O/p1.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/$O/p2 -o p --output-format csv -- python $R/scripts/bench_raster.py 17 > $R/$O/p2.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace -d $R/$O/p3 -o p --output-format csv -- python $R/scripts/bench_raster.py 17 > $R/$O/p3.log 2>&1
tail -n 3 $R/$O/p3.log
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r4pmc/p1", "gpurun_out/r4pmc/p2", "gpurun_out/r4pmc/p3"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "raster" not in k: continue
            k = "raster_tiles" if "raster_tiles" in k else "raster_bin"
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
        for k in acc:
            print(d, k, "dispatches", len(n[k]), {c: round(v / len(n[k])) for c, v in acc[k].items()})
PY
find $O -name "*.csv" -size +5M -delete
