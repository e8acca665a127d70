#!/bin/bash
# rocprofv3 passes for the forward kernel at S=131072 (run on the GPU box from the repo root).
# usage: bash tools_prof.sh <tag>
set -e
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$R/bench.py --seq 131072 --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python3 $ARGS > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc FETCH_SIZE -- python3 $ARGS > $OUT/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 $ARGS > $OUT/p3.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/p4 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -- python3 $ARGS > $OUT/p4.log 2>&1
cd $OUT && python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if "lv_fwd_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"])
    print(f, dict(agg))
PY
