#!/bin/bash
# rocprofv3 passes over the default bench (1M-token prefill on one GPU); run on the GPU box from
# the repo root:  bash tools_diag/prof.sh <tag> [extra bench args]
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in a run of its own
# (TRACE_ONLY=1 stops after pass 1).
TAG=${1:-x}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $@"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
if [ "$TRACE_ONLY" = 1 ]; then cat $OUT/trace/*/*kernel_stats.csv | cut -c1-200 | head -5; exit 0; fi
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python3 $ARGS > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc FETCH_SIZE -- python3 $ARGS > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 $ARGS > $OUT/p3.log 2>&1 || exit 1
cd $OUT && python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if "lv_" in row["Kernel_Name"]:
            key = row["Kernel_Name"].split("(")[0] + "|" + row["Counter_Name"]
            agg[key] += float(row["Counter_Value"]); n[key] += 1
    for k in agg:
        out[k] = {"sum": agg[k], "dispatches": n[k], "per_dispatch": agg[k] / n[k]}
json.dump(out, open("pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
cat $OUT/trace/*/*kernel_stats.csv | cut -c1-200 | head -4
