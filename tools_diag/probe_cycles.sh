#!/bin/bash
# Cycles (not wall time: probe builds change the chip's clock) of the forward kernel for a list of library builds.
# tools_diag/probe_cycles.sh "<lib> ..." [seq]   -- run on the GPU box from the repo root
LIBS=$1; SEQ=${2:-131072}
R=$GRAFT_REPO_ROOT
L=$R/long_vita_amd/liblvattn.so
cp $L /tmp/lv_orig.so
export TMPDIR=/tmp
cd /tmp
for lib in $LIBS; do
  [ "$lib" = liblvattn.so ] && cp /tmp/lv_orig.so $L || cp $R/long_vita_amd/$lib $L
  rm -rf /tmp/pc_out
  timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc_out --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum -- python3 $R/bench.py ${BENCH_ARGS:---seq $SEQ} --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline > /tmp/pc.log 2>&1 || { cp /tmp/lv_orig.so $L; tail -5 /tmp/pc.log; exit 1; }
  python3 - "$lib" <<'PY'
import csv, glob, sys, collections
import os
names = sorted({row["Kernel_Name"].split("(")[0] for f in glob.glob("/tmp/pc_out/**/*kernel_trace.csv", recursive=True) for row in csv.DictReader(open(f))
                if any(k in row["Kernel_Name"] for k in os.environ.get("KFILTER", "lv_fwd").split(","))})
for name in names:
  agg = collections.defaultdict(list)
  for f in glob.glob("/tmp/pc_out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].split("(")[0] == name:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
  dur = []
  for f in glob.glob("/tmp/pc_out/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].split("(")[0] == name:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
  print(name[:60], end="  ")
  g = agg["GRBM_GUI_ACTIVE"]; m = agg["SQ_VALU_MFMA_BUSY_CYCLES"]
  cyc = sum(g) / len(g) / 8
  h = sum(agg["TCC_HIT_sum"]); mi = sum(agg["TCC_MISS_sum"])
  wc = max(1.0, sum(agg["SQ_WAVE_CYCLES"])); wi = sum(agg["SQ_WAIT_INST_ANY"]) / wc; wa = sum(agg["SQ_WAIT_ANY"]) / wc; ac = sum(agg["SQ_ACTIVE_INST_ANY"]) / wc
  ms = sum(dur) / len(dur)
  print(f"{sys.argv[1]:24s} Mcycles/XCD {cyc/1e6:9.2f}  ms {ms:8.3f}  GHz {cyc/ms/1e6:5.3f}  mfma_busy {sum(m)/len(m)/1024/cyc:5.3f}  L2 hit {h/max(1.0,h+mi):5.3f}  wait_inst {wi:5.3f} wait_any {wa:5.3f} active {ac:5.3f} launches {len(g)}")
PY
done
cp /tmp/lv_orig.so $L
