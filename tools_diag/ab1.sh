#!/bin/bash
# Same-box A/B of library builds: tools_diag/ab1.sh <workload> <lib1> <lib2> ...  (each a copy of liblvattn.so under long_vita_amd/)
WL=$1; shift
L=long_vita_amd/liblvattn.so
cp $L /tmp/lv_orig.so
for f in "$@"; do
  cp long_vita_amd/$f $L
  echo "== $f"
  TRACE_ONLY=1 bash tools_diag/prof.sh ab_$f --workload $WL | grep "lv_" | cut -d, -f1-4 || exit 1
done
cp /tmp/lv_orig.so $L
echo "== default"
TRACE_ONLY=1 bash tools_diag/prof.sh ab_default --workload $WL | grep "lv_" | cut -d, -f1-4
