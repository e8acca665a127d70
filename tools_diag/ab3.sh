#!/bin/bash
# Same-box A/B of library builds x kernel selections: tools_diag/ab3.sh "<bench args>" "<lib:kernel:waves> ..." [rounds]
ARGS=$1; VARS=$2; ROUNDS=${3:-2}
L=long_vita_amd/liblvattn.so
cp $L /tmp/lv_orig.so
for r in $(seq $ROUNDS); do
  for v in $VARS; do
    IFS=: read lib kern waves <<< "$v"
    [ "$lib" = liblvattn.so ] && cp /tmp/lv_orig.so $L || cp long_vita_amd/$lib $L
    LVATTN_FWD_KERNEL=$kern LVATTN_FWD_WAVES=$waves timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))" || { cp /tmp/lv_orig.so $L; exit 1; }
  done
done
cp /tmp/lv_orig.so $L
