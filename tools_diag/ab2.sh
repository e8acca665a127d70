#!/bin/bash
# Same-box A/B of library builds by plain bench runs: tools_diag/ab2.sh "<bench args>" <lib1> <lib2> ...
ARGS=$1; shift
L=long_vita_amd/liblvattn.so
cp $L /tmp/lv_orig.so
for f in "$@" default; do
  if [ $f = default ]; then cp /tmp/lv_orig.so $L; else cp long_vita_amd/$f $L; fi
  echo "== $f"
  timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])" || exit 1
done
cp /tmp/lv_orig.so $L
