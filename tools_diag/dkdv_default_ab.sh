#!/bin/bash
# Same-box A/B of the dK/dV kernel choice over the bench workloads: tools_diag/dkdv_default_ab.sh [rounds]
# (alternating arms; prints ms per step of each run)
R=${1:-2}
for r in $(seq $R); do
  for wl in "cfg3_fwd_bwd" "cfg3_packed" "cfg3_packed --heads 5,1"; do
    for k in rs g; do
      LVATTN_DKDV_KERNEL=$k timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
      python - "$wl" $k <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1]:40s} dkdv={sys.argv[2]:3s} {d['ms_per_step']:9.3f} ms  {d['roofline']['achieved']:7.1f} TFLOP/s", flush=True)
PY
    done
  done
done
