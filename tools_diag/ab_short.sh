#!/bin/bash
# Same-box A/B of the default forward kernel choice against the ping-pong kernel on packed and short workloads
# (run on the GPU box from the repo root; prints workload, kernel, ms per step, TFLOP/s).
for args in "--workload cfg3_packed" "--seq 8192" "--seq 32768" "--workload cfg3_fwd_bwd"; do
  for k in default pp default pp; do
    LVATTN_FWD_KERNEL=$k timeout -k 10 200 python bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', '$k', round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))" || exit 1
  done
done
