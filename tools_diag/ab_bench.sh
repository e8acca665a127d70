#!/bin/bash
# One A/B arm: the bench lines of three workloads (forward kernel forced to ping-pong so both arms run the same kernels).
export LVATTN_FWD_KERNEL=pp
for w in cfg3_fwd_bwd cfg3_packed cfg2_vit; do
  python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],3), 'ms', round(d['roofline']['achieved'],1), 'TF')" || exit 1
done
