#!/bin/bash
# Same-box A/B of two library builds under any command: tools_diag/ab_lib.sh <other .so under long_vita_amd/> <rounds> <command...>
# Each round runs the command with the in-tree library, then with the other one copied in its place; the original is restored.
OTHER=$1; ROUNDS=$2; shift 2
L=long_vita_amd/liblvattn.so
cp $L /tmp/lv_orig.so
trap 'cp /tmp/lv_orig.so $L' EXIT
for r in $(seq $ROUNDS); do
  cp /tmp/lv_orig.so $L; echo "== round $r: in-tree build"; timeout -k 10 300 "$@" || exit 1
  cp long_vita_amd/$OTHER $L; echo "== round $r: $OTHER"; timeout -k 10 300 "$@" || exit 1
done
