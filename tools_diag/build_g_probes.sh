#!/bin/bash
# Probe builds of the dK/dV g kernel: tools_diag/build_g_probes.sh "1 2 4 ..." -> long_vita_amd/liblvattn_gp<N>.so
# (timing only; a probe build computes wrong gradients).  Every object is rebuilt: a stale one must never be linked.
set -e
cd "$(dirname "$0")/../long_vita_amd/csrc"
rm -f build/g_probe_*.o ../liblvattn_gp*.so   # (bits: 1 exponentials, 2 barrier, 4 staging requests, 8 row reads, 16 transposed reads, 32 vector arithmetic, 64 row constants, 128 the counted vmcnt wait)
make -j8 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-slp-vectorize -Wno-uninitialized"
OTHERS=$(ls build/*.o | grep -v attn_bwd_dkdv_g.o | grep -v g_probe_)
for n in $1; do
  /opt/rocm/bin/hipcc $FLAGS -DG_PROBE=$n -c attn_bwd_dkdv_g.hip -o build/g_probe_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblvattn_gp$n.so $OTHERS build/g_probe_$n.o
done
ls -la ../liblvattn_gp*.so
