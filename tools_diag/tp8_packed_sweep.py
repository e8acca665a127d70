"""Piece-length sweep for the packed 131,072-token batch of bench.py (cfg3-B packing) at 5 / 1 heads: does cutting the
long segments' sweeps help a launch that already has more work items than twice the compute units?
python tools_diag/tp8_packed_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops  # noqa: E402
import math  # noqa: E402

HQ, HKV, D = 5, 1, 128
rs = np.random.RandomState(1234)      # bench.py's cfg3_packed packing
lens = []
while sum(lens) < 131072:
    lens.append(int(math.exp(rs.uniform(math.log(256), math.log(32768)))))
cu = [0]
for L in lens:
    cu.append(min(131072, cu[-1] + L))
cu[-1] = 131072
cu = [int(x) for x in cu]
S = cu[-1]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
q, k, v, do = rnd(S, HQ, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HQ, D)
print("segments:", [b - a for a, b in zip(cu[:-1], cu[1:])])
print("planner:", ops.plan_split(cu, True, HQ, HKV, dev))
for pf, pb in ((0, 0), (0, 16384), (0, 8192), (0, 4096), (0, 2048), (8192, 0), (4096, 0), (8192, 8192), (4096, 4096), (4096, 2048)):
    ops.set_split_override(pf, pb)

    def step():
        q.grad = k.grad = v.grad = None
        out, _ = ops.varlen_attention(q, k, v, cu, D ** -0.5, True)
        out.backward(do)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"keys/piece {pf:5d} rows/piece {pb:5d}: {e0.elapsed_time(e1) / 6:8.3f} ms", flush=True)
ops.set_split_override(None, None)
