"""Forward + backward at one tensor-parallel rank's head count (5 query heads / 1 KV head: the reference's TP = 8 layout)
on short batches, with the sweeps left whole and with the planner's split: ms and algorithmic TFLOP/s per shape.
Diagnostic; run on a GPU box:  python tools_diag/tp8_shapes.py"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops  # noqa: E402

HQ, HKV, D = 5, 1, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)


def packed(S):
    rs = np.random.RandomState(1234)
    lens = []
    while sum(lens) < S:
        lens.append(int(math.exp(rs.uniform(math.log(256), math.log(32768)))))
    cu = [0]
    for L in lens:
        cu.append(min(S, cu[-1] + L))
    cu[-1] = S
    return cu


shapes = [("8K segment", [0, 8192]), ("16K segment", [0, 16384]), ("32K segment", [0, 32768]), ("128K packed (cfg3-B)", packed(131072)),
          ("32K in 4K segments", list(range(0, 32768 + 1, 4096)))]
for name, cu in shapes:
    S = cu[-1]
    q, k, v, do = rnd(S, HQ, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HQ, D)
    flops = 3.5 * sum(2.0 * D * HQ * float(b - a) ** 2 for a, b in zip(cu[:-1], cu[1:]))
    scale = D ** -0.5
    line = f"{name:24s} pieces {ops.plan_split(cu, True, HQ, HKV, dev)}:"
    for tag, over in (("whole", (0, 0)), ("split", (None, None))):
        ops.set_split_override(*over)

        def step():
            q.grad = k.grad = v.grad = None
            out, _ = ops.varlen_attention(q, k, v, cu, scale, True)
            out.backward(do)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        line += f"  {tag} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TFLOP/s"
    print(line, flush=True)
ops.set_split_override(None, None)
