"""Piece-length sweep of the split sweeps at 5 / 1 heads: forward + backward ms per (keys per piece, query rows per piece).
python tools_diag/tp8_sweep.py <tokens>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
HQ, HKV, D = 5, 1, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
q, k, v, do = rnd(S, HQ, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HQ, D)
flops = 3.5 * 2.0 * D * HQ * float(S) ** 2
print("planner:", ops.plan_split([0, S], True, HQ, HKV, dev))
for pf in (0, 512, 1024, 2048, 4096, 8192):
    for pb in (0, 512, 1024, 2048, 4096, 8192):
        ops.set_split_override(pf, pb)

        def step():
            q.grad = k.grad = v.grad = None
            out, _ = ops.varlen_attention(q, k, v, [0, S], D ** -0.5, True)
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
        ms = e0.elapsed_time(e1) / 6
        print(f"S {S} keys/piece {pf:5d} rows/piece {pb:5d}: {ms:8.3f} ms {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
