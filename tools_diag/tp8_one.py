"""One shape of tools_diag/tp8_shapes.py, for a kernel trace: python tools_diag/tp8_one.py <tokens> [whole]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
if len(sys.argv) > 2 and sys.argv[2] == "whole":
    ops.set_split_override(0, 0)
HQ, HKV, D = 5, 1, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
q, k, v, do = rnd(S, HQ, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HQ, D)
for _ in range(12):
    q.grad = k.grad = v.grad = None
    out, _ = ops.varlen_attention(q, k, v, [0, S], D ** -0.5, True)
    out.backward(do)
torch.cuda.synchronize()
