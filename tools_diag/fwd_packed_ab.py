#!/usr/bin/env python3
"""Forward-only timing of packed causal attention (40/8 heads, d = 128, 131,072 tokens) under each forward kernel,
alternating arms on one box: which kernel should a launch made of SHORT segments take?
Prints one line per (packing, kernel): mean milliseconds over the timed passes."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops, _lib  # noqa: E402

HQ, HKV, D, S = 40, 8, 128, 131072
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
q = (torch.randn(S, HQ, D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
k = (torch.randn(S, HKV, D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
v = (torch.randn(S, HKV, D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty_like(q)
lse = torch.empty((HQ, S), dtype=torch.float32, device=dev)
scale = 1.0 / math.sqrt(D)


def log_uniform():
    rs = np.random.RandomState(1234)
    lens = []
    while sum(lens) < S:
        lens.append(int(math.exp(rs.uniform(math.log(256), math.log(32768)))))
    cu = [0]
    for L in lens:
        cu.append(min(S, cu[-1] + L))
    cu[-1] = S
    return cu


packings = {"log_uniform_256_32768": log_uniform()}
for seg in (1024, 2048, 4096, 8192, 16384):
    packings[f"uniform_{seg}"] = list(range(0, S + 1, seg))

for name, cu in packings.items():
    res = {}
    for rnd in range(3):
        for kern in ("pp",) if os.environ.get("LV_AB_PP_ONLY") else ("a4", "pp", "default"):
            _lib.check(_lib.lib().lvattn_select_fwd_kernel(kern.encode()), "select")
            desc = ops.blocks_on_device(cu, True, False, dev)
            ops.attn_fwd_launch(q, k, v, desc, scale, True, out, lse)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.attn_fwd_launch(q, k, v, desc, scale, True, out, lse)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(kern, []).append(e0.elapsed_time(e1) / 5)
    lens = [b - a for a, b in zip(cu[:-1], cu[1:])]
    flops = sum(2.0 * D * HQ * L * L for L in lens)
    if os.environ.get("LV_AB_PP_ONLY"):
        pp = min(res["pp"])
        print(f"{name:28s} segs={len(lens):4d} pp {pp:8.3f} ms ({flops / pp / 1e9:7.1f} TF)", flush=True)
        continue
    a4, pp, df = min(res["a4"]), min(res["pp"]), min(res["default"])
    print(f"{name:28s} segs={len(lens):4d} a4 {a4:8.3f} ms ({flops / a4 / 1e9:7.1f} TF)  pp {pp:8.3f} ms ({flops / pp / 1e9:7.1f} TF)  a4/pp {a4 / pp:.3f}"
          f"  library choice {df:8.3f} ms  ", flush=True)
_lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")
