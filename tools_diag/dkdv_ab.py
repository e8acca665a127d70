"""Same-process A/B of the dK/dV kernels at 131,072 tokens (one causal segment, 40 / 8 heads): device time of the
backward launch per kernel choice, several rounds, plus a bit-for-bit run-to-run check and the difference between the
kernels' results.  Diagnostic; run on a GPU box:  python tools_diag/dkdv_ab.py [rounds] [seq]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import _lib, ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["rs", "g"]
HQ, HKV, D = 40, 8, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
q, k, v, do = rnd(S, HQ, D), rnd(S, HKV, D), rnd(S, HKV, D), rnd(S, HQ, D)
scale = D ** -0.5
dq_desc = ops.blocks_on_device([0, S], True, False, dev)
dk_desc = ops.blocks_on_device([0, S], True, True, dev)
out, lse = ops.attn_fwd_launch(q, k, v, dq_desc, scale, True)
res = {}
for rnd_i in range(rounds):
    for kind in kinds:
        _lib.check(_lib.lib().lvattn_select_bwd_kernels(b"pp", kind.encode()), "select")
        ops.attn_bwd_launch(q, k, v, out, do, lse, dq_desc, dk_desc, scale, True)       # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            dq_, dk_, dv_, _d = ops.attn_bwd_launch(q, k, v, out, do, lse, dq_desc, dk_desc, scale, True)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd_i} dkdv={kind:6s} backward (dq + dk/dv + delta) {e0.elapsed_time(e1) / 3:8.2f} ms", flush=True)
        if kind in res:
            assert torch.equal(res[kind][0], dk_) and torch.equal(res[kind][1], dv_), f"{kind}: not reproducible"
        res[kind] = (dk_.clone(), dv_.clone())
base = kinds[0]
for kind in kinds[1:]:
    for name, a, b in (("dk", res[base][0], res[kind][0]), ("dv", res[base][1], res[kind][1])):
        err = (a.float() - b.float()).abs().max().item() / a.float().abs().max().item()
        print(f"{kind} vs {base}: {name} max rel diff {err:.3e}", flush=True)
_lib.check(_lib.lib().lvattn_select_bwd_kernels(b"default", b"default"), "select")
