"""Measured log-sum-exp error of the forward kernels on the hard-numerics inputs of tests/test_gpu_attention.py
(rising / falling scores, spiked keys, scores in the thousands), per forward kernel: prints max |lse - float64| next
to max |lse| so that the tests' bounds can be set from data.  Diagnostic; run on a GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import _lib, ops   # noqa: E402
from oracle import attention_ref as A  # noqa: E402


def report(tag, lse, ref):
    err = np.abs(lse.cpu().numpy().astype(np.float64) - ref)
    a = np.abs(ref)
    rel = (err / np.maximum(a, 1.0)).max()
    print(f"{tag:60s} max|lse| {a.max():9.2f}  max err {err.max():.3e}  max err/max(1,|lse|) {rel:.3e}  "
          f"in 2^-23|lse| units {(err / (np.maximum(a, 1.0) * 2.0 ** -23)).max():7.1f}", flush=True)


for kern in ("default", "a4", "pp16", "lockstep", "pp"):
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(kern.encode()), "select")
    for D, T in ((128, 3000), (64, 1500)):
        g = torch.Generator().manual_seed(5)
        u = torch.randn(D, generator=g)
        u = u / u.norm() * (D ** 0.5)
        ramp = torch.linspace(0.0, 6.0, T)[:, None, None]
        q = (torch.randn(T, 4, D, generator=g) * 0.2 + 1.5 * u).to(torch.bfloat16)
        v = torch.randn(T, 2, D, generator=g).to(torch.bfloat16)
        scale = A.default_scale(D)
        for sign in (1.0, -1.0):
            k = (torch.randn(T, 2, D, generator=g) * 0.2 + sign * ramp * u).to(torch.bfloat16)
            for causal in (True, False):
                _, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, causal)
                _, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), [0, T], scale, causal)
                report(f"{kern} rising/falling D={D} sign={sign:+.0f} causal={causal}", lse, ref_lse)
    T, Hq, Hkv, D = 2048, 4, 2, 128
    g = torch.Generator().manual_seed(77)
    u = torch.randn(D, generator=g)
    u = u / u.norm() * (D ** 0.5)
    q = torch.randn(T, Hq, D, generator=g) * 0.3 + 2.0 * u
    k = torch.randn(T, Hkv, D, generator=g) * 0.3
    v = torch.randn(T, Hkv, D, generator=g)
    k[500] = 3.0 * u
    k[1500, 0] = 6.0 * u
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    scale = A.default_scale(D)
    for causal in (True, False):
        _, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, causal)
        _, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), [0, T], scale, causal)
        report(f"{kern} spikes causal={causal}", lse, ref_lse)
    for D, T in ((128, 1500), (64, 700)):
        g = torch.Generator().manual_seed(9 + D)
        q = torch.randn(T, 4, D, generator=g) * 12.0
        k = torch.randn(T, 2, D, generator=g) * 12.0
        v = torch.randn(T, 2, D, generator=g)
        q[T // 3: T // 3 + 100] = 0.0
        q[5] = 0.0
        q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        scale = A.default_scale(D)
        for causal in (True, False):
            _, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, causal)
            _, lse = ops.attn_fwd_launch(q.cuda(), k.cuda(), v.cuda(), ops.blocks_on_device([0, T], causal, False, "cuda"), scale, causal)
            report(f"{kern} extreme D={D} causal={causal}", lse, ref_lse)
_lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")
