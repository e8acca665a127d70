"""Sampled rows of the attention gradients in float64.  TEST INFRASTRUCTURE ONLY.

``oracle.attention_ref.varlen_attention_grad`` materialises a whole [L, L] score matrix per head: fine for the small
parity cases, out of reach at the 131,072- and 1,048,576-token sizes the benchmarks run.  A single ROW of a gradient is
cheap at any length, though:

  dq[i, h]  needs the softmax row i of head h:                      O(L d)
  dk[j, g], dv[j, g]  need column j of P for the group's heads, i.e. the log-sum-exp and delta = <dO_i, O_i> of every
  later row i of the segment: one float64 forward over those rows (``row_stats``), then O((L - j) d group) per key.

The functions are plain torch float64 and run on whatever device the tensors live on (the full-size tests keep the
operands on the GPU and use its float64 units as a calculator; tests/test_oracle_golden.py pins them on the CPU against
``varlen_attention_grad``).  Same algebra as the textbook backward restated there: dV = P^T dO, dP = dO V^T,
dS = P * (dP - delta), dQ = scale dS K, dK = scale dS^T Q, causal inside a segment
(the autograd of the reference's TND call, lcvlm_modellink/core/transformer/dot_product_attention.py:352-365).
"""
from __future__ import annotations

import torch


def row_stats(q, k, v, do, heads, seg, rows, scale: float, chunk: int = 2048):
    """float64 log-sum-exp and delta = <dO_i, O_i> of the causal rows ``rows`` = (r0, r1) of segment ``seg`` = (a, b)
    (a <= r0 < r1 <= b, global row indices) for the query heads ``heads``.  q/do [T, Hq, D], k/v [T, Hkv, D].
    Returns (lse, delta), each [len(heads), r1 - r0].  (``saved_delta`` gives the other delta: from a SAVED output.)"""
    a, b = seg
    r0, r1 = rows
    rep = q.shape[1] // k.shape[1]
    lse = torch.empty((len(heads), r1 - r0), dtype=torch.float64, device=q.device)
    delta = torch.empty_like(lse)
    for n, h in enumerate(heads):
        g = h // rep
        for c0 in range(r0, r1, chunk):
            c1 = min(r1, c0 + chunk)
            kk = k[a:c1, g].double()
            s = (q[c0:c1, h].double() @ kk.T) * scale
            col = torch.arange(a, c1, device=q.device)[None, :]
            row = torch.arange(c0, c1, device=q.device)[:, None]
            s.masked_fill_(col > row, float("-inf"))
            l = torch.logsumexp(s, dim=1)
            p = torch.exp(s - l[:, None])
            o = p @ v[a:c1, g].double()
            lse[n, c0 - r0:c1 - r0] = l
            delta[n, c0 - r0:c1 - r0] = (o * do[c0:c1, h].double()).sum(-1)
    return lse, delta


def saved_delta(out, do, heads, rows):
    """delta = <dO_i, O_i> from a SAVED forward output ``out`` [T, Hq, D] (what a fused backward reads: the forward's
    result in its storage precision, here bf16) instead of the exact float64 O: [len(heads), r1 - r0].  The two deltas
    differ by the rounding of O; on a short row with one dominant key, where dP - delta nearly cancels, that rounding
    is most of the row's gradient error, so the tight per-row comparisons use this one."""
    r0, r1 = rows
    return torch.stack([(out[r0:r1, h].double() * do[r0:r1, h].double()).sum(-1) for h in heads])


def dq_row(q, k, v, do, i: int, h: int, seg, scale: float, out_saved=None):
    """float64 dq[i, h] ([D]) of the causal segment ``seg`` = (a, b) that holds row i; also returns (lse, out_row).
    With ``out_saved`` [T, Hq, D], delta comes from that saved output row instead of the exact one."""
    a, _ = seg
    g = h // (q.shape[1] // k.shape[1])
    kk, vv = k[a:i + 1, g].double(), v[a:i + 1, g].double()
    s = (kk @ q[i, h].double()) * scale
    l = torch.logsumexp(s, dim=0)
    p = torch.exp(s - l)
    o = p @ vv
    d_o = do[i, h].double()
    delta = (d_o * (o if out_saved is None else out_saved[i, h].double())).sum()
    ds = p * (vv @ d_o - delta)
    return (ds @ kk) * scale, l, o


def dkdv_row(q, k, v, do, j: int, g: int, seg, lse, delta, stats_row0: int, scale: float):
    """float64 (dk[j, g], dv[j, g]) of key j of KV head g inside the causal segment ``seg`` = (a, b): sums over the rows
    j..b-1 of the group's query heads.  ``lse`` / ``delta`` [rep, n] are ``row_stats`` of the group's heads (in order)
    for rows starting at ``stats_row0`` <= j and reaching b."""
    _, b = seg
    rep = q.shape[1] // k.shape[1]
    kj, vj = k[j, g].double(), v[j, g].double()
    dk = torch.zeros_like(kj)
    dv = torch.zeros_like(vj)
    off = j - stats_row0
    for n in range(rep):
        h = g * rep + n
        qq, dd = q[j:b, h].double(), do[j:b, h].double()
        p = torch.exp((qq @ kj) * scale - lse[n, off:off + b - j])
        dv += p @ dd
        ds = p * (dd @ vj - delta[n, off:off + b - j])
        dk += (ds @ qq) * scale
    return dk, dv


def row_excess(got, ref, rtol: float = 2e-2, floor: float = 0.25, tiny: float = 0.0) -> float:
    """Per-element tolerance of ONE row: max |got - ref| / (rtol |ref| + rtol floor max|ref| + tiny); <= 1 passes.  The
    floor is relative to the row's own largest element (the tensor-level tests use the whole tensor's); ``tiny`` is an
    absolute allowance for rows whose exact value is zero (dq of a segment's first row: one key, P = 1, dS = 0)."""
    got, ref = got.double(), ref.double()
    bound = rtol * ref.abs() + rtol * floor * ref.abs().max() + tiny
    return float(((got - ref).abs() / bound).max())
