"""CPU oracle for the attention hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package imports this module: it is the checker that
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` call, never the thing that is measured or shipped.

Two families of functions live here.

1. ``eager_attention`` restates the reference's eager PyTorch attention
   (lcvlm_modellink/core/transformer/dot_product_attention.py:148-288):
   GQA ``repeat_interleave`` (:168-172) -> ``baddbmm`` with
   ``alpha = 1/norm_factor`` (:206-212) -> Megatron's scale-mask-softmax in
   fp32 (mask filled with -10000, :241) -> ``bmm`` with V (:274) -> the
   ``[sq, b, np*hn]`` layout (:280-284).  It is pinned by
   ``tests/golden/eager_attention.npz``, which was produced by running the
   reference's own wrapper under an import shim (tests/golden/make_golden.py).
   It is also the CPU baseline that ``bench.py`` times.

2. float64 numpy restatements of what the reference delegates to kernels that
   are not in its tree: packed variable-length causal GQA attention with
   log-sum-exp (the ``npu_fusion_attention`` TND call,
   dot_product_attention.py:352-365), its gradients, the dense non-causal ViT
   call (:314-325), the block-pair attention + LSE merge of the zig-zag ring
   (``ringattn_context_parallel``, call site :70-71).  Those kernels
   (torch_npu / MindSpeed 341d549c) are absent from the reference tree, so their
   results are "parity unpinned" by the reference itself; the float64 functions
   below are anchored on the eager path above (same softmax, same GQA mapping,
   same scale) and on the published flash-attention / ring-attention algebra.
"""
from __future__ import annotations

import math

import numpy as np
import torch


# --------------------------------------------------------------------------
# 1. eager path (reference: dot_product_attention.py:148-288)
# --------------------------------------------------------------------------
def causal_mask(sq: int, sk: int, device="cpu") -> torch.Tensor:
    """Boolean mask, True = masked, shape [1, 1, sq, sk] (Megatron convention,
    what ``get_attention_mask()`` supplies at dot_product_attention.py:151-152)."""
    return torch.triu(torch.ones(sq, sk, dtype=torch.bool, device=device), diagonal=1 + sk - sq)[None, None]


def scale_mask_softmax(scores: torch.Tensor, mask: torch.Tensor | None) -> torch.Tensor:
    """Megatron FusedScaleMaskSoftmax, unfused branch with
    attention_softmax_in_fp32: masked_fill(mask, -10000.0) then fp32 softmax,
    result cast back to the input dtype (call site dot_product_attention.py:241)."""
    dt = scores.dtype
    x = scores.float()
    if mask is not None:
        x = x.masked_fill(mask, -10000.0)
    return torch.softmax(x, dim=-1).to(dt)


def eager_attention(query, key, value, attention_mask, norm_factor: float):
    """query [sq,b,np,hn], key/value [sk,b,ng,hn]  ->  [sq, b, np*hn].

    Follows dot_product_attention.py:168-286 step by step (no alibi, no
    softcapping, dropout p = 0 as in every shipped script)."""
    sq, b, np_, hn = query.shape
    sk, ng = key.shape[0], key.shape[2]
    rep = np_ // ng
    if rep > 1:                                                    # :170-172
        key = key.repeat_interleave(rep, dim=2)
        value = value.repeat_interleave(rep, dim=2)
    q2 = query.reshape(sq, b * np_, hn)                            # :195
    k2 = key.reshape(sk, b * np_, hn)                              # :197
    buf = torch.empty(b * np_, sq, sk, dtype=query.dtype)          # :201-203
    scores = torch.baddbmm(buf, q2.transpose(0, 1), k2.transpose(0, 1).transpose(1, 2),
                           beta=0.0, alpha=1.0 / norm_factor)      # :206-212
    scores = scores.view(b, np_, sq, sk)                           # :228
    probs = scale_mask_softmax(scores, attention_mask)             # :241
    v2 = value.reshape(sk, b * np_, hn)                            # :268
    ctx = torch.bmm(probs.view(b * np_, sq, sk), v2.transpose(0, 1))   # :274
    ctx = ctx.view(b, np_, sq, hn).permute(2, 0, 1, 3).contiguous()     # :277-280
    return ctx.view(sq, b, np_ * hn)                               # :283-284


def eager_attention_rows(query, key, value, row0: int, rows: int, heads, norm_factor: float):
    """The eager path restricted to query rows [row0, row0+rows) and the query
    heads in ``heads`` of a single causal sequence (b = 1).  Used by the CPU
    baseline to time a bounded sample of a long sequence: the [np, sq, sk]
    score matrix of the full problem does not fit in memory."""
    sq, b, np_, hn = query.shape
    ng = key.shape[2]
    rep = np_ // ng
    heads = list(heads)
    q = query[row0:row0 + rows, :, heads]                          # [rows,1,h,hn]
    kv_heads = [h // rep for h in heads]
    k = key[:, :, kv_heads]
    v = value[:, :, kv_heads]
    sk = key.shape[0]
    mask = (torch.arange(sk)[None, :] > (torch.arange(rows)[:, None] + row0))[None, None]
    return eager_attention(q, k, v, mask, norm_factor)


# --------------------------------------------------------------------------
# 2. float64 restatements of the delegated kernels
# --------------------------------------------------------------------------
def _f64(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        return x.detach().to(torch.float64).cpu().numpy()
    return np.asarray(x, dtype=np.float64)


def block_attention(q, k, v, scale: float, mask: np.ndarray | None):
    """One attention block.  q [Sq,Hq,D], k/v [Sk,Hkv,D]; ``mask`` is a boolean
    [Sq,Sk] array, True = masked.  Returns out [Sq,Hq,D] and lse [Hq,Sq]
    (natural log of sum exp of scaled scores; -inf for fully masked rows, whose
    output rows are zero)."""
    q, k, v = _f64(q), _f64(k), _f64(v)
    Sq, Hq, D = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    out = np.zeros((Sq, Hq, v.shape[2]))
    lse = np.full((Hq, Sq), -np.inf)
    for h in range(Hq):
        s = (q[:, h] @ k[:, h // rep].T) * scale
        if mask is not None:
            s = np.where(mask, -np.inf, s)
        m = s.max(axis=1)
        ok = np.isfinite(m)
        msafe = np.where(ok, m, 0.0)
        p = np.exp(s - msafe[:, None])
        l = p.sum(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            out[:, h] = np.where(ok[:, None], (p @ v[:, h // rep]) / np.where(ok, l, 1.0)[:, None], 0.0)
            lse[h] = np.where(ok, msafe + np.log(np.where(ok, l, 1.0)), -np.inf)
    return out, lse


def segment_mask(L: int, causal: bool, band: int | None = None):
    """Boolean [L, L] mask of one segment, True = masked: keys after the query when ``causal``; with ``band`` also keys
    more than ``band`` positions before it (what pre_tockens / --sliding-window ask of the fused op in its band modes,
    dot_product_attention.py:403-406, 454-467: a query sees itself and the ``band`` keys before it)."""
    if not causal:
        return None
    mask = np.triu(np.ones((L, L), dtype=bool), 1)
    if band is not None:
        mask |= np.tril(np.ones((L, L), dtype=bool), -int(band) - 1)
    return mask


def varlen_attention(q, k, v, cu_seqlens, scale: float, causal: bool = True, band: int | None = None):
    """Packed variable-length attention, one (causal) block per segment.

    q [T,Hq,D], k/v [T,Hkv,D] ("TND"); ``cu_seqlens`` = [0] + the reference's
    ``actual_seq_len`` cumulative end offsets
    (dot_product_attention.py:345-365: sparse_mode=2 + actual_seq_qlen).  ``band``: see ``segment_mask``."""
    q, k, v = _f64(q), _f64(k), _f64(v)
    T, Hq, D = q.shape
    out = np.zeros((T, Hq, v.shape[2]))
    lse = np.zeros((Hq, T))
    for a, b in zip(cu_seqlens[:-1], cu_seqlens[1:]):
        a, b = int(a), int(b)
        if b <= a:
            continue
        L = b - a
        mask = segment_mask(L, causal, band)
        o, s = block_attention(q[a:b], k[a:b], v[a:b], scale, mask)
        out[a:b] = o
        lse[:, a:b] = s
    return out, lse


def varlen_attention_grad(q, k, v, dout, cu_seqlens, scale: float, causal: bool = True, band: int | None = None):
    """float64 gradients of ``varlen_attention`` w.r.t. q, k, v for upstream
    gradient ``dout`` [T,Hq,D] (the ``npu_fusion_attention_grad`` the reference
    reaches through autograd).  Textbook softmax-attention backward:
    dV = P^T dO, dP = dO V^T, dS = P * (dP - rowsum(dO*O)), dQ = dS K * scale,
    dK = dS^T Q * scale; GQA sums dK/dV over the query heads of a group."""
    q, k, v, dout = _f64(q), _f64(k), _f64(v), _f64(dout)
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    for a, b in zip(cu_seqlens[:-1], cu_seqlens[1:]):
        a, b = int(a), int(b)
        L = b - a
        if L <= 0:
            continue
        mask = segment_mask(L, causal, band)
        for h in range(Hq):
            g = h // rep
            s = (q[a:b, h] @ k[a:b, g].T) * scale
            if mask is not None:
                s = np.where(mask, -np.inf, s)
            p = np.exp(s - s.max(axis=1, keepdims=True))
            p /= p.sum(axis=1, keepdims=True)
            o = p @ v[a:b, g]
            do = dout[a:b, h]
            dv[a:b, g] += p.T @ do
            dp = do @ v[a:b, g].T
            delta = (do * o).sum(axis=1, keepdims=True)
            ds = p * (dp - delta)
            dq[a:b, h] = ds @ k[a:b, g] * scale
            dk[a:b, g] += ds.T @ q[a:b, h] * scale
    return dq, dk, dv


def merge_blocks(o1, lse1, o2, lse2):
    """Combine two partial attention results over disjoint key sets.
    o [S,H,D], lse [H,S].  lse = log(exp(lse1)+exp(lse2)),
    o = o1*exp(lse1-lse) + o2*exp(lse2-lse)  (rows with lse = -inf contribute 0)."""
    o1, o2, lse1, lse2 = _f64(o1), _f64(o2), _f64(lse1), _f64(lse2)
    m = np.maximum(lse1, lse2)
    msafe = np.where(np.isfinite(m), m, 0.0)
    w1 = np.exp(lse1 - msafe)
    w2 = np.exp(lse2 - msafe)
    tot = w1 + w2
    with np.errstate(divide="ignore", invalid="ignore"):
        lse = np.where(tot > 0, msafe + np.log(np.where(tot > 0, tot, 1.0)), -np.inf)
        a1 = np.where(tot > 0, w1 / np.where(tot > 0, tot, 1.0), 0.0)
        a2 = np.where(tot > 0, w2 / np.where(tot > 0, tot, 1.0), 0.0)
    o = o1 * a1.T[:, :, None] + o2 * a2.T[:, :, None]
    return o, lse


def zigzag_chunks(cp_size: int, rank: int):
    """Chunk ids a rank owns: (r, 2*CP-1-r)  (training/utils.py:359-361)."""
    return rank, 2 * cp_size - 1 - rank


def ring_attention_simulated(q, k, v, scale: float, cp_size: int):
    """Single-process simulation of the zig-zag causal ring
    (``ringattn_context_parallel``, reached from dot_product_attention.py:70-71).

    q/k/v are the GLOBAL [S,H,D] tensors of one causal sequence.  Each rank
    holds chunks (r, 2CP-1-r) of the 2*CP equal chunks; at step i it sees the
    KV block of rank (r - i) mod CP and computes
      i == 0          : causal attention on its own [2c] tokens (block-diagonal
                        in chunk space: chunk r causal, chunk 2CP-1-r sees
                        chunk r fully and itself causally),
      kv from earlier rank (src < r): all local Q rows x first KV chunk, no mask,
      kv from later rank  (src > r): second-half Q rows x both KV chunks, no mask,
    merging the partial results with (out, lse).  Returns a list with every
    rank's local output [2c,H,D] and lse [H,2c], in zig-zag local order."""
    q, k, v = _f64(q), _f64(k), _f64(v)
    S = q.shape[0]
    c = S // (2 * cp_size)

    def local(x, r):
        a, b = zigzag_chunks(cp_size, r)
        return np.concatenate([x[a * c:(a + 1) * c], x[b * c:(b + 1) * c]], axis=0)

    results = []
    for r in range(cp_size):
        ql = local(q, r)
        out = np.zeros_like(ql)
        lse = np.full((q.shape[1], 2 * c), -np.inf)
        for step in range(cp_size):
            src = (r - step) % cp_size
            kl, vl = local(k, src), local(v, src)
            if step == 0:
                mask = np.triu(np.ones((2 * c, 2 * c), dtype=bool), 1)
                o, s = block_attention(ql, kl, vl, scale, mask)
                out, lse = merge_blocks(out, lse, o, s)
            elif src < r:
                o, s = block_attention(ql, kl[:c], vl[:c], scale, None)
                out, lse = merge_blocks(out, lse, o, s)
            else:
                o, s = block_attention(ql[c:], kl, vl, scale, None)
                o2, s2 = merge_blocks(out[c:], lse[:, c:], o, s)
                out[c:] = o2
                lse[:, c:] = s2
        results.append((out, lse))
    return results


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round a float array to the nearest bfloat16-representable values."""
    return torch.from_numpy(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def default_scale(head_dim: int) -> float:
    """1/sqrt(d): dot_product_attention.py:337 when scale_mask_softmax.scale is None."""
    return 1.0 / math.sqrt(head_dim)
