"""``npu_fusion_attention``-shaped entry point backed by the gfx950 kernels.

The reference calls ``torch_npu.npu_fusion_attention`` at
lcvlm_modellink/core/transformer/dot_product_attention.py:314-325 (ViT: SBH, no
mask, sparse_mode 0), :352-365 (LLM: TND, causal template mask, sparse_mode 2,
actual_seq_qlen/kvlen) and :454-467 (generic SBH / BNSD / TND).  This function takes
the same arguments and returns a tuple whose element 0 is the attention output in
the input layout; elements 1 and 2 are (softmax_max, softmax_sum) in the
decomposition max = log-sum-exp, sum = 1 (lse = max + ln(sum) as in the NPU op).

``npu_incre_flash_attention`` / ``npu_prompt_flash_attention`` stand in for the two ops of the
``--use-kv-cache`` branch (:428-451): one new token against the cache, and the prompt pass.
"""
from __future__ import annotations


import torch

from . import ops


def _is_causal(atten_mask, sparse_mode, pre_tockens, next_tockens):
    if atten_mask is None:
        if sparse_mode == 0:
            return False
        raise ValueError(f"sparse_mode={sparse_mode} needs an atten_mask")
    if sparse_mode in (2, 3, 4):
        # 2 = left-up causal, 3 = right-down causal (identical when q and kv lengths agree),
        # 4 = band with pre/next tokens; the reference only uses the causal band (next_tockens = 0).
        if sparse_mode == 4 and next_tockens not in (0,):
            raise NotImplementedError("band attention with next_tockens != 0 is not supported")
        return True
    if sparse_mode in (0, 1):
        if atten_mask is True:
            return True
        m = atten_mask
        if m.dim() >= 2 and m.shape[-1] == m.shape[-2]:
            n = m.shape[-1]
            tri = torch.triu(torch.ones(n, n, dtype=torch.bool, device=m.device), 1)
            if bool((m.reshape(-1, n, n)[0].bool() == tri).all()):
                return True
            if not bool(m.any()):
                return False
        raise NotImplementedError("only causal or empty masks are supported")
    raise NotImplementedError(f"sparse_mode={sparse_mode}")


def _band(causal, sparse_mode, pre_tockens, longest):
    """In its band modes (sparse_mode 4, or 0 / 1 with a mask) the NPU op limits a query to itself and the
    ``pre_tockens`` keys before it (dot_product_attention.py:403-406 sets pre_tockens from --sliding-window, :462 passes
    it); modes 2 / 3 ignore it (the compressed causal template).  Returns the band to apply, or None when no segment
    is long enough to feel it."""
    if causal and sparse_mode in (0, 1, 4) and pre_tockens is not None and 0 <= pre_tockens < longest - 1:
        return int(pre_tockens)
    return None


def npu_fusion_attention(query, key, value, head_num, input_layout, pse=None, padding_mask=None, atten_mask=None,
                         scale=1.0, keep_prob=1.0, pre_tockens=2147483647, next_tockens=2147483647, inner_precise=0,
                         prefix=None, actual_seq_qlen=None, actual_seq_kvlen=None, sparse_mode=0, gen_mask_parallel=True,
                         sync=False):
    if pse is not None:
        raise NotImplementedError("pse (alibi) is not supported")
    if padding_mask is not None or prefix is not None:
        raise NotImplementedError("padding_mask / prefix are not supported")
    if keep_prob != 1.0:
        raise NotImplementedError("attention dropout is not supported (the shipped scripts use 0)")
    causal = _is_causal(atten_mask, sparse_mode, pre_tockens, next_tockens)

    if input_layout == "TND":
        if actual_seq_qlen is None:
            raise ValueError("TND layout needs actual_seq_qlen")
        if actual_seq_kvlen is not None and list(actual_seq_kvlen) != list(actual_seq_qlen):
            raise NotImplementedError("different q / kv segmentations are not supported")
        cu = ops.cu_seqlens_from_actual_seq_len(actual_seq_qlen)
        band = _band(causal, sparse_mode, pre_tockens, max((b - a for a, b in zip(cu[:-1], cu[1:])), default=0))
        out, lse = ops.varlen_attention(query, key, value, cu, scale, causal, band)
        return out, lse, torch.ones_like(lse)

    if input_layout == "SBH":
        s, b, hd = query.shape
        d = hd // head_num
        hkv = key.shape[2] // d
        # [s, b, h*d] is [s rows, b*h heads, d] with row stride b*h*d: no copy, batch folds into heads
        q3 = query.reshape(s, b * head_num, d)
        k3 = key.reshape(key.shape[0], b * hkv, d)
        v3 = value.reshape(value.shape[0], b * hkv, d)
        if k3.shape[0] != s:
            raise NotImplementedError("q and kv lengths differ")
        out, lse = ops.varlen_attention(q3, k3, v3, [0, s], scale, causal, _band(causal, sparse_mode, pre_tockens, s))
        return out.reshape(s, b, hd), lse, torch.ones_like(lse)

    if input_layout in ("BSH", "BSND", "BNSD"):
        if input_layout == "BNSD":
            b, n, s, d = query.shape
            to_t = lambda x: x.permute(0, 2, 1, 3).reshape(b * s, x.shape[1], d)
        elif input_layout == "BSND":
            b, s, n, d = query.shape
            to_t = lambda x: x.reshape(b * s, x.shape[2], d)
        else:
            b, s, hd = query.shape
            d = hd // head_num
            to_t = lambda x: x.reshape(b * s, x.shape[2] // d, d)
        cu = [i * s for i in range(b + 1)]
        out, lse = ops.varlen_attention(to_t(query), to_t(key), to_t(value), cu, scale, causal, _band(causal, sparse_mode, pre_tockens, s))
        if input_layout == "BNSD":
            out = out.reshape(b, s, -1, d).permute(0, 2, 1, 3)
        elif input_layout == "BSND":
            out = out.reshape(b, s, -1, d)
        else:
            out = out.reshape(b, s, -1)
        return out, lse, torch.ones_like(lse)

    raise ValueError(f"unknown input_layout {input_layout!r}")


def _kv_heads(query, key, num_heads, num_key_value_heads):
    d = query.shape[-1] // num_heads
    hkv = num_key_value_heads or key.shape[-1] // d
    if hkv * d != key.shape[-1] or num_heads % hkv:
        raise ValueError(f"key width {key.shape[-1]} does not hold {hkv} heads of {d}")
    return d, hkv


def npu_incre_flash_attention(query, key, value, num_heads=1, input_layout="BSH", pse_shift=None, padding_mask=None,
                              atten_mask=None, scale_value=1.0, num_key_value_heads=0, actual_seq_lengths=None):
    """query [b, 1, hq*d], key / value [b, k_len, hkv*d] (any strides: the 'b s h' views the reference makes of its
    [s, b, h] cache are taken as they are) -> [b, 1, hq*d]; every cached key is visible (dot_product_attention.py:430-438)."""
    if input_layout != "BSH":
        raise NotImplementedError(f"input_layout {input_layout!r}: the reference calls this op with BSH only")
    if pse_shift is not None or padding_mask is not None or atten_mask is not None or actual_seq_lengths is not None:
        raise NotImplementedError("pse_shift / padding_mask / atten_mask / actual_seq_lengths are not supported")
    b, sq, hd = query.shape
    if sq != 1:
        raise ValueError("npu_incre_flash_attention takes one new token per sequence")
    d, hkv = _kv_heads(query, key, num_heads, num_key_value_heads)
    out = ops.decode_attention(query.reshape(b, num_heads, d), key.unflatten(2, (hkv, d)), value.unflatten(2, (hkv, d)),
                               scale_value)
    return out.reshape(b, 1, hd)


def npu_prompt_flash_attention(query, key, value, num_heads=1, input_layout="BSH", pse_shift=None, padding_mask=None,
                               atten_mask=None, scale_value=1.0, pre_tokens=2147483647, next_tokens=0, sparse_mode=0,
                               num_key_value_heads=0, actual_seq_lengths=None):
    """Prompt pass of the kv-cache branch (dot_product_attention.py:440-451): query [b, sq, hq*d], key / value
    [b, sk, hkv*d].  With a mask the attention is causal, the query block aligned to the END of the keys (the new
    tokens are the last sq positions of the cache); without one it is full."""
    if input_layout != "BSH":
        raise NotImplementedError(f"input_layout {input_layout!r}: the reference calls this op with BSH only")
    if pse_shift is not None or padding_mask is not None or actual_seq_lengths is not None:
        raise NotImplementedError("pse_shift / padding_mask / actual_seq_lengths are not supported")
    causal = _is_causal(atten_mask, sparse_mode if atten_mask is not None else 0, pre_tokens, next_tokens)
    b, sq, hd = query.shape
    sk = key.shape[1]
    d, hkv = _kv_heads(query, key, num_heads, num_key_value_heads)
    if sk < sq:
        raise ValueError("fewer keys than queries")
    band = _band(causal, sparse_mode if atten_mask is not None else 0, pre_tokens, sk)
    outs = []
    for i in range(b):      # one rectangle per sequence; the views keep the caller's strides
        q3, k3, v3 = query[i].unflatten(1, (num_heads, d)), key[i].unflatten(1, (hkv, d)), value[i].unflatten(1, (hkv, d))
        if band is not None:    # the band bounds every row's sweep: no need to cut the key range
            desc = ops.rect_band_blocks_on_device(0, sq, 0, sk, sk - sq, band, q3.device)
            outs.append(ops.attn_fwd_launch(q3, k3, v3, desc, scale_value, 2)[0].reshape(sq, hd))
            continue
        # a short new block against a long cache is cut along the keys so that it fills the chip (ops.attn_fwd_rect_split_keys)
        outs.append(ops.attn_fwd_rect_split_keys(q3, k3, v3, scale_value, causal, sk - sq)[0].reshape(sq, hd))
    return torch.stack(outs)
