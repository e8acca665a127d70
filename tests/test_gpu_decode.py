"""Incremental attention over a KV cache (the --use-kv-cache branch of the reference's flash path) against the
float64 oracle.  The NPU ops it stands in for are not in the reference tree: parity unpinned beyond the
softmax-attention definition the eager path pins.  Tolerances: output 2e-2 of the largest magnitude, LSE 1e-3."""
import math
import types

import numpy as np
import pytest
import torch

from oracle import attention_ref as A

pytestmark = pytest.mark.gpu
REL_TOL, LSE_TOL = 2e-2, 1e-3


def _rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def _check(out, lse, q, k, v, scale):
    b = q.shape[0]
    for i in range(b):
        ref_o, ref_l = A.block_attention(q[i][None], k[i], v[i], scale, None)     # q row [1, hq, d]
        got = out[i].float().cpu().numpy().astype(np.float64)
        assert np.abs(got - ref_o[0]).max() <= REL_TOL * np.abs(ref_o).max()
        if lse is not None:
            assert np.abs(lse[i].cpu().numpy() - ref_l[:, 0]).max() <= LSE_TOL


CASES = [
    # b, hq, hkv, d, k_len, n_split
    (1, 40, 8, 128, 1, 0),
    (1, 40, 8, 128, 5, 0),
    (1, 40, 8, 128, 127, 0),
    (1, 40, 8, 128, 128, 0),
    (1, 40, 8, 128, 129, 0),
    (1, 40, 8, 128, 1000, 3),
    (1, 40, 8, 128, 40000, 0),
    (2, 40, 8, 128, 3001, 7),
    (3, 16, 16, 64, 777, 0),
    (1, 32, 1, 128, 2500, 0),      # a full 32-row group
    (2, 6, 2, 64, 4096, 1),        # one split walks every tile
]


@pytest.mark.parametrize("b,hq,hkv,d,k_len,n_split", CASES)
def test_decode_matches_oracle(b, hq, hkv, d, k_len, n_split):
    from long_vita_amd import ops
    q, k, v = _rnd(b, hq, d, seed=1), _rnd(b, k_len, hkv, d, seed=2), _rnd(b, k_len, hkv, d, seed=3)
    scale = 1.0 / math.sqrt(d)
    out, lse = ops.decode_attention(q.cuda(), k.cuda(), v.cuda(), scale, n_split=n_split, return_lse=True)
    _check(out, lse, q, k, v, scale)


def test_decode_peaked_scores_and_split_merge():
    """A few keys dominate, placed in different splits and tiles: exercises the running-maximum rescale inside a wave,
    the merge of the four waves and the merge of the splits."""
    from long_vita_amd import ops
    b, hq, hkv, d, k_len = 1, 10, 2, 128, 6000
    q, k, v = _rnd(b, hq, d, seed=4), _rnd(b, k_len, hkv, d, seed=5, scale=0.3), _rnd(b, k_len, hkv, d, seed=6)
    for pos, gain in ((17, 2.0), (2900, 3.0), (5999, 4.0)):
        k[0, pos, 0] = (q[0, 0].float() * gain).to(torch.bfloat16)
        k[0, pos, 1] = (q[0, 7].float() * gain).to(torch.bfloat16)
    scale = 1.0 / math.sqrt(d)
    for n_split in (1, 4, 0):
        out, lse = ops.decode_attention(q.cuda(), k.cuda(), v.cuda(), scale, n_split=n_split, return_lse=True)
        _check(out, lse, q, k, v, scale)


def test_decode_on_views_of_a_sequence_major_cache():
    """The reference keeps the cache as [s, b, heads, d] and hands 'b s h' views to the op."""
    from long_vita_amd import ops
    s_max, b, hq, hkv, d, k_len = 700, 2, 10, 2, 128, 515
    cache_k, cache_v = _rnd(s_max, b, hkv, d, seed=7).cuda(), _rnd(s_max, b, hkv, d, seed=8).cuda()
    q = _rnd(b, hq, d, seed=9)
    kview, vview = cache_k[:k_len].transpose(0, 1), cache_v[:k_len].transpose(0, 1)
    assert not kview.is_contiguous()
    out = ops.decode_attention(q.cuda(), kview, vview, 0.1)
    _check(out, None, q, kview.cpu(), vview.cpu(), 0.1)


def test_decode_rejects_bad_arguments():
    from long_vita_amd import ops
    q, k = _rnd(1, 66, 128, seed=1).cuda(), _rnd(1, 64, 2, 128, seed=2).cuda()
    with pytest.raises(ops.LvAttnError):
        ops.decode_attention(q, k, k, 0.1)                 # 33 query heads per KV head
    with pytest.raises(ops.LvAttnError):
        ops.decode_attention(q.cpu(), k.cpu(), k.cpu(), 0.1)
    with pytest.raises(ops.LvAttnError):
        ops.decode_attention(q[:, :4], k[:, :0], k[:, :0], 0.1)   # empty cache


def test_kv_cache_branch_of_flash_attention_forward():
    """Prompt pass + token-by-token incremental passes through flash_attention_forward (--use-kv-cache) give the rows
    of one causal attention over the whole sequence."""
    from long_vita_amd import core_attention, state
    S, P, b, hq, hkv, d = 300, 291, 2, 10, 2, 128
    q, k, v = _rnd(S, b, hq, d, seed=11), _rnd(S, b, hkv, d, seed=12), _rnd(S, b, hkv, d, seed=13)
    scale = 1.0 / math.sqrt(d)
    state.set_args(state.make_args(use_kv_cache=True))
    try:
        self = types.SimpleNamespace(hidden_size_per_attention_head=d, scale_mask_softmax=types.SimpleNamespace(scale=None),
                                     softmax_scale=scale, norm_factor=math.sqrt(d), alibi=None,
                                     attention_dropout=types.SimpleNamespace(p=0.0))
        qg, kg, vg = q.cuda(), k.cuda(), v.cuda()
        rows = [core_attention.flash_attention_forward(self, qg[:P], kg[:P], vg[:P], None, None, None)]
        assert rows[0].shape == (P, b, hq * d)
        # a second multi-token chunk against the longer cache (causal, aligned to the end of the keys)
        rows.append(core_attention.flash_attention_forward(self, qg[P:P + 4], kg[:P + 4], vg[:P + 4], None, None, None))
        for t in range(P + 4, S):
            rows.append(core_attention.flash_attention_forward(self, qg[t:t + 1], kg[:t + 1], vg[:t + 1], None, None, None))
        got = torch.cat(rows, dim=0).float().cpu().numpy().reshape(S, b, hq, d)
    finally:
        state.set_args(None)
    for i in range(b):
        ref, _ = A.varlen_attention(q[:, i], k[:, i], v[:, i], [0, S], scale, True)
        assert np.abs(got[:, i] - ref).max() <= REL_TOL * np.abs(ref).max()


def test_sharded_cache_partials_merge_to_the_whole():
    """Two zig-zag shards of a cache through the incremental kernel, merged in log-sum-exp form
    (what decode_context_parallel does after its all_gather), against the oracle over the whole cache."""
    from long_vita_amd import ops
    from long_vita_amd.ring_context_parallel import merge_partial_attention
    b, hq, hkv, d, S, cp = 2, 10, 2, 128, 4096, 2
    q, k, v = _rnd(b, hq, d, seed=21), _rnd(b, S, hkv, d, seed=22), _rnd(b, S, hkv, d, seed=23)
    scale = 1.0 / math.sqrt(d)
    chunks = torch.arange(S).view(2 * cp, -1)
    outs, lses = [], []
    for r in range(cp):
        idx = torch.cat([chunks[r], chunks[2 * cp - 1 - r]])
        o, l = ops.decode_attention(q.cuda(), k[:, idx].cuda(), v[:, idx].cuda(), scale, return_lse=True)
        outs.append(o.float())
        lses.append(l)
    merged = merge_partial_attention(torch.stack(outs), torch.stack(lses))
    _check(merged, None, q, k, v, scale)


@pytest.mark.parametrize("sq,sk,causal,pieces", [(300, 70000, True, None), (300, 70000, False, 5), (70, 40000, True, 3),
                                                 (513, 33000, True, 7), (256, 1000, True, 4), (300, 300, True, 3)])
def test_prompt_block_on_long_cache_split_along_the_keys(sq, sk, causal, pieces):
    """A short block of new tokens against a long cache (dot_product_attention.py:440-451) runs as (query block, key
    piece) work items of one launch, merged in log-sum-exp form: against the float64 oracle on the whole rectangle, and
    against the single-sweep launch of the same rectangle (which differs only by the bf16 rounding of the pieces)."""
    from long_vita_amd import ops
    Hq, Hkv, D = 10, 2, 128
    g = torch.Generator().manual_seed(sq + sk)
    q = (torch.randn(sq, Hq, D, generator=g) * 0.8).to(torch.bfloat16)
    k = (torch.randn(sk, Hkv, D, generator=g) * 0.8).to(torch.bfloat16)
    v = torch.randn(sk, Hkv, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    shift = sk - sq
    if pieces is None:
        assert ops._key_pieces_for(sq, sk, Hq, torch.device("cuda", 0)) == 4        # 20 work items on 256 CUs, 70,000 keys: 4 pieces of >= 16,384
    out, lse = ops.attn_fwd_rect_split_keys(q.cuda(), k.cuda(), v.cuda(), scale, causal, shift, n_pieces=pieces)
    one, one_lse = ops.attn_fwd_rect_split_keys(q.cuda(), k.cuda(), v.cuda(), scale, causal, shift, n_pieces=1)
    torch.cuda.synchronize()
    assert out.shape == (sq, Hq, D) and lse.shape == (Hq, sq)
    assert (out.float() - one.float()).abs().max().item() <= 2e-2 * one.float().abs().max().item()
    assert (lse - one_lse).abs().max().item() <= 1e-3
    # float64 on the rectangle: the rows are the LAST sq rows of a causal segment of sk tokens (or see every key)
    rep = Hq // Hkv
    qd, kd, vd = q.double(), k.double().repeat_interleave(rep, 1), v.double().repeat_interleave(rep, 1)
    sc = torch.einsum("ihd,jhd->hij", qd, kd) * scale
    if causal:
        vis = torch.arange(sk)[None, :] <= (torch.arange(sq)[:, None] + shift)
        sc = sc.masked_fill(~vis[None], float("-inf"))
    ref_lse = torch.logsumexp(sc, dim=-1)
    ref = torch.einsum("hij,jhd->ihd", torch.softmax(sc, dim=-1), vd)
    assert (out.double().cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert (lse.double().cpu() - ref_lse).abs().max().item() <= 1e-3


def test_prompt_block_pieces_long_enough_for_the_one_wave_kernel():
    """Key pieces of 70,000 keys: the library runs them on the one-wave-per-SIMD forward kernel (a work item's sweep is
    what selects it), here with a query block of 300 rows only, the second work item of a head holding 44 of them."""
    from long_vita_amd import ops
    sq, sk, Hq, Hkv, D = 300, 140000, 2, 1, 128
    g = torch.Generator().manual_seed(5)
    q = (torch.randn(sq, Hq, D, generator=g) * 0.8).to(torch.bfloat16)
    k = (torch.randn(sk, Hkv, D, generator=g) * 0.8).to(torch.bfloat16)
    v = torch.randn(sk, Hkv, D, generator=g).to(torch.bfloat16)
    scale, shift = A.default_scale(D), sk - sq
    out, lse = ops.attn_fwd_rect_split_keys(q.cuda(), k.cuda(), v.cuda(), scale, True, shift, n_pieces=2)
    torch.cuda.synchronize()
    qd, kd, vd = q.double(), k.double().repeat_interleave(Hq // Hkv, 1), v.double().repeat_interleave(Hq // Hkv, 1)
    sc = torch.einsum("ihd,jhd->hij", qd, kd) * scale
    vis = torch.arange(sk)[None, :] <= (torch.arange(sq)[:, None] + shift)
    sc = sc.masked_fill(~vis[None], float("-inf"))
    ref = torch.einsum("hij,jhd->ihd", torch.softmax(sc, dim=-1), vd)
    assert (out.double().cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert (lse.double().cpu() - torch.logsumexp(sc, dim=-1)).abs().max().item() <= 1e-3


def test_prompt_flash_attention_op_takes_the_split_path():
    """npu_prompt_flash_attention on a 200-token block against a 50,000-token cache (batch 2): equal to the one-sweep
    rectangle to bf16 rounding, through the reference-shaped op."""
    from long_vita_amd import ops
    from long_vita_amd.fusion_attention import npu_prompt_flash_attention
    b, sq, sk, Hq, Hkv, D = 2, 200, 50000, 8, 2, 128
    g = torch.Generator(device="cuda").manual_seed(8)
    rnd = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.7).to(torch.bfloat16)
    q, k, v = rnd(b, sq, Hq * D), rnd(b, sk, Hkv * D), rnd(b, sk, Hkv * D)
    scale = D ** -0.5
    out = npu_prompt_flash_attention(q, k, v, num_heads=Hq, input_layout="BSH", atten_mask=True, scale_value=scale,
                                     sparse_mode=2, num_key_value_heads=Hkv)
    assert out.shape == (b, sq, Hq * D)
    for i in range(b):
        desc = ops.rect_blocks_on_device(0, sq, 0, sk, True, sk - sq, False, "cuda")
        one, _ = ops.attn_fwd_launch(q[i].unflatten(1, (Hq, D)), k[i].unflatten(1, (Hkv, D)), v[i].unflatten(1, (Hkv, D)), desc, scale, True)
        assert (out[i].float() - one.reshape(sq, -1).float()).abs().max().item() <= 2e-2 * one.float().abs().max().item()

