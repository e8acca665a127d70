"""Banded causal attention on the GPU (the fused op's band modes: a query sees itself and the pre_tockens keys before
it; --sliding-window sets pre_tockens; lcvlm_modellink/core/transformer/dot_product_attention.py:403-406, 454-467):
forward + backward against the fixture made by the reference's eager wrapper under band masks (autograd through it),
against the float64 oracle on longer packed cases, and through the op-shaped entry points."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention_ref as A

REL_TOL = 2e-2
LSE_TOL = 1e-3


def _rel(got, ref):
    g = got.detach().float().cpu().numpy().astype(np.float64)
    return float(np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-30))


def _bits(bits):
    return torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16)


@pytest.mark.parametrize("name", ["band_causal", "band_packed", "band_d64", "band_self_only", "band_wider_than_rows"])
def test_band_forward_backward_match_autograd_of_reference_wrapper(name, golden_dir):
    from long_vita_amd import ops
    g = np.load(os.path.join(golden_dir, "eager_attention_band.npz"))
    sq, np_, ng, hn, band = (int(x) for x in g[f"{name}.meta"])
    cu = [int(x) for x in g[f"{name}.cu"]]
    q, k, v = (_bits(g[f"{name}.{x}"])[:, 0].cuda().requires_grad_(True) for x in "qkv")
    do = _bits(g[f"{name}.dout"])[:, 0].reshape(sq, np_, hn).cuda()
    out, _ = ops.varlen_attention(q, k, v, cu, None, True, band=band)
    out.backward(do)
    torch.cuda.synchronize()
    ref_out = _bits(g[f"{name}.out"]).double().numpy().reshape(sq, np_, hn)
    assert _rel(out, ref_out) <= REL_TOL, name
    for got, key in ((q.grad, "dq"), (k.grad, "dk"), (v.grad, "dv")):
        ref = g[f"{name}.{key}"].astype(np.float64)[:, 0]
        if np.abs(ref).max() < 1e-6:      # a band of zero: every row sees itself only, P = 1, dS = 0: dq and dk vanish
            assert got.float().abs().max().item() <= 1e-3, (name, key)
        else:
            assert _rel(got, ref) <= REL_TOL, (name, key)


@pytest.mark.parametrize("cu,Hq,Hkv,D,band", [([0, 3000], 5, 1, 128, 700), ([0, 100, 1400, 1401, 2600], 10, 2, 128, 300),
                                             ([0, 900, 2048], 4, 2, 64, 129), ([0, 2000], 2, 1, 128, 63), ([0, 1300], 2, 2, 128, 256)])
def test_band_forward_backward_match_oracle(cu, Hq, Hkv, D, band):
    """Bands that cross several 256-row query blocks, 64-key tiles and 128-key blocks, packed and not, d = 128 and 64."""
    from long_vita_amd import ops
    T = cu[-1]
    g = torch.Generator().manual_seed(T + band)
    q, k, v, do = (torch.randn(T, H, D, generator=g).to(torch.bfloat16) for H in (Hq, Hkv, Hkv, Hq))
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, True, band=band)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, True, band=band)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, lse = ops.varlen_attention(qc, kc, vc, cu, scale, True, band=band)
    out.backward(do.cuda())
    torch.cuda.synchronize()
    assert _rel(out, ref_o) <= REL_TOL and np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL
    assert _rel(qc.grad, rdq) <= REL_TOL and _rel(kc.grad, rdk) <= REL_TOL and _rel(vc.grad, rdv) <= REL_TOL
    # and it differs from plain causal attention (the band is felt)
    plain, _ = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), cu, scale, True)
    assert (plain.float() - out.detach().float()).abs().max().item() > 1e-2


def test_band_through_the_op_shaped_entry_points():
    """npu_fusion_attention in a band mode (sparse_mode 4 / 0 with pre_tockens shorter than a segment) applies the band,
    mode 2 ignores pre_tockens; flash_attention_forward with --sliding-window; the kv-cache prompt op with pre_tokens."""
    from long_vita_amd import state
    from long_vita_amd.core_attention import flash_attention_forward
    from long_vita_amd.fusion_attention import npu_fusion_attention, npu_prompt_flash_attention
    T, Hq, Hkv, D, band = 1000, 4, 2, 128, 200
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(T, H, D, generator=g).to(torch.bfloat16) for H in (Hq, Hkv, Hkv))
    scale = A.default_scale(D)
    banded, _ = A.varlen_attention(q, k, v, [0, 400, T], scale, True, band=band)
    plain, _ = A.varlen_attention(q, k, v, [0, 400, T], scale, True)
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    for mode, want in ((4, banded), (0, banded), (2, plain)):
        out = npu_fusion_attention(qc, kc, vc, Hq, "TND", atten_mask=True, scale=scale, pre_tockens=band, next_tockens=0,
                                   sparse_mode=mode, actual_seq_qlen=[400, T], actual_seq_kvlen=[400, T])[0]
        assert _rel(out, want) <= REL_TOL, mode
    # the patched entry point, generic SBH branch, sliding window (:403-406 -> pre_tockens)
    banded1, _ = A.varlen_attention(q, k, v, [0, T], scale, True, band=band)
    me = types.SimpleNamespace(hidden_size_per_attention_head=D, scale_mask_softmax=types.SimpleNamespace(scale=None),
                               softmax_scale=None, attention_dropout=types.SimpleNamespace(p=0.0), alibi=None, pse=None, pse_type=None)
    state.set_args(state.make_args(sliding_window=band, sparse_mode=4))
    out = flash_attention_forward(me, qc[:, None], kc[:, None], vc[:, None], None, None, None)
    assert _rel(out.reshape(T, Hq, D), banded1) <= REL_TOL
    state.set_args(state.make_args(sliding_window=None, sparse_mode=0, pre_tockens=65536))
    out = flash_attention_forward(me, qc[:, None], kc[:, None], vc[:, None], None, None, None)
    assert _rel(out.reshape(T, Hq, D), A.varlen_attention(q, k, v, [0, T], scale, True)[0]) <= REL_TOL
    # prompt pass over a cache: the last 300 tokens are new, each sees itself and the `band` keys before it
    sq = 300
    out = npu_prompt_flash_attention(qc[T - sq:].reshape(1, sq, Hq * D), kc.reshape(1, T, Hkv * D), vc.reshape(1, T, Hkv * D),
                                     num_heads=Hq, input_layout="BSH", atten_mask=True, scale_value=scale, pre_tokens=band,
                                     next_tokens=0, sparse_mode=0, num_key_value_heads=Hkv)
    assert _rel(out.reshape(sq, Hq, D), banded1[T - sq:]) <= REL_TOL
