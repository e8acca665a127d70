"""Seeded random shapes through the forward and backward kernels against the float64 oracle: ragged packings
(empty, one-token and block-straddling segments), GQA ratios 1..8, both head dims, causal and full."""
import numpy as np
import pytest
import torch

from oracle import attention_ref as A

pytestmark = pytest.mark.gpu
REL_TOL, LSE_TOL = 2e-2, 1e-3


def _case(seed):
    rs = np.random.RandomState(seed)
    D = int(rs.choice([64, 128]))
    Hkv = int(rs.choice([1, 2, 3]))
    group = int(rs.choice([1, 2, 5, 8]))
    causal = bool(rs.rand() < 0.7)
    nseg = int(rs.randint(1, 7))
    lens = []
    for _ in range(nseg):
        kind = rs.rand()
        if kind < 0.15:
            lens.append(0)
        elif kind < 0.3:
            lens.append(int(rs.randint(1, 4)))
        elif kind < 0.6:
            lens.append(int(rs.choice([31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257])))
        else:
            lens.append(int(rs.randint(1, 900)))
    if sum(lens) == 0:
        lens[-1] = 77
    cu = [0] + list(np.cumsum(lens))
    return D, Hkv, Hkv * group, causal, [int(x) for x in cu]


def _rel(got, ref):
    g = got.detach().float().cpu().numpy().astype(np.float64)
    return float(np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-3))   # exact-zero gradients (one-key rows): absolute floor


@pytest.mark.parametrize("seed", range(40))
def test_random_packing_forward_backward(seed):
    from long_vita_amd import ops
    D, Hkv, Hq, causal, cu = _case(seed)
    T = cu[-1]
    g = torch.Generator().manual_seed(1000 + seed)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, causal)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, lse = ops.varlen_attention(qc, kc, vc, cu, scale, causal)
    out.backward(do.cuda())
    torch.cuda.synchronize()
    tag = f"D={D} Hq={Hq} Hkv={Hkv} causal={causal} cu={cu}"
    assert _rel(out, ref_o) <= REL_TOL, tag
    assert np.abs(lse.detach().cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL, tag
    assert _rel(qc.grad, rdq) <= REL_TOL, tag
    assert _rel(kc.grad, rdk) <= REL_TOL, tag
    assert _rel(vc.grad, rdv) <= REL_TOL, tag


@pytest.mark.parametrize("seed", [s for s in range(40) if _case(s)[0] == 128])
def test_random_packing_forward_one_wave_kernel(seed):
    """The same ragged cases at head dim 128 with the one-wave-per-SIMD forward kernel forced: launches this small take the
    ping-pong kernel when the choice is the library's, so the kernel that serves long contexts is exercised on empty,
    one-token and block-straddling segments here."""
    from long_vita_amd import ops, _lib
    D, Hkv, Hq, causal, cu = _case(seed)
    T = cu[-1]
    g = torch.Generator().manual_seed(1000 + seed)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal)
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"a4"), "select")
    try:
        with torch.no_grad():
            out, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), cu, scale, causal)
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")
    tag = f"Hq={Hq} Hkv={Hkv} causal={causal} cu={cu}"
    assert _rel(out, ref_o) <= REL_TOL, tag
    assert np.abs(lse.detach().cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL, tag


@pytest.mark.parametrize("seed", range(40))
def test_random_band_or_split_forward_backward(seed):
    """The same random packings through the other two table forms: a band of random width (causal cases), or the sweeps
    cut at random forced piece lengths with the fp32 workspaces pre-filled with NaN."""
    from long_vita_amd import ops
    D, Hkv, Hq, causal, cu = _case(seed)
    rs = np.random.RandomState(7000 + seed)
    banded = causal and rs.rand() < 0.5
    T = cu[-1]
    g = torch.Generator().manual_seed(9000 + seed)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    band = int(rs.choice([0, 1, 5, 31, 32, 33, 63, 64, 100, 255, 256, 257, 500])) if banded else None
    pieces = (int(rs.choice([0, 256, 320, 512, 1024])), int(rs.choice([0, 32, 96, 128, 256, 512])))
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal, band)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, causal, band)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    ops._poison_workspaces = True
    if not banded:
        ops.set_split_override(*pieces)
    try:
        out, lse = ops.varlen_attention(qc, kc, vc, cu, scale, causal, band)
        out.backward(do.cuda())
        torch.cuda.synchronize()
    finally:
        ops.set_split_override(None, None)
        ops._poison_workspaces = False
    tag = f"D={D} Hq={Hq} Hkv={Hkv} causal={causal} band={band} pieces={pieces} cu={cu}"
    assert _rel(out, ref_o) <= REL_TOL, tag
    fin = np.isfinite(ref_lse)
    if fin.any():
        assert np.abs(lse.detach().cpu().numpy().astype(np.float64)[fin] - ref_lse[fin]).max() <= LSE_TOL, tag
    for got, ref, name in ((qc.grad, rdq, "dq"), (kc.grad, rdk, "dk"), (vc.grad, rdv, "dv")):
        assert torch.isfinite(got.float()).all(), (name, tag)
        assert _rel(got, ref) <= REL_TOL, (name, tag)
