"""Split sweeps on the GPU (few work items for the chip: one tensor-parallel rank's head count, short batches): the
forward, dQ and dK/dV launches over (block, piece) work items + the combine kernels against the float64 oracle, against
the unsplit launches, and bit for bit from run to run."""
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


@pytest.fixture
def split_override():
    """Forces piece lengths; the backward's fp32 workspaces start out as NaN, so that a partial result the combine reads
    without a work item having written it (they are stored, not added to a zeroed buffer) cannot pass."""
    from long_vita_amd import ops
    ops._poison_workspaces = True
    yield ops.set_split_override
    ops.set_split_override(None, None)
    ops._poison_workspaces = False


CASES = [
    ("one_segment", [0, 1500], 5, 1, 128, True, 256, 256),
    ("packed", [0, 100, 356, 357, 1290, 1290, 3000], 5, 1, 128, True, 512, 128),
    ("packed_gqa2", [0, 700, 2048], 10, 2, 128, True, 256, 512),
    ("dense_d64", [0, 1025], 4, 4, 64, False, 256, 256),
    ("causal_d64", [0, 130, 900], 4, 2, 64, True, 320, 96),
    ("only_dkdv_cut", [0, 1200], 5, 1, 128, True, 0, 256),
    ("only_fwd_cut", [0, 1200], 5, 1, 128, True, 256, 0),
]


@pytest.mark.parametrize("name,cu,Hq,Hkv,D,causal,pf,pb", CASES, ids=[c[0] for c in CASES])
def test_split_forward_backward_match_oracle_and_unsplit(name, cu, Hq, Hkv, D, causal, pf, pb, split_override):
    from long_vita_amd import ops
    T = cu[-1]
    g = torch.Generator().manual_seed(len(name) + T)
    q, k, v, do = (torch.randn(T, H, D, generator=g).to(torch.bfloat16) for H in (Hq, Hkv, Hkv, Hq))
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, causal)
    res = {}
    for tag, over in (("split", (pf, pb)), ("split_again", (pf, pb)), ("whole", (0, 0))):
        split_override(*over)
        qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
        out, lse = ops.varlen_attention(qc, kc, vc, cu, scale, causal)
        out.backward(do.cuda())
        torch.cuda.synchronize()
        res[tag] = (out.detach(), lse, qc.grad, kc.grad, vc.grad)
    out, lse, dq, dk, dv = res["split"]
    assert _rel(out, ref_o) <= REL_TOL and np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL
    assert _rel(dq, rdq) <= REL_TOL and _rel(dk, rdk) <= REL_TOL and _rel(dv, rdv) <= REL_TOL
    assert all(torch.equal(a, b) for a, b in zip(res["split"], res["split_again"])), "not reproducible"
    for a, b, what in zip(res["split"], res["whole"], ("out", "lse", "dq", "dk", "dv")):
        tol = 1e-4 if what == "lse" else REL_TOL * float(b.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= tol, what


def test_planner_cuts_one_tensor_parallel_ranks_batch_and_result_is_right():
    """8,192 tokens at 5 / 1 heads (one tensor-parallel rank of the reference's TP = 8 layout): the planner cuts both
    sweeps by itself; forward + backward against the float64 oracle, and the same call at 40 / 8 heads is left whole."""
    from long_vita_amd import ops
    T, D = 8192, 128
    assert all(x > 0 for x in ops.plan_split([0, T], True, 5, 1, "cuda"))
    assert ops.plan_split([0, 131072], True, 40, 8, "cuda") == (0, 0)
    g = torch.Generator().manual_seed(8)
    q, k, v, do = (torch.randn(T, H, D, generator=g).to(torch.bfloat16) * 0.5 for H in (5, 1, 1, 5))
    scale = A.default_scale(D)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, lse = ops.varlen_attention(qc, kc, vc, [0, T], scale, True)
    out.backward(do.cuda())
    torch.cuda.synchronize()
    ref_o, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, True)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, [0, T], scale, True)
    assert _rel(out, ref_o) <= REL_TOL and np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL
    assert _rel(qc.grad, rdq) <= REL_TOL and _rel(kc.grad, rdk) <= REL_TOL and _rel(vc.grad, rdv) <= REL_TOL
