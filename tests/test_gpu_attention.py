"""GPU parity tests of the attention kernels (through the C ABI) against the CPU
oracle.  Tolerances are the ones the project states for bf16.  Outputs and gradients, tensor level:
max|got - ref| <= 2e-2 * max|ref| (``rel_err``); per element: |got - ref| <= 2e-2 * |ref| + 5e-3 * max|ref|
(``elementwise_excess``).  LSE within 1e-3 absolute; the hard-numerics tests use the tighter, relative ``lse_bound``."""
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention_ref as A

REL_TOL = 2e-2
LSE_TOL = 1e-3


def lse_bound(ref_lse: np.ndarray) -> np.ndarray:
    """The log-sum-exp bound of the hard-numerics tests: 6 * 2^-23 * max(1, |lse|), i.e. six units in the last place of
    an fp32 number of that magnitude.  Measured on those tests' inputs under every forward kernel (tools_diag/lse_err.py):
    at most 2.8 such units (1.5e-4 absolute at |lse| = 839, 3.3e-5 at 141, 8.6e-7 at 3.5).  The bound stays below the
    1e-3 absolute of the ordinary tests up to |lse| = 1398; fp32 itself cannot hold 1e-3 once |lse| passes 8192 (one
    unit in the last place is 9.8e-4 there), which is why the rule is relative."""
    return 6.0 * 2.0 ** -23 * np.maximum(1.0, np.abs(ref_lse))


def rel_err(got: torch.Tensor, ref: np.ndarray) -> float:
    g = got.detach().float().cpu().numpy().astype(np.float64)
    return float(np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-30))


def elementwise_excess(got: torch.Tensor, ref: np.ndarray, rtol: float = REL_TOL, floor: float = 0.25) -> float:
    """Per-element form of the tolerance: max over elements of |got - ref| / (rtol * |ref| + rtol * floor * max|ref|).
    <= 1 means every element is within ``rtol`` of its own magnitude plus an absolute floor of ``floor * rtol`` of the
    tensor's largest magnitude (elements near zero are sums of cancelling terms and have no relative precision)."""
    g = got.detach().float().cpu().numpy().astype(np.float64)
    bound = rtol * np.abs(ref) + rtol * floor * np.abs(ref).max()
    return float((np.abs(g - ref) / bound).max())


def make_qkv(T, Hq, Hkv, D, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(T, Hq, D, generator=g) * scale).to(torch.bfloat16)
    k = (torch.randn(T, Hkv, D, generator=g) * scale).to(torch.bfloat16)
    v = (torch.randn(T, Hkv, D, generator=g) * scale).to(torch.bfloat16)
    return q, k, v


CASES = [
    # name, cu_seqlens, Hq, Hkv, D, causal
    ("one_tile", [0, 64], 2, 1, 128, True),
    ("ragged", [0, 100, 356, 357, 700, 1290], 10, 2, 128, True),
    ("multi_block", [0, 258, 1290, 3000], 5, 1, 128, True),
    ("dense_d64", [0, 1025], 4, 4, 64, False),
    ("dense_d128", [0, 300, 900], 4, 2, 128, False),
    ("causal_d64", [0, 130, 700], 4, 2, 64, True),
    ("empty_segments", [0, 0, 100, 100, 100, 228, 228], 4, 2, 128, True),
]


@pytest.mark.parametrize("name,cu,Hq,Hkv,D,causal", CASES, ids=[c[0] for c in CASES])
def test_forward_matches_oracle(name, cu, Hq, Hkv, D, causal):
    from long_vita_amd import ops
    T = cu[-1]
    q, k, v = make_qkv(T, Hq, Hkv, D, seed=zlib.crc32(name.encode()) % 1000)
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal)
    out, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), cu, scale, causal)
    torch.cuda.synchronize()
    assert rel_err(out, ref_o) <= REL_TOL
    assert np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL


def test_forward_exact_layout_integers():
    """Small-integer data: every product and sum is exact in bf16/fp32, so any
    fragment-layout mistake shows as a large error rather than rounding noise."""
    from long_vita_amd import ops
    T, Hq, Hkv, D = 192, 2, 1, 128
    g = torch.Generator().manual_seed(5)
    q = torch.zeros(T, Hq, D)
    k = torch.zeros(T, Hkv, D)
    # one-hot rows: q.k is 0 except where the hot columns agree -> softmax over {0, c}
    qi = torch.randint(0, D, (T, Hq), generator=g)
    ki = torch.randint(0, D, (T, Hkv), generator=g)
    q.scatter_(2, qi[..., None], 8.0)
    k.scatter_(2, ki[..., None], 1.0)
    v = torch.randint(-4, 5, (T, Hkv, D), generator=g).float()
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    ref_o, ref_lse = A.varlen_attention(q, k, v, [0, T], 1.0, True)
    out, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), [0, T], 1.0, True)
    assert rel_err(out, ref_o) <= 1e-2
    assert np.abs(lse.cpu().numpy() - ref_lse).max() <= LSE_TOL


def test_forward_strided_views_no_copy():
    """q/k/v as views into one fused [T, Hkv, (G+2)*D] projection, as Megatron hands them over."""
    from long_vita_amd import ops
    T, Hkv, G, D = 700, 2, 5, 128
    g = torch.Generator().manual_seed(11)
    fused = torch.randn(T, Hkv, (G + 2) * D, generator=g).to(torch.bfloat16).cuda()
    q = fused[:, :, :G * D].reshape(T, Hkv, G, D)          # not a view-compatible reshape -> copy (as Megatron does)
    q = q.reshape(T, Hkv * G, D)
    k = fused[:, :, G * D:(G + 1) * D]
    v = fused[:, :, (G + 1) * D:]
    assert not k.is_contiguous()
    cu = [0, 300, 700]
    ref_o, ref_lse = A.varlen_attention(q.cpu(), k.cpu(), v.cpu(), cu, A.default_scale(D), True)
    out, lse = ops.varlen_attention(q, k, v, cu, None, True)
    assert rel_err(out, ref_o) <= REL_TOL
    assert np.abs(lse.cpu().numpy() - ref_lse).max() <= LSE_TOL


def test_forward_matches_reference_eager_golden(golden_dir):
    """Against outputs of the reference's own eager attention wrapper."""
    from long_vita_amd import ops
    g = np.load(os.path.join(golden_dir, "eager_attention.npz"))
    for name, sq, b, np_, ng, hn in [("mha_d64", 65, 1, 4, 4, 64), ("qwen_ratio", 96, 1, 10, 2, 128)]:
        q, k, v = (torch.from_numpy(g[f"{name}.{x}"])[:, 0].to(torch.bfloat16).cuda() for x in "qkv")
        out, _ = ops.varlen_attention(q, k, v, [0, sq], None, True)
        assert rel_err(out.reshape(sq, 1, -1), g[f"{name}.out"].astype(np.float64)) <= REL_TOL
        out, _ = ops.varlen_attention(q, k, v, [0, sq], None, False)
        assert rel_err(out.reshape(sq, 1, -1), g[f"{name}.out_nomask"].astype(np.float64)) <= REL_TOL


def _bits_to_bf16(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16)


@pytest.mark.parametrize("name", ["qwen5to1_causal", "qwen5to1_packed", "gqa_packed_d64"])
def test_forward_and_backward_match_autograd_of_reference_wrapper(name, golden_dir):
    """HIP forward + backward against outputs and dq / dk / dv that torch autograd produced THROUGH THE REFERENCE'S
    OWN eager wrapper (dot_product_attention.py:148-288; fixture eager_attention_grad.npz): plain causal and packed
    (block-diagonal-causal) masks incl. a one-token segment, 5:1 heads at d = 128 over more than one 256-row block
    with a segment straddling the block boundary.  Checked with the tensor-level tolerance and per element."""
    from long_vita_amd import ops
    g = np.load(os.path.join(golden_dir, "eager_attention_grad.npz"))
    sq, np_, ng, hn = (int(x) for x in g[f"{name}.meta"])
    cu = [int(x) for x in g[f"{name}.cu"]]
    q, k, v = (_bits_to_bf16(g[f"{name}.{x}"])[:, 0].cuda().requires_grad_(True) for x in "qkv")
    do = _bits_to_bf16(g[f"{name}.dout"])[:, 0].reshape(sq, np_, hn).cuda()
    out, _ = ops.varlen_attention(q, k, v, cu, None, True)
    out.backward(do)
    torch.cuda.synchronize()
    for got, key in ((out.reshape(sq, 1, -1), "out"), (q.grad[:, None], "dq"), (k.grad[:, None], "dk"), (v.grad[:, None], "dv")):
        ref = g[f"{name}.{key}"].astype(np.float64)
        assert tuple(got.shape) == ref.shape, (name, key)
        assert rel_err(got, ref) <= REL_TOL, (name, key, rel_err(got, ref))
        assert elementwise_excess(got, ref) <= 1.0, (name, key, elementwise_excess(got, ref))


BWD_CASES = [
    ("bwd_one", [0, 128], 2, 1, 128, True),
    ("bwd_ragged", [0, 100, 356, 357, 700, 1000], 10, 2, 128, True),
    ("bwd_dense_d64", [0, 1025], 2, 2, 64, False),
    ("bwd_dense_d128", [0, 200, 520], 4, 2, 128, False),
    ("bwd_causal_d64", [0, 130, 450], 4, 2, 64, True),
    ("bwd_tiny_segments", [0, 1, 3, 40, 41, 300], 4, 2, 128, True),
    ("bwd_group5_long", [0, 2500], 5, 1, 128, True),
    ("bwd_empty_segments", [0, 0, 100, 100, 228, 228], 4, 2, 128, True),
    ("bwd_odd_unit_count", [0, 96], 1, 1, 128, True),      # 3 query slices x 1 head for the first key block
    ("bwd_odd_unit_count_d64", [0, 33, 200], 3, 1, 64, False),
]


@pytest.mark.parametrize("name,cu,Hq,Hkv,D,causal", BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_backward_matches_oracle(name, cu, Hq, Hkv, D, causal):
    from long_vita_amd import ops
    T = cu[-1]
    q, k, v = make_qkv(T, Hq, Hkv, D, seed=zlib.crc32(name.encode()) % 1000)
    g = torch.Generator().manual_seed(99)
    do = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, causal)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, _ = ops.varlen_attention(qc, kc, vc, cu, scale, causal)
    out.backward(do.cuda())
    torch.cuda.synchronize()
    assert rel_err(qc.grad, rdq) <= REL_TOL
    assert rel_err(kc.grad, rdk) <= REL_TOL
    assert rel_err(vc.grad, rdv) <= REL_TOL
    for got, ref in ((out, A.varlen_attention(q, k, v, cu, scale, causal)[0]), (qc.grad, rdq), (kc.grad, rdk), (vc.grad, rdv)):
        assert elementwise_excess(got, ref) <= 1.0, (name, elementwise_excess(got, ref))


@pytest.fixture(params=["default", "a4", "pp16", "lockstep"])
def forced_forward_kernel(request):
    """The hard-numerics tests below run under the library's own kernel choice (the ping-pong kernel at these sizes) and
    under each other forward kernel forced by name: the one-wave kernel serves the long launches, where a float64 check of
    every row is out of reach, so its optimistic and exact softmax passes meet the awkward inputs here."""
    from long_vita_amd import _lib
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(request.param.encode()), "select")
    yield request.param
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")


def test_forward_rising_and_falling_scores(forced_forward_kernel):
    """Scores that climb steadily along the keys (the running maximum has to be raised again and
    again: the optimistic softmax pass keeps falling back to the exact one) and scores that fall
    steadily (everything after the first tiles is far below the running maximum), d = 128 and 64."""
    from long_vita_amd import ops
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
                ref_o, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, causal)
                out, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), [0, T], scale, causal)
                assert rel_err(out, ref_o) <= REL_TOL, (D, sign, causal)
                err = np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse)
                assert (err <= lse_bound(ref_lse)).all(), (D, sign, causal, err.max())


def test_forward_rescale_branch_is_exercised(forced_forward_kernel):
    """The kernel defers the running-max update until a row's maximum grows by more than
    2^8; spike two keys so that the update fires in the middle of the key sweep (and a
    second time later), far from the first tile, and check every row against float64."""
    from long_vita_amd import ops
    T, Hq, Hkv, D = 2048, 4, 2, 128
    g = torch.Generator().manual_seed(77)
    u = torch.randn(D, generator=g)
    u = u / u.norm() * (D ** 0.5)
    q = torch.randn(T, Hq, D, generator=g) * 0.3 + 2.0 * u
    k = torch.randn(T, Hkv, D, generator=g) * 0.3
    v = torch.randn(T, Hkv, D, generator=g)
    k[500] = 3.0 * u          # score ~ 6*128/sqrt(128) = 68 -> far beyond the 2^8 threshold
    k[1500, 0] = 6.0 * u      # second jump, one KV head only
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, True)
    out, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), [0, T], scale, True)
    assert rel_err(out, ref_o) <= REL_TOL
    assert (np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse) <= lse_bound(ref_lse)).all()
    # non-causal: every row meets both spikes
    ref_o, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, False)
    out, lse = ops.varlen_attention(q.cuda(), k.cuda(), v.cuda(), [0, T], scale, False)
    assert rel_err(out, ref_o) <= REL_TOL
    assert (np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse) <= lse_bound(ref_lse)).all()


def test_forward_extreme_score_ranges(forced_forward_kernel):
    """Scores in the thousands (every tile's first look overflows the optimistic pass's bound, softmax rows are
    one-hot), all-zero queries (uniform attention over the visible keys: the output is the running mean of V), and one
    segment where both occur, d = 128 and 64, causal and full: outputs and log-sum-exp against float64."""
    from long_vita_amd import ops
    for D, T in ((128, 1500), (64, 700)):
        g = torch.Generator().manual_seed(9 + D)
        q = torch.randn(T, 4, D, generator=g) * 12.0
        k = torch.randn(T, 2, D, generator=g) * 12.0
        v = torch.randn(T, 2, D, generator=g)
        q[T // 3: T // 3 + 100] = 0.0                    # uniform rows
        q[5] = 0.0
        q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        scale = A.default_scale(D)
        for causal in (True, False):
            ref_o, ref_lse = A.varlen_attention(q, k, v, [0, T], scale, causal)
            out, lse = ops.attn_fwd_launch(q.cuda(), k.cuda(), v.cuda(), ops.blocks_on_device([0, T], causal, False, "cuda"), scale, causal)
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all(), (D, causal)
            assert rel_err(out, ref_o) <= REL_TOL, (D, causal, rel_err(out, ref_o))
            err = np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse)
            assert (err <= lse_bound(ref_lse)).all(), (D, causal, err.max())   # fp32 log-sum-exp of scores in the thousands


@pytest.mark.parametrize("dq_kernel,dkdv_kernel", [("default", "default"), ("lockstep", "single"), ("pp", "rs"), ("pp", "rs1"), ("pp", "g")])
def test_backward_hard_score_ranges(dq_kernel, dkdv_kernel):
    """The backward kernels recompute P from the saved log-sum-exp: scores that climb along the keys with a spike (rows
    dominated by one late key), moderately large scores (sharp but not one-hot rows) and all-zero queries (uniform rows),
    two packed segments, d = 128 and 64: dq / dk / dv against the float64 oracle under every backward kernel pair."""
    from long_vita_amd import ops, _lib
    _lib.check(_lib.lib().lvattn_select_bwd_kernels(dq_kernel.encode(), dkdv_kernel.encode()), "select")
    try:
        for D, cu in ((128, [0, 700, 1500]), (64, [0, 300, 700])):
            T = cu[-1]
            g = torch.Generator().manual_seed(21 + D)
            u = torch.randn(D, generator=g)
            u = u / u.norm() * (D ** 0.5)
            ramp = torch.linspace(0.0, 4.0, T)[:, None, None]
            q = torch.randn(T, 4, D, generator=g) * 2.0 + 1.0 * u
            k = torch.randn(T, 2, D, generator=g) * 2.0 + ramp * u
            v = torch.randn(T, 2, D, generator=g)
            do = torch.randn(T, 4, D, generator=g)
            k[T // 2] = 5.0 * u
            q[T // 3: T // 3 + 64] = 0.0
            q, k, v, do = (t.to(torch.bfloat16) for t in (q, k, v, do))
            scale = A.default_scale(D)
            rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, True)
            qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
            out, _ = ops.varlen_attention(qc, kc, vc, cu, scale, True)
            out.backward(do.cuda())
            torch.cuda.synchronize()
            for name, got, ref in (("dq", qc.grad, rdq), ("dk", kc.grad, rdk), ("dv", vc.grad, rdv)):
                assert torch.isfinite(got.float()).all(), (D, name)
                assert rel_err(got, ref) <= REL_TOL, (D, name, rel_err(got, ref))
    finally:
        _lib.check(_lib.lib().lvattn_select_bwd_kernels(b"default", b"default"), "select")


def test_full_size_1m_prefill_rows_against_oracle_and_decode_kernel():
    """The bench's full-size case (one causal segment of 1,048,576 tokens, 40 / 8 heads, d = 128): a sample of output
    rows against the float64 oracle (a row of attention is one softmax over its visible keys, cheap at any length), and
    the last row against the incremental-attention kernel, which shares no code path with the prefill kernel's loop."""
    from long_vita_amd import ops
    S, Hq, Hkv, D = 1048576, 40, 8, 128
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *shape: (torch.randn(*shape, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v = rnd(S, Hq, D), rnd(S, Hkv, D), rnd(S, Hkv, D)
    scale = A.default_scale(D)
    out, lse = ops.varlen_attention(q, k, v, [0, S], scale, True)
    torch.cuda.synchronize()
    rows = [0, 1, 255, 256, 65535, 524288, 1048575]
    rep = Hq // Hkv
    for r in rows:
        for kvh in (0, Hkv - 1):
            kk = k[: r + 1, kvh].double().cpu().numpy()
            vv = v[: r + 1, kvh].double().cpu().numpy()
            for h in (kvh * rep, kvh * rep + rep - 1):
                s = (kk @ q[r, h].double().cpu().numpy()) * scale
                m = s.max()
                pr = np.exp(s - m)
                ref = (pr @ vv) / pr.sum()
                got = out[r, h].double().cpu().numpy()
                assert np.abs(got - ref).max() <= REL_TOL * max(np.abs(ref).max(), 1e-3), (r, h)
                assert abs(float(lse[h, r]) - (m + np.log(pr.sum()))) <= LSE_TOL, (r, h)
    dec, dlse = ops.decode_attention(q[S - 1][None], k[None], v[None], scale, return_lse=True)
    assert (dec[0].float() - out[S - 1].float()).abs().max().item() <= REL_TOL * out[S - 1].float().abs().max().item()
    assert (dlse[0] - lse[:, S - 1]).abs().max().item() <= LSE_TOL


@pytest.mark.parametrize("kernel", ["a4", "pp", "pp16", "lockstep"])
def test_every_forward_kernel_variant(kernel):
    """The forward kernels that are not the default for a head dim stay selectable (LVATTN_FWD_KERNEL or
    lvattn_select_fwd_kernel) and give the same results to tolerance."""
    from long_vita_amd import ops, _lib
    cases = [([0, 100, 356, 357, 700, 1290], 10, 2, 128, True), ([0, 300, 900], 4, 2, 128, False), ([0, 130, 700], 4, 2, 64, True)]
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(kernel.encode()), "select")
    try:
        for cu, Hq, Hkv, D, causal in cases:
            q, k, v = make_qkv(cu[-1], Hq, Hkv, D, seed=len(kernel) + D)
            scale = A.default_scale(D)
            ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal)
            out, lse = ops.attn_fwd_launch(q.cuda(), k.cuda(), v.cuda(), ops.blocks_on_device(cu, causal, False, "cuda"), scale, causal)
            torch.cuda.synchronize()
            assert rel_err(out, ref_o) <= REL_TOL, (kernel, cu)
            assert np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL, (kernel, cu)
    finally:
        _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")
    with pytest.raises(_lib.LvAttnError):
        _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"no-such-kernel"), "select")


def test_default_forward_kernel_follows_the_key_sweep():
    """Head dim 128, the library's own choice: a launch of many short packed segments runs the ping-pong kernel and a
    long single segment the one-wave kernel, whatever the launch's total row count (the descriptors' mean key count
    per work item goes to lvattn_fwd_hinted); each result equals, bit for bit, the one of the kernel forced by name."""
    from long_vita_amd import ops, _lib
    T, D = 131072, 128
    q, k, v = (t.cuda() for t in make_qkv(T, 2, 1, D, seed=77))
    scale = A.default_scale(D)

    def run(cu, kernel):
        _lib.check(_lib.lib().lvattn_select_fwd_kernel(kernel.encode()), "select")
        try:
            out, lse = ops.attn_fwd_launch(q, k, v, ops.blocks_on_device(cu, True, False, "cuda"), scale, True)
            torch.cuda.synchronize()
        finally:
            _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")
        return out, lse
    packed, single = list(range(0, T + 1, 8192)), [0, T]
    assert ops.blocks_on_device(packed, True, False, "cuda").lv_mean_keys == 4224
    assert ops.blocks_on_device(single, True, False, "cuda").lv_mean_keys == 65664
    for cu, expected in ((packed, "pp"), (single, "a4")):
        out, lse = run(cu, "default")
        ref_out, ref_lse = run(cu, expected)
        assert torch.equal(out, ref_out) and torch.equal(lse, ref_lse), expected
    # and the packed result is right: rows of the last segment against the oracle
    out, lse = run(packed, "default")
    lo = T - 8192
    ro, rl = A.varlen_attention(q[lo:].cpu(), k[lo:].cpu(), v[lo:].cpu(), [0, 8192], scale, True)
    assert rel_err(out[lo:], ro) <= REL_TOL
    assert np.abs(lse[:, lo:].cpu().numpy().astype(np.float64) - rl).max() <= LSE_TOL


@pytest.mark.parametrize("rows,k_rows,Hq,Hkv,D,causal", [
    (1025, 1025, 32, 32, 64, False),      # the vision tower's shape: 8 x 128 rows + 1 (4-wave workgroups)
    (1024, 1024, 16, 16, 64, False),      # no rows past the last full item (8-wave workgroups)
    (37, 37, 4, 4, 64, False), (300, 300, 4, 2, 64, True),
    (700, 700, 10, 2, 128, True), (130, 450, 10, 2, 128, True),   # the second: causal rule aligned to the end of a longer key range
    (257, 513, 4, 2, 64, True), (513, 300, 5, 5, 128, False)])
def test_dense_entry_point(rows, k_rows, Hq, Hkv, D, causal, forced_forward_kernel):
    """lvattn_fwd_dense (one segment, no descriptors, workgroup size chosen from the row count) against the oracle and,
    bit for bit, against lvattn_fwd on the descriptors of the same rectangle."""
    from long_vita_amd import ops
    g = torch.Generator().manual_seed(rows + k_rows + D)
    q = (torch.randn(rows, Hq, D, generator=g) * 0.7).to(torch.bfloat16)
    k = (torch.randn(k_rows, Hkv, D, generator=g) * 0.7).to(torch.bfloat16)
    v = (torch.randn(k_rows, Hkv, D, generator=g) * 0.7).to(torch.bfloat16)
    scale = A.default_scale(D)
    out, lse = ops.attn_fwd_dense_launch(q.cuda(), k.cuda(), v.cuda(), scale, causal)
    desc = ops.rect_blocks_on_device(0, rows, 0, k_rows, causal, k_rows - rows, False, "cuda")
    out2, lse2 = ops.attn_fwd_launch(q.cuda(), k.cuda(), v.cuda(), desc, scale, causal)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    if rows == k_rows:
        ref_o, ref_lse = A.varlen_attention(q, k, v, [0, rows], scale, causal)
        assert rel_err(out, ref_o) <= REL_TOL
        assert np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL


def test_dense_entry_point_strided_views_and_errors():
    """The tower's layout: q, k, v as strided views of one [s, b * heads, 3 * d] GEMM output; bad calls fail loudly."""
    from long_vita_amd import ops, _lib
    s, heads, d = 1025, 24, 64
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(s, heads, 3 * d, generator=g) * 0.7).to(torch.bfloat16).cuda()
    q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
    out, lse = ops.attn_fwd_dense_launch(q, k, v, A.default_scale(d), False)
    ref_o, ref_lse = A.varlen_attention(q.cpu().contiguous(), k.cpu().contiguous(), v.cpu().contiguous(), [0, s], A.default_scale(d), False)
    torch.cuda.synchronize()
    assert rel_err(out, ref_o) <= REL_TOL
    assert np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL
    with pytest.raises(_lib.LvAttnError, match="k_rows >= q_rows"):
        ops.attn_fwd_dense_launch(q, k[:100], v[:100], 0.125, True)
    with pytest.raises(_lib.LvAttnError):
        ops.attn_fwd_dense_launch(q[:, :, :32], k[:, :, :32], v[:, :, :32], 0.125, False)     # head dim 32


def test_head_dim_128_row_strides_with_and_without_256_byte_granularity(forced_forward_kernel):
    """Head dim 128: q / k / v as views of one packed [rows, (Hq + 2 Hkv) * 128] buffer (row strides that are multiples of
    128 elements: every kernel takes them, the one-wave kernel included), and as views of a buffer with 8 padding elements
    per row (strides the one-wave kernel does not take: forced or not, the launch falls back to the ping-pong kernel).
    Both match the oracle under every forward kernel."""
    from long_vita_amd import ops
    Hq, Hkv, D = 4, 2, 128
    cu = [0, 300, 301, 1000]
    width = (Hq + 2 * Hkv) * D
    for pad in (0, 8):
        g = torch.Generator().manual_seed(21 + pad)
        buf = (torch.randn(cu[-1], width + pad, generator=g) * 0.7).to(torch.bfloat16).cuda()
        q = buf[:, :Hq * D].unflatten(1, (Hq, D))
        k = buf[:, Hq * D:(Hq + Hkv) * D].unflatten(1, (Hkv, D))
        v = buf[:, (Hq + Hkv) * D:width].unflatten(1, (Hkv, D))
        assert k.stride(0) == width + pad and not k.is_contiguous()
        out, lse = ops.varlen_attention(q, k, v, cu, A.default_scale(D), True)
        ref_o, ref_lse = A.varlen_attention(q.cpu().contiguous(), k.cpu().contiguous(), v.cpu().contiguous(), cu, A.default_scale(D), True)
        torch.cuda.synchronize()
        assert rel_err(out, ref_o) <= REL_TOL, pad
        assert np.abs(lse.cpu().numpy().astype(np.float64) - ref_lse).max() <= LSE_TOL, pad


def test_full_size_backward_invariants_cfg3():
    """Backward at the 131,072-token / 40-8-head size of the forward+backward measurements, checked through identities
    that hold at any size: rows of P sum to one, so  sum_j dV[j] = sum over the group's heads and all rows of dO;
    dQ = scale dS K and dK = scale dS^T Q, so  sum_i <q_i, dq_i> (over the group's heads) = sum_j <k_j, dk_j>."""
    from long_vita_amd import ops
    S, Hq, Hkv, D = 131072, 40, 8, 128
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(3)
    rnd = lambda *shape: (torch.randn(*shape, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v, do = rnd(S, Hq, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hq, D)
    out, _ = ops.varlen_attention(q, k, v, [0, S], A.default_scale(D), True)
    out.backward(do)
    torch.cuda.synchronize()
    rep = Hq // Hkv
    dv_sum = v.grad.double().sum(0)                                            # [Hkv, D]
    do_sum = do.double().sum(0).reshape(Hkv, rep, D).sum(1)
    assert (dv_sum - do_sum).abs().max().item() <= 1e-2 * do_sum.abs().max().item()
    qdq = (q.detach().double() * q.grad.double()).sum((0, 2)).reshape(Hkv, rep).sum(1)   # [Hkv]
    kdk = (k.detach().double() * k.grad.double()).sum((0, 2))
    assert (qdq - kdk).abs().max().item() <= 2e-2 * max(qdq.abs().max().item(), kdk.abs().max().item())


@pytest.mark.parametrize("dq,dkdv", [("lockstep", "single"), ("pp", "rs1"), ("pp", "rs"), ("pp", "g")])
def test_every_backward_kernel_variant(dq, dkdv):
    """The backward kernels that are not the default stay selectable and correct."""
    from long_vita_amd import ops, _lib
    _lib.check(_lib.lib().lvattn_select_bwd_kernels(dq.encode(), dkdv.encode()), "select")
    try:
        for cu, Hq, Hkv, D, causal in [([0, 100, 356, 357, 700, 1000], 10, 2, 128, True), ([0, 200, 520], 4, 2, 128, False),
                                       ([0, 130, 450], 4, 2, 64, True)]:
            T = cu[-1]
            q, k, v = make_qkv(T, Hq, Hkv, D, seed=T + D)
            g = torch.Generator().manual_seed(5)
            do = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
            scale = A.default_scale(D)
            rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, causal)
            qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
            out, _ = ops.varlen_attention(qc, kc, vc, cu, scale, causal)
            out.backward(do.cuda())
            torch.cuda.synchronize()
            assert rel_err(qc.grad, rdq) <= REL_TOL and rel_err(kc.grad, rdk) <= REL_TOL and rel_err(vc.grad, rdv) <= REL_TOL, (dq, dkdv, cu)
    finally:
        _lib.check(_lib.lib().lvattn_select_bwd_kernels(b"default", b"default"), "select")


def test_full_size_backward_invariants_cfg3_packed():
    """cfg3-B at full size: 131,072 tokens packed into segments with log-uniform lengths in [256, 32768] (seed 1234, the
    benchmark's packing), 40 / 8 heads, forward + backward.  The two size-independent identities of the single-segment
    test hold PER SEGMENT (attention never crosses a boundary): sum_j dV[j] over a segment's keys = sum of dO over the
    segment's rows and the group's heads;  sum_i <q_i, dq_i> = sum_j <k_j, dk_j> over the segment.  A sample of rows is
    also checked against the float64 oracle on its own segment."""
    import math
    from long_vita_amd import ops
    S, Hq, Hkv, D = 131072, 40, 8, 128
    rs = np.random.RandomState(1234)
    lens = []
    while sum(lens) < S:
        lens.append(int(math.exp(rs.uniform(math.log(256), math.log(32768)))))
    cu = [0]
    for L in lens:
        cu.append(min(S, cu[-1] + L))
    cu[-1] = S
    assert len(cu) > 5
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(4)
    rnd = lambda *shape: (torch.randn(*shape, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v, do = rnd(S, Hq, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hq, D)
    scale = A.default_scale(D)
    out, lse = ops.varlen_attention(q, k, v, cu, scale, True)
    out.backward(do)
    torch.cuda.synchronize()
    rep = Hq // Hkv
    for a, b in zip(cu[:-1], cu[1:]):
        dv_sum = v.grad[a:b].double().sum(0)
        do_sum = do[a:b].double().sum(0).reshape(Hkv, rep, D).sum(1)
        assert (dv_sum - do_sum).abs().max().item() <= 1e-2 * max(do_sum.abs().max().item(), 1.0), (a, b)
        qdq = (q.detach()[a:b].double() * q.grad[a:b].double()).sum((0, 2)).reshape(Hkv, rep).sum(1)
        kdk = (k.detach()[a:b].double() * k.grad[a:b].double()).sum((0, 2))
        assert (qdq - kdk).abs().max().item() <= 2e-2 * max(qdq.abs().max().item(), kdk.abs().max().item(), 1.0), (a, b)
    # rows of the forward against the oracle: first, a middle and the last row of three segments
    for si in (0, len(cu) // 2, len(cu) - 2):
        a, b = cu[si], cu[si + 1]
        for row in (a, (a + b) // 2, b - 1):
            n = row - a + 1
            for h in (0, 17, 39):
                kk = k.detach()[a:row + 1, h // rep].double()
                s_ = (kk @ q.detach()[row, h].double()) * scale
                p = torch.softmax(s_, 0)
                want = p @ v.detach()[a:row + 1, h // rep].double()
                got = out.detach()[row, h].double()
                assert (got - want).abs().max().item() <= REL_TOL * max(want.abs().max().item(), 1e-2), (row, h, n)
                assert abs(lse[h, row].item() - torch.logsumexp(s_, 0).item()) <= LSE_TOL


@pytest.mark.parametrize("dkdv", ["g", "rs"])
def test_backward_with_padded_row_constants_full_of_nan(dkdv):
    """lse and delta live in [hq, lse_stride] arrays whose rows are longer than the packed batch (lse_stride > rows); the
    padding holds NaN here, and so does everything behind a segment's last row that is not another segment's.  The one-wave
    dK/dV kernel fetches row constants sixteen bytes at a time and so reads past a ragged segment end: what it reads
    there must not reach dk / dv (segments of 1000 and 41 rows: partial last slices; d = 128, grouped heads)."""
    from long_vita_amd import ops, _lib
    cu, Hq, Hkv, D = [0, 1000, 1041], 10, 2, 128
    T = cu[-1]
    q, k, v = make_qkv(T, Hq, Hkv, D, seed=77)
    g = torch.Generator().manual_seed(6)
    do = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, True)
    qc, kc, vc, doc = (t.cuda() for t in (q, k, v, do))
    stride = T + 87
    lse = torch.full((Hq, stride), float("nan"), dtype=torch.float32, device="cuda")[:, :T]
    out = torch.empty_like(qc)
    desc_q = ops.blocks_on_device(cu, True, False, qc.device)
    desc_k = ops.blocks_on_device(cu, True, True, qc.device)
    ops.attn_fwd_launch(qc, kc, vc, desc_q, scale, True, out, lse)
    assert lse.stride(0) == stride
    delta_buf = torch.full((Hq, stride), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().lvattn_bwd_delta(out.data_ptr(), doc.data_ptr(), delta_buf.data_ptr(), T, Hq, D, out.stride(0), out.stride(1),
                                          doc.stride(0), doc.stride(1), stride, torch.cuda.current_stream().cuda_stream), "delta")
    assert torch.isnan(delta_buf[:, T:]).all() and torch.isfinite(delta_buf[:, :T]).all()
    _lib.check(_lib.lib().lvattn_select_bwd_kernels(b"pp", dkdv.encode()), "select")
    try:
        dq, dk, dv, _ = ops.attn_bwd_launch(qc, kc, vc, out, doc, lse, desc_q, desc_k, scale, True, delta=delta_buf)
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib().lvattn_select_bwd_kernels(b"default", b"default"), "select")
    for got, ref, name in ((dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")):
        assert torch.isfinite(got.float()).all(), name
        assert rel_err(got, ref) <= REL_TOL, name
