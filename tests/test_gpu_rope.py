"""Rotary position embedding: the HIP kernel (through the C ABI) against the oracle and the
reference-made fixture.  Bit-exact: the kernel reproduces the reference's bf16 rounding sequence."""
import os

import numpy as np
import pytest
import torch

from oracle import rope_ref as R

pytestmark = pytest.mark.gpu


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "rope.npz"))


def _tables(freqs_np):
    """cos / sin bf16 tables computed by the ORACLE (numpy), so the comparison isolates the kernel."""
    rot = freqs_np.shape[-1]
    flat = freqs_np.reshape(-1, rot).astype(np.float32)
    cos_t = torch.from_numpy(R.bf16_round(np.cos(flat, dtype=np.float32))).to(torch.bfloat16).cuda()
    sin_t = torch.from_numpy(R.bf16_round(np.sin(flat, dtype=np.float32))).to(torch.bfloat16).cuda()
    return cos_t, sin_t


def _bf(x):
    return torch.from_numpy(x).to(torch.bfloat16).cuda()


def test_kernel_bit_exact_on_reference_fixture(golden_dir):
    from long_vita_amd import ops
    g = _g(golden_dir)
    for tag in ("a", "b", "c", "partial"):
        t, freqs = g[f"bshd.{tag}.t"], g[f"bshd.{tag}.freqs"]
        s, b, h, d = t.shape
        cos_t, sin_t = _tables(freqs)
        out = ops.rope_launch(_bf(t).view(s * b, h, d), cos_t, sin_t, None, b)
        assert np.array_equal(out.float().cpu().numpy().reshape(t.shape), g[f"bshd.{tag}.out"]), tag
        if f"bshd.{tag}.dt" in g.files:
            dt = ops.rope_launch(_bf(g[f"bshd.{tag}.dout"]).view(s * b, h, d), cos_t, sin_t, None, b, conj=True)
            assert np.array_equal(dt.float().cpu().numpy().reshape(t.shape), g[f"bshd.{tag}.dt"]), tag
    # packed: positions restart per segment
    cos_t, sin_t = _tables(g["thd.freqs"])
    pos = torch.from_numpy(R.thd_position_ids(g["thd.cu"]).astype(np.int32)).cuda()
    out = ops.rope_launch(_bf(g["thd.t"]), cos_t, sin_t, pos, 1)
    assert np.array_equal(out.float().cpu().numpy(), g["thd.out"])


def test_kernel_vs_oracle_qwen_shape_strided_and_in_place():
    """Qwen2.5 head shape, q and k as strided views of one fused qkv buffer, in-place rotation."""
    from long_vita_amd import ops
    rs = np.random.RandomState(5)
    S, HQ, HKV, D = 777, 40, 8, 128
    freqs = R.angle_table(S, D, 1000000)
    cos_t, sin_t = _tables(freqs)
    qkv = torch.from_numpy(rs.randn(S, (HQ + 2 * HKV) * D).astype(np.float32)).to(torch.bfloat16).cuda()
    q = qkv[:, : HQ * D].view(S, HQ, D)
    k = qkv[:, HQ * D:(HQ + HKV) * D].view(S, HKV, D)
    want_q = R.apply_bshd(q.float().cpu().numpy()[:, None], freqs)[:, 0]
    want_k = R.apply_bshd(k.float().cpu().numpy()[:, None], freqs)[:, 0]
    v_before = qkv[:, (HQ + HKV) * D:].clone()
    ops.rope_launch(q, cos_t, sin_t, out=q)
    ops.rope_launch(k, cos_t, sin_t, out=k)
    assert np.array_equal(q.float().cpu().numpy(), want_q)
    assert np.array_equal(k.float().cpu().numpy(), want_k)
    assert torch.equal(qkv[:, (HQ + HKV) * D:], v_before)      # v untouched


def test_autograd_and_reference_api(golden_dir):
    """apply_rotary_pos_emb (the reference's entry point): forward and gradient against the fixture.
    The tables come from torch's device cos / sin here, so allow the last bf16 bit of cos / sin to differ:
    at most a few elements may move by one bf16 step."""
    from types import SimpleNamespace
    from long_vita_amd.rotary_pos_embedding import apply_rotary_pos_emb
    g = _g(golden_dir)
    cfg = SimpleNamespace(rotary_interleaved=False, apply_rope_fusion=True)
    for tag in ("a", "b", "c"):
        t = _bf(g[f"bshd.{tag}.t"]).requires_grad_(True)
        freqs = torch.from_numpy(g[f"bshd.{tag}.freqs"]).cuda()
        out = apply_rotary_pos_emb(t, freqs, cfg)
        out.backward(_bf(g[f"bshd.{tag}.dout"]))
        for got, want in ((out, g[f"bshd.{tag}.out"]), (t.grad, g[f"bshd.{tag}.dt"])):
            got = got.detach().float().cpu().numpy()
            scale = np.maximum(np.abs(want), 1e-3)
            assert np.max(np.abs(got - want) / scale) <= 2.0 ** -6, tag     # <= 2 bf16 steps
            assert np.mean(got != want) < 0.02, tag
    t = _bf(g["thd.t"])
    out = apply_rotary_pos_emb(t, torch.from_numpy(g["thd.freqs"]).cuda(), cfg, cu_seqlens=torch.from_numpy(g["thd.cu"]).cuda())
    got, want = out.float().cpu().numpy(), g["thd.out"]
    assert np.mean(got != want) < 0.02 and np.max(np.abs(got - want)) <= 2.0 ** -6 * np.max(np.abs(want))


def test_rope_rotation_properties_at_full_size():
    """131072 rows x 40 heads x 128 (cfg3 size): conj(rope(x)) returns x up to bf16 rounding, row norms
    are preserved, and position 0 is the identity."""
    from long_vita_amd import ops
    S, H, D = 131072, 40, 128
    torch.manual_seed(0)
    x = torch.randn(S, H, D, device="cuda", dtype=torch.bfloat16)
    inv = 1.0 / (1000000 ** (torch.arange(0, D, 2, dtype=torch.float32, device="cuda") / D))
    ang = torch.outer(torch.arange(S, device="cuda", dtype=torch.float32), inv)
    ang = torch.cat((ang, ang), -1)
    cos_t, sin_t = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    y = ops.rope_launch(x, cos_t, sin_t)
    assert torch.equal(y[0], x[0])
    back = ops.rope_launch(y, cos_t, sin_t, conj=True)
    err = (back.float() - x.float()).abs().max().item()
    assert err <= 0.08, err          # three bf16 roundings each way on |x| <= ~5
    n0 = x[::997].float().norm(dim=-1)
    n1 = y[::997].float().norm(dim=-1)
    assert torch.allclose(n0, n1, rtol=1e-2)


def test_rope_argument_errors():
    from long_vita_amd import _lib, ops
    x = torch.zeros(8, 2, 64, device="cuda", dtype=torch.bfloat16)
    tab = torch.zeros(4, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(_lib.LvAttnError):    # 8 rows, table of 4
        ops.rope_launch(x, tab, tab)
    with pytest.raises(_lib.LvAttnError):    # rotary width not a multiple of 16
        ops.rope_launch(x, tab[:, :24].contiguous(), tab[:, :24].contiguous(), None, 2)
    ops.rope_launch(x, tab, tab, None, 2)     # 8 rows / 2 = 4 table rows: fine


def _qwen_case(S, seed, HQ=10, HKV=2, D=128):
    rs = np.random.RandomState(seed)
    freqs = R.angle_table(S, D, 1000000)
    cos_t, sin_t = _tables(freqs)
    qkv = torch.from_numpy((0.7 * rs.randn(S, (HQ + 2 * HKV) * D)).astype(np.float32)).to(torch.bfloat16).cuda()
    q = qkv[:, : HQ * D].view(S, HQ, D)                       # strided views of one fused projection output
    k = qkv[:, HQ * D:(HQ + HKV) * D].view(S, HKV, D)
    v = qkv[:, (HQ + HKV) * D:].view(S, HKV, D)
    do = torch.from_numpy(rs.randn(S, HQ, D).astype(np.float32)).to(torch.bfloat16).cuda()
    return freqs, cos_t, sin_t, q, k, v, do


@pytest.mark.parametrize("cu", [[0, 777], [0, 258, 1290, 1291, 3000], [0, 513, 1024]])
def test_rotation_inside_the_attention_kernel_equals_the_two_pass_path(cu, kernel_with_rotation):
    """The query rotation folded into the attention kernel (lvattn_fwd_rope) + the standalone K rotation against the
    two-pass path (lv_rope on q and k, then lvattn_fwd / lvattn_bwd): out, lse, the rotated queries the kernel writes
    out, and dq / dk / dv must be IDENTICAL bit for bit; and within the attention tolerance of the float64 oracle on the
    oracle-rotated inputs."""
    from long_vita_amd import ops
    from oracle import attention_ref as A
    S = cu[-1]
    freqs, cos_t, sin_t, q, k, v, do = _qwen_case(S, 11 + len(cu))
    scale = 1.0 / np.sqrt(128.0)
    # two-pass reference path on this package's kernels
    q2 = ops.rope(q.detach().clone().requires_grad_(True), cos_t, sin_t)
    qa = q.detach().clone().requires_grad_(True)
    ka = k.detach().clone().requires_grad_(True)
    va = v.detach().clone().requires_grad_(True)
    q_rot, k_rot = ops.rope(qa, cos_t, sin_t), ops.rope(ka, cos_t, sin_t)
    # (sweeps left whole on both sides: the fused entry point never cuts them, and a launch this small would otherwise
    # be cut by the planner, whose pieces combine in another rounding order)
    ops.set_split_override(0, 0)
    try:
        out_a, lse_a = ops.varlen_attention(q_rot, k_rot, va, cu, scale, True)
        out_a.backward(do)
    finally:
        ops.set_split_override(None, None)
    # fused
    qb = q.detach().clone().requires_grad_(True)
    kb = k.detach().clone().requires_grad_(True)
    vb = v.detach().clone().requires_grad_(True)
    out_b, lse_b = ops.rope_varlen_attention(qb, kb, vb, cos_t, sin_t, cu, scale, True)
    out_b.backward(do)
    assert torch.equal(out_a, out_b) and torch.equal(lse_a, lse_b)
    for a, b, name in ((qa.grad, qb.grad, "dq"), (ka.grad, kb.grad, "dk"), (va.grad, vb.grad, "dv")):
        assert torch.equal(a, b), name
    # the side output of the kernel: the rotated queries
    desc = ops.blocks_on_device(cu, True, False, q.device)
    side = torch.zeros_like(q2)
    ops.attn_fwd_rope_launch(q, k_rot.detach(), v, desc, scale, True, cos_t, sin_t, q_rot=side)
    assert torch.equal(side, q2.detach())
    # oracle: rotate with the oracle, attend in float64
    want_q = R.apply_bshd(q.float().cpu().numpy()[:, None], freqs)[:, 0]
    want_k = R.apply_bshd(k.float().cpu().numpy()[:, None], freqs)[:, 0]
    ref_o, ref_lse = A.varlen_attention(torch.from_numpy(want_q), torch.from_numpy(want_k), v.float().cpu(), cu, scale)
    err = np.abs(out_b.detach().float().cpu().numpy() - ref_o).max() / np.abs(ref_o).max()
    assert err <= 2e-2, err
    assert np.abs(lse_b.cpu().numpy() - ref_lse).max() <= 1e-3


def test_fused_rotation_against_the_default_plain_path_on_a_long_sweep():
    """A long single segment, where a plain launch takes the one-wave forward kernel by the library's own choice while
    the fused launch stays on the ping-pong kernel (the rotation lives there): the two results are not bit-identical
    (different kernels round differently) but agree to kernel rounding, outputs and log-sum-exp."""
    from long_vita_amd import ops
    S, HQ_, HKV_, D = 65536, 2, 1, 128
    rs = np.random.RandomState(5)
    ang = torch.from_numpy(R.angle_table(S, D, 1000000)).reshape(S, D)
    cos_t, sin_t = torch.cos(ang).to(torch.bfloat16).cuda(), torch.sin(ang).to(torch.bfloat16).cuda()
    q, k, v = (torch.from_numpy((0.5 * rs.randn(S, h, D)).astype(np.float32)).to(torch.bfloat16).cuda() for h in (HQ_, HKV_, HKV_))
    scale = 1.0 / np.sqrt(128.0)
    desc = ops.blocks_on_device([0, S], True, False, q.device)
    assert desc.lv_mean_keys >= 32768                                   # the plain launch's default is the one-wave kernel
    k_rot = ops.rope_launch(k, cos_t, sin_t)
    plain, lse_p = ops.attn_fwd_launch(ops.rope_launch(q, cos_t, sin_t), k_rot, v, desc, scale, True)
    fused, lse_f = ops.attn_fwd_rope_launch(q, k_rot, v, desc, scale, True, cos_t, sin_t)
    torch.cuda.synchronize()
    err = (fused.float() - plain.float()).abs().max().item() / plain.float().abs().max().item()
    assert err <= 1e-2, err
    assert (lse_f - lse_p).abs().max().item() <= 1e-4


def test_rope_pack_kv_and_packed_positions(golden_dir):
    """kv[0] = rope(k) bit-exact against the reference-made fixture (packed position ids), kv[1] = v, from strided
    inputs in one pass."""
    from long_vita_amd import ops
    g = _g(golden_dir)
    cos_t, sin_t = _tables(g["thd.freqs"])
    pos = torch.from_numpy(R.thd_position_ids(g["thd.cu"]).astype(np.int32)).cuda()
    t = _bf(g["thd.t"])
    T, H, D = t.shape
    both = torch.zeros(T, 2 * H * D + 16, dtype=torch.bfloat16, device="cuda")
    kview = both[:, :H * D].view(T, H, D)
    vview = both[:, H * D + 16:].view(T, H, D)
    kview.copy_(t)
    vview.copy_(torch.flip(t, dims=(0,)))
    kv = ops.rope_pack_kv(kview, vview, cos_t, sin_t, pos, 1)
    assert np.array_equal(kv[0].float().cpu().numpy(), g["thd.out"])
    assert torch.equal(kv[1], vview)
    # position = row index, Qwen shape
    freqs, cos_t, sin_t, q, k, v, _ = _qwen_case(515, 3)
    kv = ops.rope_pack_kv(k, v, cos_t, sin_t)
    assert torch.equal(kv[0], ops.rope_launch(k, cos_t, sin_t)) and torch.equal(kv[1], v)


def test_fused_rope_argument_errors():
    from long_vita_amd import _lib, ops
    q = torch.zeros(64, 2, 64, device="cuda", dtype=torch.bfloat16)
    tab = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    desc = ops.blocks_on_device([0, 64], True, False, q.device)
    with pytest.raises(_lib.LvAttnError):        # head dim 64: the fused rotation is built for 128
        ops.attn_fwd_rope_launch(q, q, q, desc, 0.125, True, tab, tab)
    q = torch.zeros(64, 2, 128, device="cuda", dtype=torch.bfloat16)
    tab = torch.zeros(32, 128, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(_lib.LvAttnError):        # 64 rows, table of 32
        ops.attn_fwd_rope_launch(q, q, q, desc, 0.09, True, tab, tab)
    with pytest.raises(_lib.LvAttnError):
        ops.rope_pack_kv(q, q, tab, tab)


def test_rope_flash_attention_entry(golden_dir, kernel_with_rotation):
    """The Megatron-shaped entry ([s, 1, h, d] tensors, fp32 angle tensor) equals apply_rotary_pos_emb followed by the
    packed attention, bit for bit."""
    from types import SimpleNamespace
    from long_vita_amd import ops
    from long_vita_amd.rotary_pos_embedding import apply_rotary_pos_emb, rope_flash_attention
    S, HQ, HKV, D = 600, 10, 2, 128
    freqs_np, _, _, q, k, v, _ = _qwen_case(S, 21)
    freqs = torch.from_numpy(freqs_np).cuda()
    cfg = SimpleNamespace(rotary_interleaved=False)
    ends = [100, 356, 600]
    qr = apply_rotary_pos_emb(q.contiguous().view(S, 1, HQ, D), freqs, cfg)
    kr = apply_rotary_pos_emb(k.contiguous().view(S, 1, HKV, D), freqs, cfg)
    want, _ = ops.varlen_attention(qr.view(S, HQ, D), kr.view(S, HKV, D), v, [0] + ends, 1.0 / np.sqrt(D), True)
    got = rope_flash_attention(q.view(S, 1, HQ, D), k.view(S, 1, HKV, D), v.view(S, 1, HKV, D), freqs, ends)
    assert got.shape == (S, 1, HQ * D) and torch.equal(got.view(S, HQ, D), want)
