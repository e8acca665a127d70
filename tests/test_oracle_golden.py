"""The CPU oracle against the fixtures produced by the reference's own code
(tests/golden/make_golden.py).  No GPU needed."""
import os

import numpy as np
import torch

from oracle import attention_ref as A
from oracle import indexing_ref as I

CASES = [("gqa_small", 48, 1, 4, 2, 16), ("gqa_b2", 33, 2, 10, 2, 32),
         ("mha_d64", 65, 1, 4, 4, 64), ("qwen_ratio", 96, 1, 10, 2, 128)]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_eager_restatement_matches_reference_wrapper(golden_dir):
    g = _load(golden_dir, "eager_attention.npz")
    for name, sq, b, np_, ng, hn in CASES:
        q, k, v = (torch.from_numpy(g[f"{name}.{x}"]) for x in "qkv")
        out = A.eager_attention(q, k, v, A.causal_mask(sq, sq), float(hn) ** 0.5)
        assert torch.equal(out, torch.from_numpy(g[f"{name}.out"])), name
        out2 = A.eager_attention(q, k, v, None, float(hn) ** 0.5)
        np.testing.assert_allclose(out2.numpy(), g[f"{name}.out_nomask"], rtol=0, atol=1e-6)


def test_fp64_attention_matches_reference_eager(golden_dir):
    """The float64 varlen/dense restatement agrees with the reference's eager
    fp32 output to fp32 rounding: same softmax, GQA mapping, scale, layout."""
    g = _load(golden_dir, "eager_attention.npz")
    for name, sq, b, np_, ng, hn in CASES:
        q, k, v = (g[f"{name}.{x}"] for x in "qkv")
        ref = g[f"{name}.out"].reshape(sq, b, np_, hn)
        ref_nm = g[f"{name}.out_nomask"].reshape(sq, b, np_, hn)
        for bi in range(b):
            o, lse = A.varlen_attention(q[:, bi], k[:, bi], v[:, bi], [0, sq], A.default_scale(hn))
            np.testing.assert_allclose(o, ref[:, bi], rtol=0, atol=2e-5)
            o, _ = A.varlen_attention(q[:, bi], k[:, bi], v[:, bi], [0, sq], A.default_scale(hn), causal=False)
            np.testing.assert_allclose(o, ref_nm[:, bi], rtol=0, atol=2e-5)
            assert np.isfinite(lse).all()


def test_fp64_grad_matches_autograd_of_eager():
    torch.manual_seed(3)
    T, Hq, Hkv, D = 40, 4, 2, 16
    cu = [0, 7, 25, 40]
    q = torch.randn(T, Hq, D, dtype=torch.float64, requires_grad=True)
    k = torch.randn(T, Hkv, D, dtype=torch.float64, requires_grad=True)
    v = torch.randn(T, Hkv, D, dtype=torch.float64, requires_grad=True)
    do = torch.randn(T, Hq, D, dtype=torch.float64)
    # block-diagonal causal mask = what packed varlen attention means
    seg = torch.zeros(T, dtype=torch.long)
    for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
        seg[a:b] = i
    mask = (seg[:, None] != seg[None, :]) | torch.triu(torch.ones(T, T, dtype=torch.bool), 1)
    out = A.eager_attention(q[:, None], k[:, None], v[:, None], mask[None, None], D ** 0.5)
    out.backward(do.reshape(T, 1, Hq * D))
    dq, dk, dv = A.varlen_attention_grad(q, k, v, do, cu, A.default_scale(D))
    o64, _ = A.varlen_attention(q, k, v, cu, A.default_scale(D))
    # the eager path's softmax runs in fp32 (scale_mask_softmax), hence 1e-5 not 1e-12
    np.testing.assert_allclose(o64.reshape(T, 1, -1), out.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(dq, q.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(dk, k.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(dv, v.grad.numpy(), atol=1e-5)


def test_ring_simulation_equals_full_causal():
    rng = np.random.default_rng(0)
    for cp in (1, 2, 4):
        S, Hq, Hkv, D = 64, 4, 2, 8
        q, k, v = rng.standard_normal((S, Hq, D)), rng.standard_normal((S, Hkv, D)), rng.standard_normal((S, Hkv, D))
        full, lse = A.varlen_attention(q, k, v, [0, S], 0.3)
        res = A.ring_attention_simulated(q, k, v, 0.3, cp)
        outs = [r[0][None] for r in res]
        got = I.unzigzag_concat(outs, cp, seq_dim=1)[0]
        np.testing.assert_allclose(got, full, atol=1e-12)
        lses = [r[1] for r in res]
        np.testing.assert_allclose(I.unzigzag_concat(lses, cp, seq_dim=1), lse, atol=1e-12)


def test_indexing_oracle_matches_reference(golden_dir):
    g = _load(golden_dir, "indexing.npz")
    np.testing.assert_array_equal(I.index_of_a_in_b(g["ioab.a"], g["ioab.b"]), g["ioab.out"])
    for S, cp in [(64, 2), (256, 4), (1024, 8)]:
        tag = f"zz.S{S}.cp{cp}"
        tokens, ext, images = g[f"{tag}.tokens"], g[f"{tag}.external_indices"], g[f"{tag}.images"]
        per_rank = []
        for r in range(cp):
            got = I.zigzag_slice(tokens, cp, r, 1)
            np.testing.assert_array_equal(got, g[f"{tag}.r{r}.tokens"])
            per_rank.append(got)
            sel, src, tgt = I.cp_image_remap(ext, S, cp, r)
            if sel is None:
                assert f"{tag}.r{r}.src" not in g.files
                np.testing.assert_array_equal(images, g[f"{tag}.r{r}.images"])
            else:
                np.testing.assert_array_equal(images[sel], g[f"{tag}.r{r}.images"])
                np.testing.assert_array_equal(src, g[f"{tag}.r{r}.src"])
                np.testing.assert_array_equal(tgt, g[f"{tag}.r{r}.tgt"])
        np.testing.assert_array_equal(I.unzigzag_concat(per_rank, cp, 1), tokens)
    np.testing.assert_array_equal(
        I.collate_actual_seq_len([g["collate.in_actual0"], g["collate.in_actual1"], g["collate.in_actual2"]]),
        g["collate.actual_seq_len"])
    np.testing.assert_array_equal(
        I.collate_image_indices([g["collate.in_img0"], None, g["collate.in_img2"]]), g["collate.image_indices"])
    np.testing.assert_array_equal(I.pixel_shuffle(g["ps.small.in"]), g["ps.small.out"])
    full = np.arange(32 * 32 * 1024, dtype=np.int32).reshape(1, 32, 32, 1024)
    np.testing.assert_array_equal(I.pixel_shuffle(full), g["ps.full.index_map"])
    for cp in (2, 4):
        for r in range(cp):
            np.testing.assert_array_equal(I.zigzag_slice(g["rope.in"], cp, r, 0), g[f"rope.cp{cp}.r{r}"])


def test_pixel_shuffle_v2_oracle_matches_reference_fixture(golden_dir):
    """The stride form of the token compression (pretrain_lcvlm.py:558-570): even and odd (zero-padded) grids, strides
    2 / 3 / 4, and the full 32 x 32 x 1024 grid as an index map, against outputs of the reference's own method."""
    g = np.load(os.path.join(golden_dir, "pixel_shuffle_v2.npz"))
    for tag in ("even", "odd", "s4", "s3"):
        np.testing.assert_array_equal(I.pixel_shuffle_v2(g[f"{tag}.in"], int(g[f"{tag}.stride"])), g[f"{tag}.out"])
    full = np.arange(1, 1 + 32 * 32 * 1024, dtype=np.int32).reshape(1, 32, 32, 1024)
    np.testing.assert_array_equal(I.pixel_shuffle_v2(full, 2), g["full.index_map"])


def test_pack_samples_semantics():
    pos0, n0 = I.image_token_layout(3, 2, 4)
    assert n0 == 3 + 2 * 6 and pos0.tolist() == [[4, 5, 6, 7], [10, 11, 12, 13]]
    actual, pids, img = I.pack_samples([15, 9], [pos0, None], 32)
    assert actual.tolist() == [15, 32]
    assert pids.tolist() == list(range(15)) + list(range(9)) + [8] * 8
    assert img.shape == (2, 2, 4) and img[1].tolist() == pos0.tolist()
    actual, pids, img = I.pack_samples([10, 15], [None, pos0], 20)
    assert actual.tolist() == [10, 20] and len(pids) == 20
    assert img[1].tolist() == pos0.tolist()      # first image-bearing sample is NOT shifted (reference quirk)
    assert I.cu_seqlens_from_actual(actual).tolist() == [0, 10, 20]


def test_pack_samples_matches_reference_add_ret_process_ret(golden_dir):
    g = _load(golden_dir, "packing.npz")
    img_a = [[2, 3, 4, 5], [8, 9, 10, 11]]
    img_b = [[1, 2, 3, 4]]
    cases = {"pad": ([(15, img_a), (9, None), (6, img_b)], 40, False),
             "trunc": ([(15, None), (9, img_b), (20, None)], 32, False),
             "late_image": ([(10, None), (12, img_b), (8, img_a)], 32, False),
             "shift": ([(15, img_a), (9, None)], 32, True)}
    for name, (samples, max_len, shift) in cases.items():
        actual, pids, img = I.pack_samples([L for L, _ in samples], [i for _, i in samples], max_len, shift_token=shift)
        np.testing.assert_array_equal(actual, g[f"{name}.actual_seq_len"], err_msg=name)
        np.testing.assert_array_equal(pids, g[f"{name}.position_ids"], err_msg=name)
        np.testing.assert_array_equal(img, g[f"{name}.image_indices"], err_msg=name)


def _bits_to_f32(bits: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns -> float32 values."""
    return (bits.astype(np.uint32) << 16).view(np.float32)


GRAD_CASES = ["qwen5to1_causal", "qwen5to1_packed", "gqa_packed_d64", "mha_packed_small"]


def test_fp64_attention_and_grad_match_autograd_of_reference_wrapper(golden_dir):
    """Outputs and dq / dk / dv of the float64 restatement against torch autograd through the REFERENCE's eager
    wrapper (fixture), for causal and packed (block-diagonal-causal) masks, 5:1 head ratio, d = 128 past one
    256-row block.  The wrapper's softmax runs in fp32, hence 1e-4 of the tensor's scale rather than 1e-12."""
    g = _load(golden_dir, "eager_attention_grad.npz")
    for name in GRAD_CASES:
        sq, np_, ng, hn = (int(x) for x in g[f"{name}.meta"])
        cu = [int(x) for x in g[f"{name}.cu"]]
        q, k, v = (_bits_to_f32(g[f"{name}.{x}"])[:, 0] for x in "qkv")
        do = _bits_to_f32(g[f"{name}.dout"])[:, 0].reshape(sq, np_, hn)
        o, _ = A.varlen_attention(q, k, v, cu, A.default_scale(hn), True)
        dq, dk, dv = A.varlen_attention_grad(q, k, v, do, cu, A.default_scale(hn), True)
        for got, key in ((o.reshape(sq, 1, -1), "out"), (dq[:, None], "dq"), (dk[:, None], "dk"), (dv[:, None], "dv")):
            ref = g[f"{name}.{key}"].astype(np.float64)
            assert got.shape == ref.shape, (name, key)
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (name, key, np.abs(got - ref).max())


def test_scatter_oracle_matches_reference_embedding_forward(golden_dir):
    """oracle scatter (+ its gradient) against LanguageModelEmbedding.forward run in bf16 with autograd (fixture):
    pure data movement, so bit-exact."""
    g = _load(golden_dir, "embedding_scatter.npz")
    for tag in ("indices", "pre_len", "src_tgt", "none"):
        we, feats, do = (_bits_to_f32(g[f"{tag}.{x}"]) for x in ("we", "features", "dout"))
        kw = {}
        if tag == "indices":
            kw["indices"] = g["indices.indices"]
        elif tag == "pre_len":
            kw["pre_len"] = int(g["pre_len.pre_len"])
        elif tag == "src_tgt":
            kw["src_indices"], kw["tgt_indices"] = g["src_tgt.src_indices"], g["src_tgt.tgt_indices"]
        ref_df = _bits_to_f32(g[f"{tag}.dfeatures"])
        if tag == "pre_len":      # the fixture handed the first b images' features over; the rest got no gradient
            assert not ref_df[we.shape[0]:].any()
            feats, ref_df = feats[:we.shape[0]], ref_df[:we.shape[0]]
        out = I.scatter_features(we, feats, **kw).transpose(1, 0, 2)          # [b,s,h] -> [s,b,h] (:142)
        np.testing.assert_array_equal(out, _bits_to_f32(g[f"{tag}.out"]), err_msg=tag)
        dwe, df = I.scatter_features_grad(do.transpose(1, 0, 2), feats.shape, **kw)
        np.testing.assert_array_equal(dwe, _bits_to_f32(g[f"{tag}.dwe"]), err_msg=tag)
        np.testing.assert_array_equal(df, ref_df, err_msg=tag)


def test_sampled_gradient_rows_equal_the_full_float64_backward():
    """oracle/sampled_rows.py (one row of dq / dk / dv at a time, what the full-size GPU tests use) against the whole
    float64 backward on packed segments with a 5:1 head ratio: to 1e-12."""
    from oracle import sampled_rows as R
    cu = [0, 37, 38, 200, 333]
    T, Hq, Hkv, D = cu[-1], 10, 2, 32
    g = torch.Generator().manual_seed(3)
    q, k, v, do = (torch.randn(T, H, D, generator=g).to(torch.bfloat16) for H in (Hq, Hkv, Hkv, Hq))
    scale = A.default_scale(D)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, True)
    ro, rl = A.varlen_attention(q, k, v, cu, scale, True)
    rep = Hq // Hkv
    for a, b in zip(cu[:-1], cu[1:]):
        for i in {a, (a + b) // 2, b - 1}:
            for h in (0, 7):
                dq, l, o = R.dq_row(q, k, v, do, i, h, (a, b), scale)
                assert np.abs(dq.numpy() - rdq[i, h]).max() < 1e-12 and abs(float(l) - rl[h, i]) < 1e-12
                assert np.abs(o.numpy() - ro[i, h]).max() < 1e-12
                dq2, _, _ = R.dq_row(q, k, v, do, i, h, (a, b), scale, out_saved=torch.from_numpy(ro))   # the exact output "saved"
                assert np.abs(dq2.numpy() - rdq[i, h]).max() < 1e-12
        for gkv in range(Hkv):
            r0 = a + (b - a) // 3                                  # statistics from a row inside the segment onwards
            lse, delta = R.row_stats(q, k, v, do, list(range(gkv * rep, gkv * rep + rep)), (a, b), (r0, b), scale, chunk=50)
            assert np.abs(lse.numpy() - rl[gkv * rep:gkv * rep + rep, r0:b]).max() < 1e-12
            sv = R.saved_delta(torch.from_numpy(ro), do, list(range(gkv * rep, gkv * rep + rep)), (r0, b))
            assert float((sv - delta).abs().max()) < 1e-12
            for j in {r0, (r0 + b) // 2, b - 1}:
                dk, dv = R.dkdv_row(q, k, v, do, j, gkv, (a, b), lse, delta, r0, scale)
                assert np.abs(dk.numpy() - rdk[j, gkv]).max() < 1e-12 and np.abs(dv.numpy() - rdv[j, gkv]).max() < 1e-12
    assert R.row_excess(torch.tensor([1.0, 0.0]), torch.tensor([1.0, 0.0])) == 0.0
    assert R.row_excess(torch.tensor([1.03, 0.0]), torch.tensor([1.0, 0.0])) > 1.0


def _bf16_from_bits(bits):
    return torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16)


def test_band_masked_attention_oracle_against_reference_wrapper(golden_dir):
    """The float64 oracle with a band (a query sees itself and the `band` keys before it, inside its segment) against
    outputs and autograd gradients of the reference's eager wrapper under the same masks (eager_attention_band.npz)."""
    g = np.load(os.path.join(golden_dir, "eager_attention_band.npz"))
    for name in ("band_causal", "band_packed", "band_d64", "band_self_only", "band_wider_than_rows"):
        sq, np_, ng, hn, band = (int(x) for x in g[f"{name}.meta"])
        cu = [int(x) for x in g[f"{name}.cu"]]
        q, k, v = (_bf16_from_bits(g[f"{name}.{x}"])[:, 0] for x in "qkv")
        do = _bf16_from_bits(g[f"{name}.dout"])[:, 0].reshape(sq, np_, hn)
        scale = A.default_scale(hn)
        out, _ = A.varlen_attention(q, k, v, cu, scale, True, band=band)
        dq, dk, dv = A.varlen_attention_grad(q, k, v, do, cu, scale, True, band=band)
        ref_out = _bf16_from_bits(g[f"{name}.out"]).double().numpy().reshape(sq, np_, hn)
        assert np.abs(out - ref_out).max() <= 8e-3 * np.abs(ref_out).max(), name       # the fixture's out is stored in bf16
        for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
            ref = g[f"{name}.{key}"].astype(np.float64)[:, 0]                            # stored in fp16
            assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-6), (name, key)
