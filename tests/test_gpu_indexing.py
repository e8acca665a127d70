"""Bit-exact GPU tests of the data-movement kernels (pixel shuffle, feature scatter,
zig-zag row gather) and of the core-attention dispatch mirror, against the oracle and
the fixtures produced by the reference's own code."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention_ref as A
from oracle import indexing_ref as I


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy()


def test_pixel_shuffle_matches_reference_fixture(golden_dir):
    from long_vita_amd import vision
    g = np.load(os.path.join(golden_dir, "indexing.npz"))
    # the fixture's values (0..2047) are not all bf16-exact: send value % 256 and value // 256 separately
    xin = g["ps.small.in"].astype(np.int64)
    lo_s = vision.pixel_shuffle(torch.from_numpy((xin % 256).astype(np.float32)).to(torch.bfloat16).cuda(), 0.5)
    hi_s = vision.pixel_shuffle(torch.from_numpy((xin // 256).astype(np.float32)).to(torch.bfloat16).cuda(), 0.5)
    np.testing.assert_array_equal((lo_s.float() + 256 * hi_s.float()).cpu().numpy(), g["ps.small.out"])
    # full-size permutation [1,32,32,1024] -> [1,16,16,4096]: encode the flat source index in 2 bf16-exact planes
    idx = np.arange(32 * 32 * 1024, dtype=np.int64).reshape(1, 32, 32, 1024)
    lo = torch.from_numpy((idx % 256).astype(np.float32)).to(torch.bfloat16).cuda()
    hi = torch.from_numpy((idx // 256 % 256).astype(np.float32)).to(torch.bfloat16).cuda()
    top = torch.from_numpy((idx // 65536).astype(np.float32)).to(torch.bfloat16).cuda()
    rec = (vision.pixel_shuffle(lo).float() + 256 * vision.pixel_shuffle(hi).float()
           + 65536 * vision.pixel_shuffle(top).float()).cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(rec, g["ps.full.index_map"].astype(np.int64))


def _planes(values: np.ndarray):
    """Integer values < 2^24 as three bf16-exact planes (value % 256, value // 256 % 256, value // 65536)."""
    v = values.astype(np.int64)
    return [torch.from_numpy(p.astype(np.float32)).to(torch.bfloat16).cuda() for p in (v % 256, v // 256 % 256, v // 65536)]


def test_pixel_shuffle_v2_matches_reference_fixture(golden_dir):
    """The stride form of the token compression (pretrain_lcvlm.py:472-476, 558-570) on the HIP kernel, bit for bit
    against outputs of the reference's own method: even and odd (zero-padded) grids, strides 2 / 3 / 4, and the full
    32 x 32 x 1024 grid as an index map; then the fused class-token drop and the gradient (the inverse movement)."""
    from long_vita_amd import ops, vision
    g = np.load(os.path.join(golden_dir, "pixel_shuffle_v2.npz"))
    for tag in ("even", "odd", "s4", "s3"):
        st = int(g[f"{tag}.stride"])
        lo, hi, top = (vision.pixel_shuffle_v2(p, st).float() for p in _planes(g[f"{tag}.in"]))
        np.testing.assert_array_equal((lo + 256 * hi + 65536 * top).cpu().numpy(), g[f"{tag}.out"], err_msg=tag)
    idx = np.arange(1, 1 + 32 * 32 * 1024, dtype=np.int64).reshape(1, 32, 32, 1024)
    lo, hi, top = (vision.pixel_shuffle_v2(p, 2).float() for p in _planes(idx))
    np.testing.assert_array_equal((lo + 256 * hi + 65536 * top).cpu().numpy().astype(np.int64), g["full.index_map"].astype(np.int64))
    # class-token drop fused in, odd grid, against the oracle on the bit patterns
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(3, 1 + 7 * 7, 64, generator=gen).to(torch.bfloat16)
    ref = I.pixel_shuffle_v2(x[:, 1:].reshape(3, 7, 7, 64).view(torch.int16).numpy(), 2)
    got = vision.vit_output_to_llm_tokens(x.cuda(), add_class_token=True, scale_stride=2)
    assert got.shape == (3, 16, 256)
    np.testing.assert_array_equal(bf16_bits(got), ref)
    # gradient: every token's gradient comes back from the one place it went to, the class token and the padding get none
    xg = x.cuda().requires_grad_(True)
    w = torch.randn(3, 16, 256, generator=gen).to(torch.bfloat16).cuda()
    (ops.pixel_shuffle_v2_tokens(xg, 1, 2).float() * w.float()).sum().backward()
    xr = x.float().requires_grad_(True)
    grid = torch.nn.functional.pad(xr[:, 1:].reshape(3, 7, 7, 64), (0, 0, 0, 1, 0, 1))
    yr = grid.reshape(3, 4, 2, 4, 2, 64).permute(0, 1, 3, 2, 4, 5).flatten(3).reshape(3, -1, 256)
    (yr * w.float().cpu()).sum().backward()
    assert torch.equal(xg.grad.float().cpu(), xr.grad)


def test_vit_output_to_llm_tokens_bit_exact():
    from long_vita_amd import vision
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 1025, 1024, generator=g).to(torch.bfloat16)
    ref = I.vit_tokens_to_llm_tokens(x.view(torch.int16).numpy())       # pure movement: compare bit patterns
    got = vision.vit_output_to_llm_tokens(x.cuda())
    assert got.shape == (5, 256, 4096)
    np.testing.assert_array_equal(bf16_bits(got), ref)


def test_scatter_features_bit_exact(golden_dir):
    from long_vita_amd import vision
    g = torch.Generator().manual_seed(1)
    b, s, h, n_img, n_tok = 2, 600, 256, 3, 64
    emb = torch.randn(b, s, h, generator=g).to(torch.bfloat16)
    feats = torch.randn(n_img, n_tok, h, generator=g).to(torch.bfloat16)
    starts = [5, 300, 100]
    ind_s = torch.stack([torch.arange(st, st + n_tok) for st in starts])
    ind_b = torch.tensor([0, 0, 1])[:, None].repeat(1, n_tok)
    indices = torch.stack([ind_b, ind_s])
    ref = I.scatter_features(emb.view(torch.int16).numpy(), feats.view(torch.int16).numpy(), indices=indices.numpy())
    got = vision.scatter_external_features(emb.clone().cuda(), {"features": feats.cuda(), "indices": indices.cuda()})
    np.testing.assert_array_equal(bf16_bits(got), ref)
    # src / tgt form, as produced by the CP partition
    sel, src, tgt = I.cp_image_remap(indices.numpy(), s + 40, 4, 1)   # seq 640, CP 4
    emb2 = torch.randn(b, (s + 40) // 4, h, generator=g).to(torch.bfloat16)
    fsel = feats[torch.from_numpy(sel)]
    ref2 = I.scatter_features(emb2.view(torch.int16).numpy(), fsel.view(torch.int16).numpy(), src_indices=src, tgt_indices=tgt)
    got2 = vision.scatter_external_features(emb2.clone().cuda(), {"features": fsel.cuda(), "src_indices": torch.from_numpy(src).cuda(),
                                                                  "tgt_indices": torch.from_numpy(tgt).cuda()})
    np.testing.assert_array_equal(bf16_bits(got2), ref2)
    # no indices: nothing moves
    got3 = vision.scatter_external_features(emb.clone().cuda(), {"features": feats.cuda()})
    np.testing.assert_array_equal(bf16_bits(got3), emb.view(torch.int16).numpy())


@pytest.mark.parametrize("S,cp", [(64, 2), (256, 4), (1024, 8)])
def test_cp_image_index_remap_on_device_matches_reference(golden_dir, S, cp):
    """The image-index remap of the CP partition on DEVICE tensors, where the reference builds it
    (lcvlm_modellink/training/utils.py:271-281 selects the images a rank touches, :314-333 builds the (image, slot)
    source and (row, local position) target indices with index_of_a_in_b :539-542): selected images, src / tgt indices,
    tokens and position ids bit-exact against the fixture made by the reference's own function, for every rank; the
    device-built indices then drive the HIP scatter, against the oracle's scatter on the fixture's indices."""
    from long_vita_amd import cp_partition as P, state, vision
    g = np.load(os.path.join(golden_dir, "indexing.npz"))
    tag = f"zz.S{S}.cp{cp}"
    ext = torch.from_numpy(g[f"{tag}.external_indices"])
    n_img, n_tok, h = ext.shape[1], ext.shape[2], 64
    gen = torch.Generator().manual_seed(S + cp)
    feats_all = torch.randn(n_img, n_tok, h, generator=gen).to(torch.bfloat16)
    dev = torch.device("cuda", 0)
    for r in range(cp):
        state.set_args(state.make_args(seq_length=S, context_parallel_size=cp))
        state.mpu.initialize(None, cp, r)

        def batch(images):
            return {"tokens": torch.from_numpy(g[f"{tag}.tokens"]).to(dev), "position_ids": torch.arange(S).repeat(2, 1).to(dev),
                    "attention_mask": None, "external_images": images.to(dev), "external_indices": ext.clone().to(dev)}
        out = P.get_batch_on_this_cp_rank(batch(torch.from_numpy(g[f"{tag}.images"])))
        assert all(v.is_cuda for k, v in out.items() if v is not None)
        np.testing.assert_array_equal(out["tokens"].cpu().numpy(), g[f"{tag}.r{r}.tokens"])
        np.testing.assert_array_equal(out["position_ids"].cpu().numpy(), g[f"{tag}.r{r}.position_ids"])
        np.testing.assert_array_equal(out["external_images"].cpu().numpy(), g[f"{tag}.r{r}.images"])
        assert "external_indices" not in out
        if f"{tag}.r{r}.src" not in g.files:
            assert "external_src_indices" not in out
            continue
        src, tgt = out["external_src_indices"], out["external_tgt_indices"]
        assert src.dtype == torch.int64 and tgt.dtype == torch.int64
        np.testing.assert_array_equal(src.cpu().numpy(), g[f"{tag}.r{r}.src"])
        np.testing.assert_array_equal(tgt.cpu().numpy(), g[f"{tag}.r{r}.tgt"])
        # the same partition carrying per-image features instead of pixels: what the encoder of this rank would return
        fsel = P.get_batch_on_this_cp_rank(batch(feats_all))["external_images"]
        assert fsel.shape[0] == g[f"{tag}.r{r}.images"].shape[0]
        emb = torch.randn(2, S // cp, h, generator=gen).to(torch.bfloat16)
        ref = I.scatter_features(emb.view(torch.int16).numpy(), fsel.cpu().view(torch.int16).numpy(),
                                 src_indices=g[f"{tag}.r{r}.src"], tgt_indices=g[f"{tag}.r{r}.tgt"])
        got = vision.scatter_external_features(emb.clone().to(dev), {"features": fsel, "src_indices": src, "tgt_indices": tgt})
        np.testing.assert_array_equal(bf16_bits(got), ref)


def test_index_of_a_in_b_on_device(golden_dir):
    from long_vita_amd.cp_partition import index_of_a_in_b
    g = np.load(os.path.join(golden_dir, "indexing.npz"))
    got = index_of_a_in_b(torch.from_numpy(g["ioab.a"]).cuda(), torch.from_numpy(g["ioab.b"]).cuda())
    assert got.is_cuda
    np.testing.assert_array_equal(got.cpu().numpy(), g["ioab.out"])
    # a 1M-token rank's worth: positions of 4096 frames x 256 slots inside a CP = 8 rank's zig-zag index
    S, cp, r = 1 << 20, 8, 5
    calib = P_zigzag(S, cp, r)
    pick = calib[torch.randperm(calib.numel(), generator=torch.Generator().manual_seed(0))[:65536]]
    got = index_of_a_in_b(pick.cuda(), calib.cuda())
    assert torch.equal(calib.cuda()[got], pick.cuda())


def P_zigzag(S, cp, r):
    from long_vita_amd import cp_partition as P
    return P.zigzag_index(S, cp, r)


def test_zigzag_gather_on_gpu_bit_exact():
    from long_vita_amd import cp_partition as P
    table = torch.arange(4096 * 128, dtype=torch.float32).reshape(4096, 1, 1, 128)
    for cp in (2, 8):
        for r in (0, cp - 1):
            got = P.get_pos_emb_on_this_cp_rank(table.cuda(), 0, cp, r)
            np.testing.assert_array_equal(got.cpu().numpy(), I.zigzag_slice(table.numpy(), cp, r, 0))
    pos = torch.randint(0, 4096, (300, 2))
    got = P.gather_rotary_by_position_ids(table.cuda(), pos.cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), I.rope_position_gather(table.numpy(), pos.numpy()))
    odd = torch.arange(37 * 3, dtype=torch.int32).reshape(37, 3).cuda()     # 12-byte rows: 4-byte path
    idx = torch.tensor([5, 0, 36, 5]).cuda()
    from long_vita_amd import ops
    assert torch.equal(ops.gather_rows(odd, idx).cpu(), odd.cpu()[idx.cpu()])


def _self(d):
    return types.SimpleNamespace(hidden_size_per_attention_head=d, scale_mask_softmax=types.SimpleNamespace(scale=None),
                                 softmax_scale=None, attention_dropout=types.SimpleNamespace(p=0.0), alibi=None,
                                 pse=None, pse_type=None)


def _rel(got, ref):
    g = got.detach().float().cpu().numpy().astype(np.float64)
    return float(np.abs(g - ref).max() / np.abs(ref).max())


def test_flash_attention_forward_branches_match_oracle():
    """cfg1-shaped plumbing: 8192 packed tokens with the first segment one 448px image (258
    tokens), Qwen2 head ratio, through the same entry point the reference patches in."""
    from long_vita_amd import state
    from long_vita_amd.core_attention import flash_attention_forward
    g = torch.Generator().manual_seed(2)
    s, b, hq, hkv, d = 8192, 1, 10, 2, 128
    actual = [258, 1290, 3000, 5120, 8192]
    q = torch.randn(s, b, hq, d, generator=g).to(torch.bfloat16)
    k = torch.randn(s, b, hkv, d, generator=g).to(torch.bfloat16)
    v = torch.randn(s, b, hkv, d, generator=g).to(torch.bfloat16)
    # LLM varlen branch
    state.set_args(state.make_args(reset_attention_mask=True, reset_position_ids=True))
    state.set_actual_seq_len(actual)
    out = flash_attention_forward(_self(d), q.cuda(), k.cuda(), v.cuda(), None, None, None)
    assert out.shape == (s, b, hq * d)
    ref, _ = A.varlen_attention(q[:, 0, :4], k[:, 0, :1], v[:, 0, :1], [0] + actual, A.default_scale(d))  # heads 0-3 of group 0
    assert _rel(out.reshape(s, hq, d)[:, :4], ref) <= 2e-2
    # generic causal branch (no reset): one causal block over the whole row
    state.set_args(state.make_args())
    state.set_actual_seq_len(None)
    out2 = flash_attention_forward(_self(d), q[:2048].cuda(), k[:2048].cuda(), v[:2048].cuda(), None, None, None)
    ref2, _ = A.varlen_attention(q[:2048, 0, :2], k[:2048, 0, :1], v[:2048, 0, :1], [0, 2048], A.default_scale(d))
    assert _rel(out2.reshape(2048, hq, d)[:, :2], ref2) <= 2e-2
    # ViT branch: chosen by the sequence length, bidirectional, b images
    sv, bv, hv, dv = 1025, 3, 4, 64
    qv, kv, vv = (torch.randn(sv, bv, hv, dv, generator=g).to(torch.bfloat16) for _ in range(3))
    out3 = flash_attention_forward(_self(dv), qv.cuda(), kv.cuda(), vv.cuda(), None, None, None)
    assert out3.shape == (sv, bv, hv * dv)
    for bi in range(bv):
        ref3, _ = A.varlen_attention(qv[:, bi], kv[:, bi], vv[:, bi], [0, sv], A.default_scale(dv), causal=False)
        assert _rel(out3[:, bi].reshape(sv, hv, dv), ref3) <= 2e-2


def test_flash_attention_forward_micro_batch_two_flattening_orders():
    """Micro-batch 2 through the two packed branches, which flatten differently: the mask-reset branch goes
    's b h d -> (b s) h d' (dot_product_attention.py:345), batch-major, matching the collate's per-row offsets of
    actual_seq_len (data_samplers.py:46-48); the generic TND branch goes '(s b)' (:382), sequence-major.  Each output is
    compared with the oracle run on the flattened layout the branch defines."""
    from long_vita_amd import state
    from long_vita_amd.core_attention import flash_attention_forward
    g = torch.Generator().manual_seed(12)
    s, b, hq, hkv, d = 300, 2, 4, 2, 128
    q = torch.randn(s, b, hq, d, generator=g).to(torch.bfloat16)
    k = torch.randn(s, b, hkv, d, generator=g).to(torch.bfloat16)
    v = torch.randn(s, b, hkv, d, generator=g).to(torch.bfloat16)
    # mask-reset branch: row 0 holds segments ending at 100 and 300, row 1 at 50 and 300 (+ 300 from the collate)
    actual = [100, 300, 350, 600]
    state.set_args(state.make_args(reset_attention_mask=True, reset_position_ids=True))
    state.set_actual_seq_len(actual)
    out = flash_attention_forward(_self(d), q.cuda(), k.cuda(), v.cuda(), None, None, None)
    assert out.shape == (s, b, hq * d)
    for row, ends in ((0, [100, 300]), (1, [50, 300])):
        ref, _ = A.varlen_attention(q[:, row], k[:, row], v[:, row], [0] + ends, A.default_scale(d))
        assert _rel(out[:, row].reshape(s, hq, d), ref) <= 2e-2, row
    # generic TND branch: tokens interleave as (s b); one causal segment over the 600 flattened positions
    state.set_args(state.make_args(shape_order="TND"))
    state.set_actual_seq_len([s * b])
    out2 = flash_attention_forward(_self(d), q.cuda(), k.cuda(), v.cuda(), None, None, None)
    flat = lambda x: x.reshape(s * b, x.shape[2], d)
    ref2, _ = A.varlen_attention(flat(q), flat(k), flat(v), [0, s * b], A.default_scale(d))
    assert _rel(out2.reshape(s * b, hq, d), ref2) <= 2e-2
    state.set_args(state.make_args())
    state.set_actual_seq_len(None)


def test_fusion_attention_layouts_agree():
    from long_vita_amd.fusion_attention import npu_fusion_attention
    g = torch.Generator().manual_seed(3)
    b, s, n, nkv, d = 2, 300, 4, 2, 128
    q = torch.randn(b, s, n, d, generator=g).to(torch.bfloat16).cuda()
    k = torch.randn(b, s, nkv, d, generator=g).to(torch.bfloat16).cuda()
    v = torch.randn(b, s, nkv, d, generator=g).to(torch.bfloat16).cuda()
    sc = A.default_scale(d)
    o_bsnd = npu_fusion_attention(q, k, v, n, "BSND", atten_mask=True, scale=sc, sparse_mode=2)[0]
    o_bnsd = npu_fusion_attention(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), n, "BNSD",
                                  atten_mask=True, scale=sc, sparse_mode=2)[0]
    o_bsh = npu_fusion_attention(q.reshape(b, s, -1), k.reshape(b, s, -1), v.reshape(b, s, -1), n, "BSH",
                                 atten_mask=True, scale=sc, sparse_mode=2)[0]
    o_sbh = npu_fusion_attention(q.permute(1, 0, 2, 3).reshape(s, b, -1), k.permute(1, 0, 2, 3).reshape(s, b, -1),
                                 v.permute(1, 0, 2, 3).reshape(s, b, -1), n, "SBH", atten_mask=True, scale=sc, sparse_mode=2)[0]
    assert torch.equal(o_bsnd, o_bnsd.permute(0, 2, 1, 3))
    assert torch.equal(o_bsnd.reshape(b, s, -1), o_bsh)
    assert torch.equal(o_bsnd.reshape(b, s, -1), o_sbh.permute(1, 0, 2))
    ref, _ = A.varlen_attention(q[1].cpu(), k[1].cpu(), v[1].cpu(), [0, s], sc)
    assert _rel(o_bsnd[1], ref) <= 2e-2
    with pytest.raises(NotImplementedError):
        npu_fusion_attention(q, k, v, n, "BSND", keep_prob=0.9)


def test_logit_mask_selection_matches_masked_select():
    """Row selection of the --logit-mask path against torch.masked_select, the form the reference writes
    (gpt_vl_model.py:312, :357); bit-exact."""
    from long_vita_amd import lm_head
    g = torch.Generator().manual_seed(5)
    s, c, vocab = 1000, 256, 512
    hidden = torch.randn(s, 1, c, generator=g).to(torch.bfloat16)
    labels = torch.randint(0, vocab, (1, s), generator=g)
    mask = torch.rand(1, s, generator=g) < 0.07
    want_h = torch.masked_select(hidden, mask.transpose(0, 1).unsqueeze(2)).reshape(-1, 1, c)
    want_l = torch.masked_select(labels, mask).reshape(1, -1)
    got_h = lm_head.select_hidden_states(hidden.cuda(), mask.cuda())
    got_l = lm_head.select_labels(labels.cuda(), mask.cuda())
    assert torch.equal(got_h.cpu(), want_h) and torch.equal(got_l.cpu(), want_l)
    w = torch.randn(vocab, c, generator=g).to(torch.bfloat16)
    logits = lm_head.masked_logits(hidden.cuda(), w.cuda(), mask.cuda())
    ref = want_h.float() @ w.float().t()
    assert logits.shape == (int(mask.sum()), 1, vocab)
    assert (logits.float().cpu() - ref).abs().max() <= 2e-2 * ref.abs().max()
    # generation: one position per row, taken modulo the local length under context parallelism
    toks = torch.zeros(1, 64, dtype=torch.int64)
    assert lm_head.logit_mask_for_context(toks, 40, 1).nonzero().tolist() == [[0, 39]]
    assert lm_head.logit_mask_for_context(toks, 64 * 3 + 5, 4).nonzero().tolist() == [[0, 4]]
    assert lm_head.logit_mask_for_context(toks, 128, 2).nonzero().tolist() == [[0, 63]]   # index -1, as the reference


def _from_bits(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16)


def test_embedding_scatter_with_gradients_matches_reference_fixture(golden_dir):
    """LanguageModelEmbedding.forward in its four external_feature_dict forms, output AND the gradients that reach
    the word embeddings and the visual features, against the fixture the reference's own forward + torch autograd
    produced (language_model_embedding.py:91-174); all data movement, so bit for bit."""
    from long_vita_amd import vision
    g = np.load(os.path.join(golden_dir, "embedding_scatter.npz"))
    b = int(g["meta"][0])
    for tag in ("indices", "pre_len", "src_tgt", "none"):
        we = _from_bits(g[f"{tag}.we"]).cuda().requires_grad_(True)
        feats = _from_bits(g[f"{tag}.features"]).cuda().requires_grad_(True)
        do = _from_bits(g[f"{tag}.dout"]).cuda()
        d = {"features": feats}
        if tag == "indices":
            d["indices"] = torch.from_numpy(g["indices.indices"]).cuda()
        elif tag == "pre_len":
            d = {"features": feats[:b], "pre_len": int(g["pre_len.pre_len"])}
        elif tag == "src_tgt":
            d["src_indices"] = tuple(torch.from_numpy(x).cuda() for x in g["src_tgt.src_indices"])
            d["tgt_indices"] = tuple(torch.from_numpy(x).cuda() for x in g["src_tgt.tgt_indices"])
        self = types.SimpleNamespace(word_embeddings=lambda ids: we.clone(), add_position_embedding=False,
                                     tokentype_embeddings=None, embedding_dropout=lambda x: x,
                                     config=types.SimpleNamespace(fp32_residual_connection=False, sequence_parallel=False))
        emb = vision.language_model_embedding_forward(self, torch.zeros(b, we.shape[1], dtype=torch.long), None, None, d)
        emb.backward(do)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(bf16_bits(emb).view(np.uint16), g[f"{tag}.out"], err_msg=tag)
        np.testing.assert_array_equal(bf16_bits(we.grad).view(np.uint16), g[f"{tag}.dwe"], err_msg=tag)
        assert feats.grad is not None, tag          # the producer of the features stays in the graph in every form
        np.testing.assert_array_equal(bf16_bits(feats.grad).view(np.uint16), g[f"{tag}.dfeatures"], err_msg=tag)


def test_scatter_refuses_to_sever_a_graph():
    """The raw launch has no gradient: it must not be usable on tensors that need one."""
    from long_vita_amd import ops, _lib
    dst = torch.zeros(8, 16, dtype=torch.bfloat16, device="cuda")
    src = torch.ones(2, 16, dtype=torch.bfloat16, device="cuda", requires_grad=True)
    with pytest.raises(_lib.LvAttnError):
        ops.scatter_rows_(dst, torch.tensor([1, 2], device="cuda"), src)
    with torch.no_grad():
        ops.scatter_rows_(dst, torch.tensor([1, 2], device="cuda"), src)
    assert dst[1].float().sum().item() == 16.0


def test_gather_rows_and_masked_logits_are_differentiable():
    """Row selection must hand gradients back to the hidden states (zero for unselected positions, summed for a
    row picked twice), as torch.masked_select / index_select do."""
    from long_vita_amd import lm_head, ops, cp_partition as P
    g = torch.Generator().manual_seed(9)
    s, c, vocab = 300, 64, 128
    hidden = torch.randn(s, 1, c, generator=g).to(torch.bfloat16)
    w = torch.randn(vocab, c, generator=g).to(torch.bfloat16)
    mask = torch.rand(1, s, generator=g) < 0.1
    dlog = torch.randn(int(mask.sum()), 1, vocab, generator=g).to(torch.bfloat16)
    hc, wc = hidden.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    lm_head.masked_logits(hc, wc, mask.cuda()).backward(dlog.cuda())
    hr, wr = hidden.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    sel = torch.masked_select(hr, mask.cuda().transpose(0, 1).unsqueeze(2)).reshape(-1, 1, c)
    torch.nn.functional.linear(sel, wr).backward(dlog.cuda())
    assert hc.grad is not None and torch.equal(hc.grad, hr.grad) and torch.equal(wc.grad, wr.grad)
    # repeated rows: gradients add up
    x = torch.randn(10, 8, generator=g).cuda().requires_grad_(True)
    idx = torch.tensor([3, 3, 0, 9, 3], device="cuda")
    ops.gather_rows(x, idx).backward(torch.ones(5, 8, device="cuda"))
    want = torch.zeros(10, 8); want[3] = 3; want[0] = 1; want[9] = 1
    assert torch.equal(x.grad.cpu(), want)
    # the zig-zag slice of an activation
    act = torch.randn(64, 2, 32, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    P.zigzag_slice(act, 4, 1, 0).backward(torch.ones(16, 2, 32, dtype=torch.bfloat16, device="cuda"))
    want = torch.zeros(64, 1, 1); want[8:16] = 1; want[48:56] = 1
    assert torch.equal(act.grad.float().cpu(), want.expand(64, 2, 32))


def test_operands_of_the_elementwise_kernels_are_validated():
    """The per-column operands travel as raw pointers: a wrong dtype, device or length must raise, not be misread."""
    from long_vita_amd import ops, _lib
    x = torch.randn(4, 128, device="cuda").to(torch.bfloat16)
    good = torch.ones(128, dtype=torch.bfloat16, device="cuda")
    for bad in (torch.ones(128, device="cuda"), torch.ones(64, dtype=torch.bfloat16, device="cuda"),
                torch.ones(128, dtype=torch.bfloat16)):
        with pytest.raises(_lib.LvAttnError):
            ops.layernorm_launch(x, bad, good, 1e-5)
        with pytest.raises(_lib.LvAttnError):
            ops.layernorm_launch(x, good, bad, 1e-5)
        with pytest.raises(_lib.LvAttnError):
            ops.bias_gelu_launch(x, bad)
        with pytest.raises(_lib.LvAttnError):
            ops.bias_scale_residual_launch(x, good, bad, x)
        with pytest.raises(_lib.LvAttnError):
            ops.layernorm_bwd_launch(x, x, bad, 1e-5)
    with pytest.raises(_lib.LvAttnError):
        ops.bias_scale_residual_launch(x, good, good, x[:2])
    ops.layernorm_launch(x, good, good, 1e-5)          # the well-formed call still goes through


def test_tensor_of_another_device_is_refused(monkeypatch):
    """The launches go to the current device's current stream: a tensor that lives on another GPU must raise instead of
    being addressed from the wrong device (one-GPU box: the current device index is faked)."""
    from long_vita_amd import ops, _lib
    q = torch.zeros(64, 2, 128, dtype=torch.bfloat16, device="cuda")
    desc = ops.blocks_on_device([0, 64], True, False, "cuda")
    real = torch.cuda.current_device()
    monkeypatch.setattr(torch.cuda, "current_device", lambda: real + 1)
    with pytest.raises(_lib.LvAttnError, match="current device"):
        ops.attn_fwd_launch(q, q, q, desc, 0.1, True)
    with pytest.raises(_lib.LvAttnError, match="current device"):
        ops.pixel_shuffle_tokens_launch(torch.zeros(1, 5, 8, dtype=torch.bfloat16, device="cuda"))
