"""Host-side mirrors of the reference's partition / packing code, bit-exact against
fixtures produced by the reference's own functions.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import indexing_ref as I


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _set_state(S, cp, r):
    from long_vita_amd import state
    state.set_args(state.make_args(seq_length=S, context_parallel_size=cp))
    state.mpu.initialize(None, cp, r)


def test_index_of_a_in_b(golden_dir):
    from long_vita_amd.cp_partition import index_of_a_in_b
    g = _g(golden_dir, "indexing.npz")
    got = index_of_a_in_b(torch.from_numpy(g["ioab.a"]), torch.from_numpy(g["ioab.b"]))
    np.testing.assert_array_equal(got.numpy(), g["ioab.out"])


@pytest.mark.parametrize("S,cp", [(64, 2), (256, 4), (1024, 8)])
def test_cp_batch_partition_matches_reference(golden_dir, S, cp):
    from long_vita_amd import cp_partition as P
    g = _g(golden_dir, "indexing.npz")
    tag = f"zz.S{S}.cp{cp}"
    gathered = []
    for r in range(cp):
        _set_state(S, cp, r)
        batch = {"tokens": torch.from_numpy(g[f"{tag}.tokens"]).clone(),
                 "position_ids": torch.arange(S).repeat(2, 1), "attention_mask": None,
                 "external_images": torch.from_numpy(g[f"{tag}.images"]).clone(),
                 "external_indices": torch.from_numpy(g[f"{tag}.external_indices"]).clone()}
        out = P.get_batch_on_this_cp_rank(batch)
        np.testing.assert_array_equal(out["tokens"].numpy(), g[f"{tag}.r{r}.tokens"])
        np.testing.assert_array_equal(out["position_ids"].numpy(), g[f"{tag}.r{r}.position_ids"])
        np.testing.assert_array_equal(out["external_images"].numpy(), g[f"{tag}.r{r}.images"])
        assert "external_indices" not in out
        if f"{tag}.r{r}.src" in g.files:
            np.testing.assert_array_equal(out["external_src_indices"].numpy(), g[f"{tag}.r{r}.src"])
            np.testing.assert_array_equal(out["external_tgt_indices"].numpy(), g[f"{tag}.r{r}.tgt"])
        else:
            assert "external_src_indices" not in out
        gathered.append(out["tokens"])
    # sync_output's reordering restores the global order
    np.testing.assert_array_equal(P.unzigzag_gathered(gathered, cp, 1).numpy(), g[f"{tag}.tokens"])


def test_rope_slice_and_gather(golden_dir):
    from long_vita_amd import cp_partition as P
    g = _g(golden_dir, "indexing.npz")
    emb = torch.from_numpy(g["rope.in"])
    for cp in (2, 4):
        for r in range(cp):
            got = P.get_pos_emb_on_this_cp_rank(emb, 0, cp, r)
            np.testing.assert_array_equal(got.numpy(), g[f"rope.cp{cp}.r{r}"])
    pos = torch.tensor([[0, 0], [1, 1], [2, 0], [0, 1]])
    got = P.gather_rotary_by_position_ids(emb, pos)
    np.testing.assert_array_equal(got.numpy(), I.rope_position_gather(g["rope.in"], pos.numpy()))


def test_packing_matches_reference(golden_dir):
    from long_vita_amd.packing import PackedSample
    g = _g(golden_dir, "packing.npz")
    img_a = [[2, 3, 4, 5], [8, 9, 10, 11]]
    img_b = [[1, 2, 3, 4]]
    cases = {"pad": ([(15, img_a), (9, None), (6, img_b)], 40, True, False),
             "trunc": ([(15, None), (9, img_b), (20, None)], 32, True, False),
             "late_image": ([(10, None), (12, img_b), (8, img_a)], 32, True, False),
             "shift": ([(15, img_a), (9, None)], 32, True, True),
             "noreset": ([(15, img_a), (9, None)], 32, False, False)}
    for name, (samples, max_len, reset, shift) in cases.items():
        ps = PackedSample(max_len, 0, reset, reset, shift_token=shift)
        for i, (L, img) in enumerate(samples):
            ii = None
            if img is not None:
                pos = torch.tensor(img, dtype=torch.int64)
                ii = torch.stack([torch.zeros_like(pos), pos])
            ps.add(list(range(1000 * i, 1000 * i + L)), list(range(L)), ii)
        out = ps.finalize()
        for k in ("tokens", "labels", "position_ids", "actual_seq_len", "loss_mask", "image_indices"):
            key = f"{name}.{k}"
            if key in g.files:
                np.testing.assert_array_equal(out[k].numpy(), g[key], err_msg=key)
            else:
                assert k not in out, key


def test_collate_matches_reference(golden_dir):
    from long_vita_amd.packing import collate_fn
    g = _g(golden_dir, "indexing.npz")
    rows = []
    for i in range(3):
        d = {"tokens": torch.arange(32) + 100 * i, "actual_seq_len": torch.from_numpy(g[f"collate.in_actual{i}"])}
        if f"collate.in_img{i}" in g.files:
            img = torch.from_numpy(g[f"collate.in_img{i}"]).clone()
            d["image_indices"] = img
            d["images"] = torch.full((img.shape[1], 3), float(i))
        rows.append(d)
    out = collate_fn(rows)
    np.testing.assert_array_equal(out["actual_seq_len"].numpy(), g["collate.actual_seq_len"])
    np.testing.assert_array_equal(out["image_indices"].numpy(), g["collate.image_indices"])
    np.testing.assert_array_equal(out["images"].numpy(), g["collate.images"])
    assert out["tokens"].shape == (3, 32)


def test_frame_token_span_layout():
    from long_vita_amd.packing import frame_token_span
    ii, n = frame_token_span(3, 4)
    pos, n2 = I.image_token_layout(3, 1, 4)
    assert n == n2 == 9 and ii[1].tolist() == pos.tolist() and ii[0].sum() == 0
    # 4096 frames of 258 tokens exceed 2**20: the "1M" context holds fewer frames
    assert 4096 * 258 == 1056768 > 2 ** 20


def test_flash_attention_forward_branch_errors():
    from long_vita_amd import state
    from long_vita_amd.core_attention import flash_attention_forward
    state.set_args(state.make_args())
    q = torch.zeros(8, 1, 2, 128, dtype=torch.bfloat16)
    with pytest.raises(AssertionError, match="packed_seq_params should be None."):
        flash_attention_forward(None, q, q, q, None, None, object())
    import types
    self = types.SimpleNamespace(hidden_size_per_attention_head=128, scale_mask_softmax=types.SimpleNamespace(scale=None),
                                 softmax_scale=None, attention_dropout=types.SimpleNamespace(p=0.0), alibi=None)
    state.set_args(state.make_args(reset_attention_mask=True))
    state.set_actual_seq_len(None)
    with pytest.raises(AssertionError):
        flash_attention_forward(self, q, q, q, None, None, None)
    state.set_actual_seq_len([8])
    # packed segments + context parallelism go to the ring with the segments from get_actual_seq_len():
    # their last end must be the length of the WHOLE sequence (local rows x CP)
    state.set_args(state.make_args(reset_attention_mask=True, context_parallel_size=2))
    state.mpu.initialize(None, 2, 0)
    try:
        with pytest.raises(ValueError, match="actual_seq_len ends at 8"):
            flash_attention_forward(self, q, q, q, None, None, None)
        state.set_actual_seq_len([5, 16])
        with pytest.raises(ValueError, match="process group"):   # accepted: the ring starts (and finds no group here)
            flash_attention_forward(self, q, q, q, None, None, None)
    finally:
        state.mpu.initialize(None, 1, 0)
        state.set_actual_seq_len(None)


def _fake_target_modules(monkeypatch):
    """Empty modules at the five dotted paths the adaptor patches (Megatron / ModelLink / MindSpeed are not installed)."""
    import sys
    import types
    made = {}
    for dotted in ("modellink.core.transformer.dot_product_attention",
                   "mindspeed.core.context_parallel.ring_context_parallel",
                   "megatron.core.models.common.embeddings.rotary_pos_embedding"):
        parts = dotted.split(".")
        for i in range(1, len(parts) + 1):
            name = ".".join(parts[:i])
            if name not in made:
                m = types.ModuleType(name)
                m.__path__ = []
                made[name] = m
                monkeypatch.setitem(sys.modules, name, m)
                if i > 1:
                    setattr(made[".".join(parts[:i - 1])], parts[i - 1], m)
    return made


def test_register_patches_hits_the_dotted_names(monkeypatch):
    """The adaptor registers its five functions under the dotted names the reference patches
    (lcvlm_modellink/megatron_adaptor.py:11-13, 92-93) and a manager that applies by dotted name lands them there."""
    from long_vita_amd import host_standins, megatron_adaptor as M
    made = _fake_target_modules(monkeypatch)
    monkeypatch.setattr(host_standins.PatchesManager, "patches_info", {})
    names = M.register_patches(host_standins.PatchesManager)
    assert len(names) == 5 and set(names) == set(M.PATCHES)
    applied = host_standins.PatchesManager.apply_patches()
    assert set(applied) == set(names)
    dpa = made["modellink.core.transformer.dot_product_attention"]
    assert dpa.flash_attention_forward is M.flash_attention_forward
    assert dpa.do_ring_context_parallel is M.do_ring_context_parallel
    assert made["mindspeed.core.context_parallel.ring_context_parallel"].ringattn_context_parallel is M.ringattn_context_parallel
    rot = made["megatron.core.models.common.embeddings.rotary_pos_embedding"]
    assert rot.RotaryEmbedding is M.RotaryEmbedding and rot.apply_rotary_pos_emb is M.apply_rotary_pos_emb

    class TwoArgManager:            # a manager without the force_patch keyword is served too
        seen = []

        @classmethod
        def register_patch(cls, name, fn):
            cls.seen.append(name)
    M.register_patches(TwoArgManager)
    assert TwoArgManager.seen == list(M.PATCHES)


def test_host_standins_serve_the_reference_imports(monkeypatch):
    """`from mindspeed.utils import set_actual_seq_len ...` (training/utils.py:29) and the other Ascend-only imports of
    the path resolve to this package's state once the stand-ins are installed."""
    import sys
    from long_vita_amd import host_standins, state
    for name in [n for n in sys.modules if n == "mindspeed" or n.startswith("mindspeed.")]:
        monkeypatch.delitem(sys.modules, name)
    installed = host_standins.install()
    try:
        assert "mindspeed.utils" in installed and "mindspeed.patch_utils" in installed
        from mindspeed.utils import compute_actual_seq_len, get_actual_seq_len, set_actual_seq_len, set_position_ids  # noqa: F401
        from mindspeed.patch_utils import MindSpeedPatchesManager
        from mindspeed.core.context_parallel.ring_context_parallel import ringattn_context_parallel  # noqa: F401
        set_actual_seq_len([3, 8])
        assert state.get_actual_seq_len() == [3, 8] and get_actual_seq_len() == [3, 8]
        assert MindSpeedPatchesManager is host_standins.PatchesManager
        pos = torch.tensor([[0, 1, 2, 0, 1, 0, 1, 2]])
        assert compute_actual_seq_len(pos) == [3, 5, 8]
    finally:
        state.set_actual_seq_len(None)
        for name in installed:
            sys.modules.pop(name, None)


def test_key_pieces_rule_for_short_blocks_on_long_caches():
    """The piece count of the key-split prompt pass (ops.key_pieces_rule) on a 256-CU chip with 40 query heads: enough
    work items to fill whole rounds of the chip, pieces of at least 16,384 keys, no split when there is little to gain."""
    from long_vita_amd.ops import key_pieces_rule
    assert key_pieces_rule(256, 1 << 20, 40, 256) == 19       # 40 items: 19 pieces = 760 items = 3 rounds of 1/19 each
    assert key_pieces_rule(64, 131072, 40, 256) == 6          # 240 items: one round of a sixth of the sweep
    assert key_pieces_rule(300, 1 << 20, 40, 256) == 16       # two query blocks per head
    assert key_pieces_rule(1024, 1 << 20, 40, 256) == 8       # 160 items -> 1280 = five full rounds
    assert key_pieces_rule(1536, 1 << 20, 40, 256) == 1       # 240 items: nothing worth a fifth
    assert key_pieces_rule(4096, 1 << 20, 40, 256) == 1       # already more work items than CUs
    assert key_pieces_rule(256, 20000, 40, 256) == 1          # a short cache: no piece of 16,384 keys to cut
    for sq, sk in ((1, 1 << 20), (200, 50000), (700, 400000)):
        p = key_pieces_rule(sq, sk, 40, 256)
        assert 1 <= p <= 64 and (p == 1 or sk // p >= 16384) and p * sq <= 16384

