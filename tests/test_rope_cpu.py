"""Rotary position embedding on the CPU side: the oracle against the fixture produced by the
reference's own functions (tests/golden/rope.npz), and the host mirror's angle tables."""
import os

import numpy as np
import pytest
import torch

from oracle import rope_ref as R


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "rope.npz"))


def test_angle_tables_match_reference(golden_dir):
    g = _g(golden_dir)
    for tag in ("qwen", "small"):
        d, base, n = (int(x) for x in g[f"table.{tag}.meta"])
        assert np.array_equal(R.inv_freq(d, base), g[f"table.{tag}.inv_freq"]), tag
        emb = R.angle_table(n, d, base)
        mine = emb if n <= 64 else emb[:: n // 32]
        assert np.array_equal(mine, g[f"table.{tag}.emb"]), tag
    emb = R.angle_table(16, 16, 10000, position_ids=g["table.packed.pos"])
    assert np.array_equal(emb, g["table.packed.emb"])


def test_appliers_bit_exact_with_reference(golden_dir):
    g = _g(golden_dir)
    for tag in ("a", "b", "c", "partial"):
        out = R.apply_bshd(g[f"bshd.{tag}.t"], g[f"bshd.{tag}.freqs"])
        assert np.array_equal(out, g[f"bshd.{tag}.out"]), tag
    for tag in ("a", "b", "c"):  # what autograd gives through the reference's eager ops
        dt = R.apply_bshd_grad(g[f"bshd.{tag}.dout"], g[f"bshd.{tag}.freqs"])
        assert np.array_equal(dt, g[f"bshd.{tag}.dt"]), tag
    out = R.apply_thd(g["thd.t"], g["thd.cu"], g["thd.freqs"])
    assert np.array_equal(out, g["thd.out"])


def test_thd_equals_bshd_with_restarting_positions(golden_dir):
    g = _g(golden_dir)
    pos = R.thd_position_ids(g["thd.cu"])
    assert pos.tolist()[:8] == [0, 1, 2, 3, 4, 5, 6, 0]
    gathered = g["thd.freqs"][pos]                      # [T, 1, 1, rot]
    out = R.apply_bshd(g["thd.t"][:, None], gathered)[:, 0]
    assert np.array_equal(out, g["thd.out"])


def test_bf16_round_is_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -1.00390625, 3.0e38, 1e-40], dtype=np.float32)
    want = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(R.bf16_round(x), want)
    r = np.random.RandomState(0).randn(10000).astype(np.float32) * 100
    assert np.array_equal(R.bf16_round(r), torch.from_numpy(r).to(torch.bfloat16).float().numpy())


def test_host_rotary_embedding_tables(golden_dir):
    """The host mirror's RotaryEmbedding.forward on CPU: same table, position-id gather and CP slice."""
    from long_vita_amd import state
    from long_vita_amd.rotary_pos_embedding import RotaryEmbedding
    g = _g(golden_dir)
    state.set_position_ids(None)
    state.mpu.initialize(None, 1, 0)
    for tag in ("qwen", "small"):
        d, base, n = (int(x) for x in g[f"table.{tag}.meta"])
        re = RotaryEmbedding(d, 1.0, rotary_base=base, device="cpu")
        assert np.array_equal(re.inv_freq.numpy(), g[f"table.{tag}.inv_freq"])
        emb = re(n).numpy()
        mine = emb if n <= 64 else emb[:: n // 32]
        assert np.array_equal(mine, g[f"table.{tag}.emb"])
    re = RotaryEmbedding(16, 1.0, rotary_base=10000, device="cpu")
    state.set_position_ids(torch.from_numpy(g["table.packed.pos"]))
    try:
        assert np.array_equal(re(16).numpy(), g["table.packed.emb"])
    finally:
        state.set_position_ids(None)
    # context parallel: rank r keeps chunks r and 2CP-1-r of the table
    try:
        for cp in (2, 4):
            for r in range(cp):
                state.mpu.initialize(None, cp, r)
                want = R.angle_table(32, 16, 10000, cp_size=cp, cp_rank=r)
                assert np.array_equal(re(32).numpy(), want)
    finally:
        state.mpu.initialize(None, 1, 0)


def test_apply_refuses_cpu_and_interleaved():
    from long_vita_amd import _lib
    from long_vita_amd.rotary_pos_embedding import apply_rotary_pos_emb_bshd
    t = torch.zeros(4, 1, 2, 16, dtype=torch.bfloat16)
    f = torch.zeros(4, 1, 1, 16)
    with pytest.raises(NotImplementedError):
        apply_rotary_pos_emb_bshd(t, f, rotary_interleaved=True)
    with pytest.raises(_lib.LvAttnError):   # no CPU fallback
        apply_rotary_pos_emb_bshd(t, f)
