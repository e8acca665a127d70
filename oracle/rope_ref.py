"""CPU restatement of the reference's rotary position embedding (TEST INFRASTRUCTURE ONLY).

Follows lcvlm_modellink/core/models/common/embeddings/rotary_pos_embedding.py:
  RotaryEmbedding.__init__/forward  :60-122   inv_freq, angle table, packed position-id gather, CP slice
  get_pos_emb_on_this_cp_rank       :36-47    zig-zag slice of the table
  _rotate_half                      :162-178  (non-interleaved form; the shipped scripts never set rotary_interleaved)
  apply_rotary_pos_emb_bshd         :181-204
  apply_rotary_pos_emb_thd          :207-229
Pinned by tests/golden/rope.npz, which holds outputs of those functions themselves.

The reference runs the appliers on bf16 tensors with eager torch ops, so every elementwise op rounds
to bf16: cos/sin are rounded to bf16 first (:200-201), then t*cos, rotate_half(t)*sin and their sum
each round once more (:203).  This restatement reproduces that rounding sequence on float32 numpy
arrays, which makes bit-exact comparison meaningful.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = rounded.view(np.float32).copy()
    nan = np.isnan(x)
    out[nan] = np.nan
    return out.reshape(x.shape)


def inv_freq(dim: int, base: float) -> np.ndarray:
    """rotary_pos_embedding.py:76-82 (float32 throughout, as torch computes it)."""
    # torch's float32 pow, not numpy's: the two differ in the last bit for some exponents, and
    # the angle at position 1e6 magnifies that bit a million times
    import torch
    e = torch.arange(0, dim, 2, dtype=torch.float32) / dim
    return (1.0 / (base ** e)).numpy().astype(np.float32)


def angle_table(max_seq_len: int, dim: int, base: float, offset: int = 0, interpolation: float | None = None,
                position_ids: np.ndarray | None = None, cp_size: int = 1, cp_rank: int = 0,
                inv: np.ndarray | None = None) -> np.ndarray:
    """RotaryEmbedding.forward (:84-122): angles [s, b|1, 1, dim] float32, halves duplicated (:106)."""
    inv = inv_freq(dim, base) if inv is None else inv
    seq = np.arange(max_seq_len, dtype=np.float32) + np.float32(offset)
    if interpolation is not None:
        seq = seq * np.float32(1.0 / interpolation)
    freqs = np.outer(seq, inv).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)[:, None, None, :]
    if position_ids is not None:  # [s, b] (:114-117)
        s, b = position_ids.shape
        emb = emb[position_ids.reshape(-1)].squeeze(1).reshape(s, b, 1, -1)
    if cp_size > 1:  # (:119-121, :36-47)
        sh = emb.shape
        v = emb.reshape(2 * cp_size, -1, *sh[1:])
        emb = v[[cp_rank, 2 * cp_size - cp_rank - 1]].reshape(-1, *sh[1:])
    return emb


def rotate_half(x: np.ndarray) -> np.ndarray:
    """(:171-173)"""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_bshd(t: np.ndarray, freqs: np.ndarray) -> np.ndarray:
    """apply_rotary_pos_emb_bshd (:181-204) on bf16-valued float32 `t` [s, b, h, d]; freqs [s, b|1, 1, rot]."""
    rot = freqs.shape[-1]
    x, x_pass = t[..., :rot], t[..., rot:]
    cos_ = bf16_round(np.cos(freqs.astype(np.float32), dtype=np.float32))
    sin_ = bf16_round(np.sin(freqs.astype(np.float32), dtype=np.float32))
    a = bf16_round(x * cos_)
    b = bf16_round(rotate_half(x) * sin_)
    return np.concatenate([bf16_round(a + b), x_pass], axis=-1)


def apply_bshd_grad(dout: np.ndarray, freqs: np.ndarray) -> np.ndarray:
    """What torch autograd yields through the eager ops of (:203) in bf16:
    dt = bf16( bf16(dout*cos) + rotate_half^T( bf16(dout*sin) ) ), rotate_half^T(z) = cat(z2, -z1)."""
    rot = freqs.shape[-1]
    g, g_pass = dout[..., :rot], dout[..., rot:]
    cos_ = bf16_round(np.cos(freqs.astype(np.float32), dtype=np.float32))
    sin_ = bf16_round(np.sin(freqs.astype(np.float32), dtype=np.float32))
    a = bf16_round(g * cos_)
    z = bf16_round(g * sin_)
    h = rot // 2
    zt = np.concatenate([z[..., h:], -z[..., :h]], axis=-1)
    return np.concatenate([bf16_round(a + zt), g_pass], axis=-1)


def apply_thd(t: np.ndarray, cu_seqlens: np.ndarray, freqs: np.ndarray) -> np.ndarray:
    """apply_rotary_pos_emb_thd (:207-229): t [T, h, d]; positions restart at every segment."""
    outs = []
    for a, b in zip(cu_seqlens[:-1], cu_seqlens[1:]):
        x = t[int(a):int(b)]
        outs.append(apply_bshd(x[:, None], freqs[: x.shape[0]])[:, 0])
    return np.concatenate(outs, axis=0)


def thd_position_ids(cu_seqlens: np.ndarray) -> np.ndarray:
    """Row -> position inside its segment: what the thd applier's per-segment freqs[:len] amounts to."""
    total = int(cu_seqlens[-1])
    pos = np.arange(total, dtype=np.int64)
    starts = np.repeat(np.asarray(cu_seqlens[:-1], dtype=np.int64), np.diff(cu_seqlens).astype(np.int64))
    return pos - starts
