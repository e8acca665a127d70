"""Rotary position embedding with the reference's interface
(lcvlm_modellink/core/models/common/embeddings/rotary_pos_embedding.py): ``RotaryEmbedding`` builds the
angle table (:60-122, with the packed position-id gather of :114-117 and the zig-zag context-parallel
slice of :119-121), ``apply_rotary_pos_emb`` (:232-259) rotates query / key.  The rotation runs in the
HIP kernel behind ``lv_rope``; there is no eager fallback.

The reference's eager appliers recompute ``cos(freqs)`` and ``sin(freqs)`` in every call, i.e. twice per
layer; here the two bf16 tables are derived once per ``freqs`` tensor and kept while that tensor lives.
"""
from __future__ import annotations

import weakref

import torch

from . import ops, state
from .cp_partition import get_pos_emb_on_this_cp_rank

__all__ = ["RotaryEmbedding", "apply_rotary_pos_emb", "apply_rotary_pos_emb_bshd", "apply_rotary_pos_emb_thd"]


class RotaryEmbedding(torch.nn.Module):
    """Same constructor and ``forward(max_seq_len, offset)`` as the reference class (:50-122)."""

    def __init__(self, kv_channels: int, rotary_percent: float, rotary_interleaved: bool = False,
                 seq_len_interpolation_factor: float = None, rotary_base: int = 10000, device=None) -> None:
        super().__init__()
        dim = kv_channels
        if rotary_percent < 1.0:
            dim = int(dim * rotary_percent)
        self.rotary_interleaved = rotary_interleaved
        self.seq_len_interpolation_factor = seq_len_interpolation_factor
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else "cpu"
        self.inv_freq = 1.0 / (rotary_base ** (torch.arange(0, dim, 2, dtype=torch.float32, device=device) / dim))

    def forward(self, max_seq_len: int, offset: int = 0) -> torch.Tensor:
        seq = torch.arange(max_seq_len, device=self.inv_freq.device, dtype=self.inv_freq.dtype) + offset
        if self.seq_len_interpolation_factor is not None:
            seq *= 1 / self.seq_len_interpolation_factor
        freqs = torch.outer(seq, self.inv_freq)
        if not self.rotary_interleaved:
            emb = torch.cat((freqs, freqs), dim=-1)
        else:
            emb = torch.stack((freqs.view(-1, 1), freqs.view(-1, 1)), dim=-1).view(freqs.shape[0], -1)
        emb = emb[:, None, None, :]
        position_ids = state.get_position_ids()
        if position_ids is not None:  # [s, b]: positions restart per packed sample
            s, b = position_ids.shape
            emb = emb[position_ids.view(-1)].squeeze(1).reshape(s, b, 1, -1)
        if state.mpu.get_context_parallel_world_size() > 1:
            emb = get_pos_emb_on_this_cp_rank(emb, 0)
        return emb

    def get_rotary_seq_len(self, inference_params, transformer, transformer_input, transformer_config) -> float:
        """(:128-159)"""
        if inference_params is not None:
            rotary_seq_len = inference_params.max_sequence_length
        else:
            if transformer.input_tensor is not None:
                rotary_seq_len = transformer.input_tensor.size(0)
            else:
                rotary_seq_len = transformer_input.size(0)
            if transformer_config.sequence_parallel:
                rotary_seq_len *= transformer_config.tensor_model_parallel_size
        rotary_seq_len *= transformer_config.context_parallel_size
        return rotary_seq_len


# cos / sin tables per live `freqs` tensor: id -> (weakref, version, cos, sin)
_TABLES: dict = {}


def _tables(freqs: torch.Tensor):
    """bf16 cos / sin of the angle tensor [n, b|1, 1, rot] as [n * b, rot] tables (:200-201)."""
    key = id(freqs)
    hit = _TABLES.get(key)
    if hit is not None and hit[0]() is freqs and hit[1] == freqs._version:
        return hit[2], hit[3]
    rot = freqs.shape[-1]
    flat = freqs.reshape(-1, rot)
    cos_t = torch.cos(flat).to(torch.bfloat16).contiguous()
    sin_t = torch.sin(flat).to(torch.bfloat16).contiguous()
    if len(_TABLES) > 16:
        for k in [k for k, v in _TABLES.items() if v[0]() is None]:
            del _TABLES[k]
    try:
        _TABLES[key] = (weakref.ref(freqs), freqs._version, cos_t, sin_t)
    except TypeError:
        pass
    return cos_t, sin_t


def _check(t: torch.Tensor, rotary_interleaved: bool):
    if rotary_interleaved:
        raise NotImplementedError("rotary_interleaved=True is not used by any Long-VITA script and is not built")
    if t.dtype != torch.bfloat16:
        raise ops.LvAttnError("apply_rotary_pos_emb: the HIP path takes bfloat16 tensors")


def apply_rotary_pos_emb_bshd(t: torch.Tensor, freqs: torch.Tensor, rotary_interleaved: bool = False) -> torch.Tensor:
    """t [s, b, h, d]; freqs [s, 1|b, 1, rot] fp32 angles (:181-204)."""
    _check(t, rotary_interleaved)
    s, b, h, d = t.shape
    if freqs.shape[0] < s:
        raise ops.LvAttnError(f"freqs covers {freqs.shape[0]} positions, the tensor has {s}")
    fb = freqs.shape[1]
    if fb not in (1, b):
        raise ops.LvAttnError(f"freqs batch dimension {fb} does not broadcast against {b}")
    cos_t, sin_t = _tables(freqs)
    out = ops.rope(t.reshape(s * b, h, d), cos_t, sin_t, None, b if fb == 1 else 1)
    return out.view(s, b, h, d)


def _thd_positions(cu_seqlens: torch.Tensor, total: int, device) -> torch.Tensor:
    cu = cu_seqlens.to(device=device, dtype=torch.int64)
    lens = cu[1:] - cu[:-1]
    starts = torch.repeat_interleave(cu[:-1], lens, output_size=total)
    return (torch.arange(total, device=device, dtype=torch.int64) - starts).to(torch.int32)


def apply_rotary_pos_emb_thd(t: torch.Tensor, cu_seqlens: torch.Tensor, freqs: torch.Tensor,
                             rotary_interleaved: bool = False) -> torch.Tensor:
    """t [T, h, d] packed; positions restart at every cu_seqlens boundary (:207-229)."""
    _check(t, rotary_interleaved)
    total = t.shape[0]
    if freqs.dim() != 4 or freqs.shape[1] != 1:
        raise ops.LvAttnError("apply_rotary_pos_emb_thd expects freqs [max_s, 1, 1, rot]")
    cos_t, sin_t = _tables(freqs)
    pos = _thd_positions(cu_seqlens, total, t.device)
    if total and int(pos.max()) >= cos_t.shape[0]:
        raise ops.LvAttnError("a packed segment is longer than the rotary table")
    return ops.rope(t, cos_t, sin_t, pos, 1)


def apply_rotary_pos_emb(t: torch.Tensor, freqs: torch.Tensor, config, cu_seqlens=None):
    """Same routing as the reference (:232-259); the fused and the eager variants both land on the HIP kernel."""
    interleaved = bool(getattr(config, "rotary_interleaved", False))
    if cu_seqlens is None:
        return apply_rotary_pos_emb_bshd(t, freqs, rotary_interleaved=interleaved)
    return apply_rotary_pos_emb_thd(t, cu_seqlens, freqs, rotary_interleaved=interleaved)
