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
    """Angle table of the rotary embedding behind the reference class's interface (:50-122): constructor
    ``(kv_channels, rotary_percent, rotary_interleaved, seq_len_interpolation_factor, rotary_base)``,
    ``forward(max_seq_len, offset) -> [s, b | 1, 1, rot]`` fp32 angles, ``get_rotary_seq_len(...)``."""

    def __init__(self, kv_channels: int, rotary_percent: float, rotary_interleaved: bool = False,
                 seq_len_interpolation_factor: float = None, rotary_base: int = 10000, device=None) -> None:
        super().__init__()
        rot = kv_channels if rotary_percent >= 1.0 else int(kv_channels * rotary_percent)
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else "cpu"
        exponent = torch.arange(0, rot, 2, dtype=torch.float32, device=device) / rot
        self.inv_freq = 1.0 / (rotary_base ** exponent)          # one frequency per rotated column pair
        self.rotary_interleaved = rotary_interleaved
        self.seq_len_interpolation_factor = seq_len_interpolation_factor

    def forward(self, max_seq_len: int, offset: int = 0) -> torch.Tensor:
        positions = torch.arange(max_seq_len, device=self.inv_freq.device, dtype=self.inv_freq.dtype) + offset
        if self.seq_len_interpolation_factor is not None:
            positions = positions * (1 / self.seq_len_interpolation_factor)   # the product, not a division: same rounding
        angles = positions[:, None] * self.inv_freq[None, :]                  # [s, rot / 2]
        # half layout [a0 .. an, a0 .. an] pairs column j with j + rot/2; interleaved layout [a0, a0, a1, a1, ..]
        table = angles.repeat_interleave(2, dim=-1) if self.rotary_interleaved else angles.repeat(1, 2)
        table = table[:, None, None, :]
        position_ids = state.get_position_ids()
        if position_ids is not None:      # packed samples: [s, b] positions that restart per sample (:114-117)
            s, b = position_ids.shape
            table = table[position_ids.reshape(-1)].reshape(s, b, 1, -1)
        if state.mpu.get_context_parallel_world_size() > 1:                   # zig-zag slice of this rank (:119-121)
            table = get_pos_emb_on_this_cp_rank(table, 0)
        return table

    def get_rotary_seq_len(self, inference_params, transformer, transformer_input, transformer_config) -> float:
        """Length of the table the model asks ``forward`` for (:128-159): the whole sequence, i.e. the local length
        times every factor by which the sequence dimension is split at this point of the model."""
        if inference_params is not None:
            local = inference_params.max_sequence_length
        else:
            source = transformer.input_tensor if transformer.input_tensor is not None else transformer_input
            local = source.size(0)
            if transformer_config.sequence_parallel:
                local = local * transformer_config.tensor_model_parallel_size
        return local * transformer_config.context_parallel_size


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


def rope_flash_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, freqs: torch.Tensor,
                         actual_seq_len=None, scale: float | None = None) -> torch.Tensor:
    """apply_rotary_pos_emb on query and key followed by causal flash attention, as ONE op (what
    --use-fused-rotary-pos-emb + --use-flash-attn amount to between Megatron's SelfAttention QKV projection and its
    output projection): query [s, 1, Hq, 128], key / value [s, 1, Hkv, 128] UNROTATED, freqs [s, 1, 1, 128] fp32 angles
    (already gathered by position ids / sliced for the CP rank as RotaryEmbedding.forward leaves them, :114-121) ->
    [s, 1, Hq*128].  ``actual_seq_len`` = cumulative END offsets of packed segments (None: one causal segment).
    The keys are rotated by the bandwidth-bound kernel, the queries inside the attention kernel; results and gradients
    are bit-identical to apply_rotary_pos_emb + flash_attention_forward on this package's ping-pong forward kernel, and
    within kernel rounding of them where the plain path picks the one-wave kernel (key sweeps of 32,768 and more)."""
    _check(query, False)
    s, b, hq, d = query.shape
    if b != 1:
        raise ops.LvAttnError("rope_flash_attention needs micro-batch 1 (as every shipped script uses)")
    if freqs.shape[0] < s or freqs.shape[-1] != d:
        raise ops.LvAttnError(f"freqs {tuple(freqs.shape)} does not cover [{s}, .., {d}]")
    cos_t, sin_t = _tables(freqs)
    cu = [0, s] if actual_seq_len is None else [0] + [int(x) for x in actual_seq_len]
    out, _ = ops.rope_varlen_attention(query.reshape(s, hq, d), key.reshape(s, -1, d), value.reshape(s, -1, d), cos_t, sin_t, cu,
                                       scale, True)
    return out.reshape(s, 1, hq * d)
