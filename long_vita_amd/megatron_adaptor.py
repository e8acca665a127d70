"""Plug-in glue: installs the MI355X attention path under the dotted names the
reference patches (lcvlm_modellink/megatron_adaptor.py:11-13, 107-120).

``register_patches(manager)`` works with MindSpeed's ``MindSpeedPatchesManager``
(anything exposing ``register_patch(dotted_name, fn, force_patch=...)``);
``bind_runtime()`` points this package's ambient-state accessors at the real
Megatron / MindSpeed ones when those packages are importable."""
from __future__ import annotations

from . import state
from .core_attention import (do_ring_context_parallel, dot_product_attention_forward_wrapper,  # noqa: F401  (the wrapper is re-exported)
                             flash_attention_forward)
from .ring_context_parallel import ringattn_context_parallel
from .rotary_pos_embedding import RotaryEmbedding, apply_rotary_pos_emb

PATCHES = {
    "modellink.core.transformer.dot_product_attention.flash_attention_forward": flash_attention_forward,
    "modellink.core.transformer.dot_product_attention.do_ring_context_parallel": do_ring_context_parallel,
    "mindspeed.core.context_parallel.ring_context_parallel.ringattn_context_parallel": ringattn_context_parallel,
    # the dotted names the reference's own (commented-out) rotary patch uses, lcvlm_modellink/megatron_adaptor.py:92-93
    "megatron.core.models.common.embeddings.rotary_pos_embedding.RotaryEmbedding": RotaryEmbedding,
    "megatron.core.models.common.embeddings.rotary_pos_embedding.apply_rotary_pos_emb": apply_rotary_pos_emb,
}


def register_patches(manager, force_patch: bool = True):
    for name, fn in PATCHES.items():
        try:
            manager.register_patch(name, fn, force_patch=force_patch)
        except TypeError:
            manager.register_patch(name, fn)
    return list(PATCHES)


def bind_runtime():
    """Read args / actual_seq_len / CP groups from Megatron + MindSpeed instead of
    long_vita_amd.state.  Raises ImportError when they are not installed."""
    from megatron.training import get_args
    from megatron.core import mpu
    from mindspeed.utils import get_actual_seq_len, get_position_ids
    state.get_args = get_args
    state.get_actual_seq_len = get_actual_seq_len
    state.get_position_ids = get_position_ids
    state.mpu = mpu
