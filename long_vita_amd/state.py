"""Ambient state the reference's attention path reads from Megatron / MindSpeed:
``get_args()`` (dot_product_attention.py:303), ``get_actual_seq_len()`` (:38,333,372;
set per step at lcvlm_modellink/training/utils.py:672,767), ``get_position_ids()``
(rotary_pos_embedding.py:114) and the context-parallel group accessors of
``megatron.core.mpu`` (:46-49).  When the real packages are importable the adaptor
binds to them; standalone use (tests, bench) sets the values here."""
from __future__ import annotations

import types

_ARGS = None
_ACTUAL_SEQ_LEN = None
_POSITION_IDS = None

DEFAULT_ARGS = dict(
    use_flash_attn=True, reset_attention_mask=False, reset_position_ids=False,
    context_parallel_size=1, context_parallel_algo="megatron_cp_algo", cp_attention_mask_type="causal",
    use_cp_send_recv_overlap=False, sparse_mode=0, shape_order="SBH",
    pre_tockens=65536, next_tockens=0, vision_seq_length=1025, vision_context_parallel=False,
    sliding_window=None, use_kv_cache=False, seq_length=None,
)


def make_args(**overrides):
    d = dict(DEFAULT_ARGS)
    d.update(overrides)
    return types.SimpleNamespace(**d)


def set_args(args):
    global _ARGS
    _ARGS = args


def get_args():
    if _ARGS is None:
        raise RuntimeError("long_vita_amd.state: call set_args(make_args(...)) first")
    return _ARGS


def set_actual_seq_len(actual_seq_len):
    """Cumulative END offsets of the packed segments (python list of ints), as
    ``mindspeed.utils.set_actual_seq_len`` keeps them."""
    global _ACTUAL_SEQ_LEN
    _ACTUAL_SEQ_LEN = actual_seq_len


def get_actual_seq_len():
    return _ACTUAL_SEQ_LEN


def set_position_ids(position_ids):
    global _POSITION_IDS
    _POSITION_IDS = position_ids


def get_position_ids():
    return _POSITION_IDS


class ContextParallelState:
    """Stand-in for the ``mpu.get_context_parallel_*`` accessors."""

    def __init__(self):
        self.group = None
        self.world_size = 1
        self.rank = 0
        self.global_ranks = [0]
        self.group_for_send_recv_overlap = None

    def initialize(self, group, world_size, rank, global_ranks=None, overlap_group=None):
        self.group, self.world_size, self.rank = group, int(world_size), int(rank)
        self.global_ranks = list(global_ranks) if global_ranks is not None else list(range(self.world_size))
        self.group_for_send_recv_overlap = overlap_group

    def get_context_parallel_group(self):
        return self.group

    def get_context_parallel_world_size(self):
        return self.world_size

    def get_context_parallel_rank(self):
        return self.rank

    def get_context_parallel_global_ranks(self):
        return self.global_ranks

    def get_context_parallel_group_for_send_recv_overlap(self):
        return self.group_for_send_recv_overlap


mpu = ContextParallelState()
