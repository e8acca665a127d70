"""Host-side mirror of the reference's core-attention dispatch, backed by the
MI355X kernels.

Same names, arguments, branch selection and error behaviour as
lcvlm_modellink/core/transformer/dot_product_attention.py:
  flash_attention_forward(self, query, key, value, attention_mask, attn_mask_type,
                          packed_seq_params)                                (:291-474)
  do_ring_context_parallel(q, k, v, head_num, softmax_scale, attn_mask,
                           dropout_p=0., pse=None, pse_type=None)           (:36-76)
  dot_product_attention_forward_wrapper(fn)                                 (:148-181)
so that ``register_patch('modellink.core.transformer.dot_product_attention.
flash_attention_forward', flash_attention_forward)`` (megatron_adaptor.py:11-13)
can point at this module unchanged.  Differences, on purpose:
  * the global ``args`` object is never written (the reference mutates
    args.sparse_mode / shape_order / pre_tockens as a side effect, :329,383,389,406);
  * packed segments + context parallelism (:339-343) calls the ring WITHOUT the keyword the reference passes
    and do_ring_context_parallel does not take (a latent TypeError there); the ring reads the segments from
    get_actual_seq_len() as :38 does and keeps attention inside them;
  * alibi / pse and sliding window are outside this path and raise NotImplementedError;
  * the kv-cache prompt op always applies the causal mask aligned to the end of the keys (the
    reference forwards args.sparse_mode and its stored mask, which describe the same thing).
"""
from __future__ import annotations

import math
from functools import wraps

from . import state
from .fusion_attention import npu_fusion_attention, npu_incre_flash_attention, npu_prompt_flash_attention
from .ring_context_parallel import ringattn_context_parallel

_RING_ALGOS = ("megatron_cp_algo",)                       # the one algorithm built here
# the reference routes these to do_ring_context_parallel too (:339, :374-375); hybrid (ring x Ulysses, on the
# hybrid-ring subgroup and its own partition, :39-44) and the adaptive ones are not built and raise there
_REF_RING_BRANCH_ALGOS = ("megatron_cp_algo", "hybrid_cp_algo")
_ALL_CP_ALGOS = _REF_RING_BRANCH_ALGOS + ("adaptive_cp_algo", "hybrid_adaptive_cp_algo")


def _softmax_scale(self):
    # dot_product_attention.py:311-312 / :337 / :370-371
    if self.scale_mask_softmax.scale is None:
        return 1.0 / math.sqrt(self.hidden_size_per_attention_head)
    return self.softmax_scale


def do_ring_context_parallel(q, k, v, head_num, softmax_scale, attn_mask, dropout_p=0., pse=None, pse_type=None):
    args = state.get_args()
    actual_seq_len = state.get_actual_seq_len()
    mpu = state.mpu
    if args.context_parallel_algo not in _RING_ALGOS:
        raise NotImplementedError(f"context_parallel_algo={args.context_parallel_algo!r}: only the zig-zag ring "
                                  "(megatron_cp_algo) is built; hybrid / adaptive context parallelism use other "
                                  "process groups and another partition of the sequence")
    cp_para = dict(
        causal=args.cp_attention_mask_type == "causal",
        cp_group=mpu.get_context_parallel_group(),
        cp_size=mpu.get_context_parallel_world_size(),
        rank=mpu.get_context_parallel_rank(),
        cp_global_ranks=mpu.get_context_parallel_global_ranks(),
        cp_group_for_send_recv_overlap=(mpu.get_context_parallel_group_for_send_recv_overlap()
                                        if args.use_cp_send_recv_overlap else None),
        pse=pse, pse_type=pse_type,
        # one 8-GPU node: a single ring, no intra/inter-window split
        cp_inner_ranks=None, cp_outer_ranks=None, cp_dkv_outer_ranks=None,
        cp_group_for_intra_window=None, cp_group_for_intra_window_send_recv_overlap=None,
    )
    return ringattn_context_parallel(q, k, v, head_num, cp_para, softmax_scale, attn_mask, dropout_p,
                                     actual_seq_len, actual_seq_len)


def flash_attention_forward(self, query, key, value, attention_mask, attn_mask_type, packed_seq_params):
    if packed_seq_params is not None:
        raise AssertionError("packed_seq_params should be None.")
    args = state.get_args()
    seq_length, batch_size, n_head, head_dim = query.shape
    dropout_p = getattr(getattr(self, "attention_dropout", None), "p", 0.0)

    # ViT: selected by the sequence length (:309)
    if seq_length == args.vision_seq_length and not args.vision_context_parallel:
        q, k, v = (x.reshape(x.shape[0], x.shape[1], -1) for x in (query, key, value))
        return npu_fusion_attention(q, k, v, n_head, "SBH", pse=None, padding_mask=None, atten_mask=None,
                                    scale=_softmax_scale(self), pre_tockens=args.pre_tockens,
                                    next_tockens=args.next_tockens, keep_prob=1 - dropout_p,
                                    inner_precise=0, sparse_mode=0)[0]

    # LLM with per-segment mask reset (:328; `or args.reset_position_ids and False` never fires)
    if args.reset_attention_mask:
        actual_seq_len = state.get_actual_seq_len()
        assert actual_seq_len is not None
        scale = _softmax_scale(self)
        if args.context_parallel_size > 1 and args.context_parallel_algo in _REF_RING_BRANCH_ALGOS:
            # :339-343.  The reference passes `actual_seq_len=` to do_ring_context_parallel, which does not take it
            # (:36) and reads the same list from get_actual_seq_len() itself (:38): called without it, the ring gets
            # the segments the way the call was meant to deliver them.
            q, k, v = (x.reshape(x.shape[0], x.shape[1], -1) for x in (query, key, value))
            return do_ring_context_parallel(q, k, v, head_num=n_head, softmax_scale=scale, attn_mask=None)
        # 's b h d -> (b s) h d' (:345): a view when b == 1, as in every shipped script
        to_t = lambda x: x.permute(1, 0, 2, 3).reshape(batch_size * x.shape[0], x.shape[2], head_dim)
        out = npu_fusion_attention(to_t(query), to_t(key), to_t(value), n_head, "TND", pse=None, padding_mask=None,
                                   atten_mask=True, scale=scale, pre_tockens=args.pre_tockens,
                                   next_tockens=args.next_tockens, keep_prob=1 - dropout_p, inner_precise=0,
                                   sparse_mode=2, actual_seq_qlen=actual_seq_len, actual_seq_kvlen=actual_seq_len)[0]
        # '(b s) h d -> s b (h d)' (:367)
        return out.reshape(batch_size, seq_length, n_head * head_dim).permute(1, 0, 2)

    scale = _softmax_scale(self)
    if args.context_parallel_size > 1 and args.context_parallel_algo in _ALL_CP_ALGOS:
        q, k, v = (x.reshape(x.shape[0], x.shape[1], -1) for x in (query, key, value))
        return do_ring_context_parallel(q, k, v, head_num=n_head, softmax_scale=scale, attn_mask=attention_mask,
                                        pse=getattr(self, "pse", None), pse_type=getattr(self, "pse_type", None))

    # generic causal path (:381-474)
    if self.hidden_size_per_attention_head == 0:
        raise AssertionError("self.hidden_size_per_attention_head should not be ZERO.")
    if getattr(self, "alibi", None) is not None:
        if args.shape_order != "SBH":
            raise ValueError("FlashAttention with Alibi requires for SBH shape_order, but is {}.".format(args.shape_order))
        raise NotImplementedError("alibi is not supported")
    # :403-406: a sliding window sets pre_tockens (the reference writes it into the global args; here it stays local)
    pre_tockens = args.sliding_window if (args.sliding_window is not None and seq_length > args.sliding_window) else args.pre_tockens
    if getattr(args, "use_kv_cache", False):
        # :428-452: 's b h -> b s h' views of the cache, one new token -> incremental op, else the prompt op
        q, k, v = (x.reshape(x.shape[0], x.shape[1], -1).transpose(0, 1) for x in (query, key, value))
        if q.shape[1] == 1 and q.shape[1] != k.shape[1]:
            output = npu_incre_flash_attention(q, k, v, num_heads=n_head, input_layout="BSH", pse_shift=None,
                                               padding_mask=None, scale_value=scale)
        else:
            # the reference forwards args.sparse_mode and args.pre_tockens (:445, :449): in the op's band modes (0 / 1 / 4) a
            # prompt longer than pre_tockens is attended under that band
            output = npu_prompt_flash_attention(q, k, v, num_heads=n_head, input_layout="BSH", pse_shift=None,
                                                sparse_mode=args.sparse_mode, padding_mask=None, atten_mask=True,
                                                scale_value=scale, pre_tokens=pre_tockens, next_tokens=0)
        return output.transpose(0, 1)
    causal_mask = True  # generic path: the causal template mask (:384) / get_attention_mask()
    if args.shape_order == "TND":
        # '(s b)' flattening (:382): sequence-major, segments from actual_seq_len
        actual_seq_len = state.get_actual_seq_len()
        to_t = lambda x: x.reshape(x.shape[0] * batch_size, x.shape[2], head_dim)
        out = npu_fusion_attention(to_t(query), to_t(key), to_t(value), n_head, "TND", atten_mask=causal_mask, scale=scale,
                                   pre_tockens=pre_tockens, next_tockens=0, keep_prob=1 - dropout_p, sparse_mode=4,
                                   actual_seq_qlen=actual_seq_len, actual_seq_kvlen=actual_seq_len)[0]
        return out.reshape(seq_length, batch_size, n_head * head_dim)
    if args.shape_order == "BNSD":
        to_b = lambda x: x.permute(1, 2, 0, 3)
        out = npu_fusion_attention(to_b(query), to_b(key), to_b(value), n_head, "BNSD", atten_mask=causal_mask, scale=scale,
                                   pre_tockens=pre_tockens, next_tockens=args.next_tockens, keep_prob=1 - dropout_p,
                                   sparse_mode=args.sparse_mode)[0]
        return out.permute(2, 0, 1, 3).reshape(seq_length, batch_size, n_head * head_dim)
    q, k, v = (x.reshape(x.shape[0], x.shape[1], -1) for x in (query, key, value))
    # the reference forwards args.sparse_mode / pre_tockens / next_tockens (:462-466): in the op's band modes (0, the
    # default, 1 and 4) a sequence longer than pre_tockens (65,536 by default) is attended under that band
    return npu_fusion_attention(q, k, v, n_head, "SBH", atten_mask=causal_mask, scale=scale, pre_tockens=pre_tockens,
                                next_tockens=args.next_tockens, keep_prob=1 - dropout_p, sparse_mode=args.sparse_mode)[0]


def dot_product_attention_forward_wrapper(fn):
    """Replacement for Megatron's DotProductAttention.forward (:148-181): the flash path only.
    The eager baddbmm/softmax/bmm branch (:183-286) is the CPU baseline of this project and is
    not part of the GPU product path."""
    @wraps(fn)
    def wrapper(self, query, key, value, attention_mask, attn_mask_type, packed_seq_params):
        assert packed_seq_params is None, (
            "Packed sequence is not supported by DotProductAttention."
            "Please use TEDotProductAttention instead."
        )
        args = state.get_args()
        if not args.use_flash_attn:
            raise NotImplementedError("only the flash-attention path (--use-flash-attn) is provided on MI355X")
        return flash_attention_forward(self, query, key, value, attention_mask, attn_mask_type, packed_seq_params)
    return wrapper
