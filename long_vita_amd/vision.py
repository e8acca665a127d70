"""Vision-token compression and the scatter of visual features into the LLM input.

  pixel_shuffle / vit_output_to_llm_tokens   lcvlm_modellink/pretrain_lcvlm.py:463-470, 546-556
  pixel_shuffle_v2                           the stride form, pretrain_lcvlm.py:472-476, 558-570
  scatter_external_features                  lcvlm_modellink/core/models/common/embeddings/
                                             language_model_embedding.py:114-134
  vit_attention                              the InternViT-300M core attention call
                                             (dot_product_attention.py:309-325; 24 layers of
                                             16 heads x 64, 1025 tokens: pretrain_lcvlm.py:191-213)
All three run as HIP kernels (pure 16-byte moves for the first two); there is no CPU path.
"""
from __future__ import annotations

import math

import torch

from . import ops
from .fusion_attention import npu_fusion_attention

INTERN_VIT_300M = dict(num_layers=24, hidden_size=1024, num_attention_heads=16, kv_channels=64,
                       seq_length=1025, patch_dim=14, img_h=448, img_w=448, image_token_length=256)


def pixel_shuffle(x: torch.Tensor, scale_factor: float = 0.5) -> torch.Tensor:
    """[N, W, H, C] -> [N, W/2, H/2, 4C] (only scale_factor 0.5, the value every script uses)."""
    if scale_factor != 0.5:
        raise NotImplementedError("only scale_factor=0.5 is built")
    n, w, h, c = x.shape
    if w != h:
        raise ValueError("pixel_shuffle expects a square token grid")
    out = ops.pixel_shuffle_tokens(x.reshape(n, w * h, c), skip=0)
    return out.reshape(n, w // 2, h // 2, 4 * c)


def pixel_shuffle_v2(x: torch.Tensor, scale_stride: int = 2) -> torch.Tensor:
    """[N, W, H, C] (W == H) -> [N, ceil(W/s) * ceil(H/s), s*s*C]: every s x s patch of the grid becomes one token, zeros
    where a patch hangs over the right / bottom edge (pretrain_lcvlm.py:558-570)."""
    n, w, h, c = x.shape
    assert w == h
    return ops.pixel_shuffle_v2_tokens(x.reshape(n, w * h, c), skip=0, scale_stride=int(scale_stride))


def vit_output_to_llm_tokens(vit_output: torch.Tensor, add_class_token: bool = True, scale_stride: int = 1) -> torch.Tensor:
    """[N, 1025, C] -> [N, 256, 4C]: class-token drop + reshape + pixel_shuffle(0.5) + flatten,
    fused into one pass over the data; with ``scale_stride`` > 1 the stride form (pixel_shuffle_v2) instead."""
    skip = 1 if add_class_token else 0
    if scale_stride != 1:
        return ops.pixel_shuffle_v2_tokens(vit_output, skip=skip, scale_stride=int(scale_stride))
    return ops.pixel_shuffle_tokens(vit_output, skip=skip)


def scatter_external_features(word_embeddings: torch.Tensor, external_feature_dict: dict) -> torch.Tensor:
    """The visual-feature scatter of LanguageModelEmbedding.forward (language_model_embedding.py:103-134), in place
    on ``word_embeddings`` [b, s, h] (bf16, contiguous; the output of the embedding lookup, not a leaf) and
    differentiable as the reference's indexed assignments are: ``features`` receives the gradient of the rows it
    was written to, the overwritten rows of ``word_embeddings`` pass nothing back to the embedding table, and with
    no index entry at all ``features`` still joins the graph with a zero gradient (:134).  Returns ``word_embeddings``."""
    assert "features" in external_feature_dict \
        and (len(external_feature_dict) == 1
             or len(external_feature_dict) == 2 and "pre_len" in external_feature_dict
             or len(external_feature_dict) == 2 and "indices" in external_feature_dict
             or len(external_feature_dict) == 3 and "src_indices" in external_feature_dict
             and "tgt_indices" in external_feature_dict), "The format of external_feature_dict is not right!"
    features = external_feature_dict["features"]
    b, s, h = word_embeddings.shape
    flat = word_embeddings.view(b * s, h)
    if "indices" in external_feature_dict:
        ind_b, ind_s = external_feature_dict["indices"].unbind(dim=0)
        dst = (ind_b.reshape(-1) * s + ind_s.reshape(-1))
        ops.scatter_rows(flat, dst, features.reshape(-1, features.shape[-1]))
    elif "pre_len" in external_feature_dict:
        pre = int(external_feature_dict["pre_len"])
        n = features.shape[1]
        if features.shape[0] != b or pre + n > s:
            raise ValueError(f"features {tuple(features.shape)} do not fit rows {pre}..{pre + n} of {tuple(word_embeddings.shape)}")
        dst = (torch.arange(b, device=features.device)[:, None] * s + torch.arange(pre, pre + n, device=features.device)[None])
        ops.scatter_rows(flat, dst.reshape(-1), features.reshape(-1, h))
    elif "src_indices" in external_feature_dict and "tgt_indices" in external_feature_dict:
        src_b, src_s = external_feature_dict["src_indices"]
        tgt_b, tgt_s = external_feature_dict["tgt_indices"]
        n_tok = features.shape[1]
        ops.scatter_rows(flat, tgt_b * s + tgt_s, features.reshape(-1, features.shape[-1]), src_b * n_tok + src_s)
    else:
        ops.keep_graph_alive_(flat, features)
    return word_embeddings


def language_model_embedding_forward(self, input_ids, position_ids, tokentype_ids=None, external_feature_dict=None):
    """``LanguageModelEmbedding.forward`` (language_model_embedding.py:91-174) with the scatter above in place of the
    indexed assignments: same arguments, same attribute reads on ``self`` (``word_embeddings``,
    ``add_position_embedding`` / ``position_embeddings``, ``tokentype_embeddings``, ``config``,
    ``embedding_dropout``), result [s, b, h].  Sequence parallelism (a tensor-parallel feature, :157-167) is not on
    this path and raises."""
    word_embeddings = self.word_embeddings(input_ids)
    if external_feature_dict is not None:
        if not word_embeddings.is_contiguous():
            word_embeddings = word_embeddings.contiguous()
        word_embeddings = scatter_external_features(word_embeddings, external_feature_dict)
    if self.add_position_embedding:
        embeddings = word_embeddings + self.position_embeddings(position_ids)
    else:
        embeddings = word_embeddings
    embeddings = embeddings.transpose(0, 1).contiguous()
    if tokentype_ids is not None:
        assert self.tokentype_embeddings is not None
        embeddings = embeddings + self.tokentype_embeddings(tokentype_ids).permute(1, 0, 2)
    else:
        assert self.tokentype_embeddings is None
    if self.config.fp32_residual_connection:
        embeddings = embeddings.float()
    if self.config.sequence_parallel:
        raise NotImplementedError("sequence parallelism belongs to tensor parallelism, which this path does not cover")
    return self.embedding_dropout(embeddings)


def vit_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, scale: float | None = None) -> torch.Tensor:
    """Full (bidirectional) multi-head attention in the ViT layout: q/k/v [s, b, h, d] -> [s, b, h*d]."""
    s, b, h, d = query.shape
    scale = 1.0 / math.sqrt(d) if scale is None else scale
    q, k, v = (x.reshape(s, b, h * d) for x in (query, key, value))
    return npu_fusion_attention(q, k, v, h, "SBH", atten_mask=None, scale=scale, sparse_mode=0)[0]
