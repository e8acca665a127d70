"""Spreading the image encode of one long sample over the context-parallel group.

In the reference every CP rank encodes, by itself, all the images that have at least one token in its two zig-zag
chunks (lcvlm_modellink/training/utils.py:271-281 keeps exactly those images on the rank; the vision model then
runs on them, pretrain_lcvlm.py:509-514).  How many images a rank gets therefore depends on where the images sit
in the sequence: a video packed at the front of a 1M-token sample lands on two ranks while six idle.  Here the
ENCODE is dealt round-robin over the ranks (image i on rank i % CP, so every rank encodes n/CP of them), and the
features are then routed point to point to exactly the ranks that own tokens of each image: the rank ends up with
the same tensor the reference's per-rank encode produces -- the features of its images, in image order -- and
``external_src_indices`` / ``external_tgt_indices`` from ``cp_partition`` apply to it unchanged.

(The reference's own option for this stage, ``--vision-context-parallel``, splits the TOKENS of every image over
the ranks instead and sends the ViT through the language model's causal ring, intern_vit_model.py:218-233 with
dot_product_attention.py:309,374-379; no shipped script uses it and it is not built.)
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .cp_partition import zigzag_index


def images_needed_by_rank(external_indices: torch.Tensor, seq_length: int, cp_size: int, cp_rank: int) -> torch.Tensor:
    """Sorted ids of the images with a token in rank ``cp_rank``'s chunks: the selection of training/utils.py:271-281
    (``mask = isin(indices_s, calibration_index); images[mask.any(-1)]``).  external_indices [2, n_img, T]."""
    calib = zigzag_index(seq_length, cp_size, cp_rank, device=external_indices.device)
    mask = torch.isin(external_indices[1], calib)
    return torch.nonzero(mask.any(dim=1), as_tuple=False).reshape(-1)


def encode_images_context_parallel(images: torch.Tensor, external_indices: torch.Tensor, seq_length: int, cp_size: int,
                                   cp_rank: int, encode_fn, group=None, global_ranks=None) -> torch.Tensor:
    """images [n_img, ...] and external_indices [2, n_img, T] as every rank holds them before the CP split;
    ``encode_fn(images_subset) -> [n, T, h]`` (the vision model).  Returns the features of the images this rank
    needs, in image order: what ``encode_fn(images[images_needed_by_rank(...)])`` gives, with 1/CP of the work."""
    n_img = images.shape[0]
    ranks = list(global_ranks) if global_ranks is not None else list(range(cp_size))
    mine = torch.arange(cp_rank, n_img, cp_size, device=images.device)          # round-robin deal
    feats = encode_fn(images[mine]) if mine.numel() else None
    if cp_size == 1:
        return feats
    need = [images_needed_by_rank(external_indices, seq_length, cp_size, r).tolist() for r in range(cp_size)]
    my_need = need[cp_rank]
    # probe the feature shape / dtype on a rank that encoded nothing (n_img < cp_size)
    meta = [None] * cp_size
    dist.all_gather_object(meta, None if feats is None else (tuple(feats.shape[1:]), str(feats.dtype)), group=group)
    shape, dtype_name = next(m for m in meta if m is not None)
    dtype = getattr(torch, dtype_name.replace("torch.", ""))
    out = torch.empty((len(my_need),) + shape, dtype=dtype, device=images.device)
    pos_in_out = {img: k for k, img in enumerate(my_need)}
    ops, recv_bufs = [], []
    for peer in range(cp_size):
        if peer == cp_rank:
            continue
        send_ids = [i for i in need[peer] if i % cp_size == cp_rank]           # mine, wanted by peer
        recv_ids = [i for i in my_need if i % cp_size == peer]                 # theirs, wanted by me
        if send_ids:
            rows = torch.tensor([i // cp_size for i in send_ids], device=images.device)
            ops.append(dist.P2POp(dist.isend, feats[rows].contiguous(), ranks[peer], group))
        if recv_ids:
            buf = torch.empty((len(recv_ids),) + shape, dtype=dtype, device=images.device)
            recv_bufs.append((recv_ids, buf))
            ops.append(dist.P2POp(dist.irecv, buf, ranks[peer], group))
    works = dist.batch_isend_irecv(ops) if ops else []
    for i in my_need:                                                          # local ones meanwhile
        if i % cp_size == cp_rank:
            out[pos_in_out[i]] = feats[i // cp_size]
    for wk in works:
        wk.wait()
    for recv_ids, buf in recv_bufs:
        out[torch.tensor([pos_in_out[i] for i in recv_ids], device=images.device)] = buf
    return out
