"""Zig-zag context-parallel partitioning of a batch and its inverse.

Mirrors, with the same names and results (bit-exact: integer / copy work only):
  get_batch_on_this_cp_rank                    lcvlm_modellink/training/utils.py:224-260
  _get_batch_on_this_cp_rank_in_megatron_cp    lcvlm_modellink/training/utils.py:263-364
  index_of_a_in_b                              lcvlm_modellink/training/utils.py:539-542
  get_pos_emb_on_this_cp_rank                  .../embeddings/rotary_pos_embedding.py:36-47
  rotary position gather                       .../embeddings/rotary_pos_embedding.py:114-117
  sync_output (un-zig-zag after all_gather)    lcvlm_modellink/tasks/inference/text_generation/utils.py:557-581
Index arithmetic is done with torch integer ops on whatever device the inputs live
on; moving the rows of large GPU tensors (RoPE tables, activations) goes through the
HIP row-gather kernel.
"""
from __future__ import annotations

import torch

from . import state


def zigzag_index(seq_length: int, cp_size: int, cp_rank: int, device=None) -> torch.Tensor:
    """Global positions owned by ``cp_rank``: chunks r and 2*CP-1-r (utils.py:273 'calibration_index')."""
    if seq_length % (2 * cp_size):
        raise ValueError(f"sequence length {seq_length} is not divisible by 2*CP = {2 * cp_size}")
    return torch.arange(seq_length, device=device).view(2 * cp_size, seq_length // (2 * cp_size))[
        [cp_rank, 2 * cp_size - cp_rank - 1]].reshape(-1)


def index_of_a_in_b(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """For every element of ``a`` its position in ``b`` (every element present, ``b`` free of duplicates):
    ``b[result] == a``.  One sort of ``b`` and a binary search instead of the reference's three argsorts
    (lcvlm_modellink/training/utils.py:539-542); the result is the same integer tensor."""
    sorted_b, where_in_b = torch.sort(b)
    return where_in_b[torch.searchsorted(sorted_b, a)]


def _take_rows(val: torch.Tensor, index: torch.Tensor, dim: int, unique: bool = False) -> torch.Tensor:
    if val.is_cuda and dim == 0 and val.dim() >= 2 and (val[0].numel() * val.element_size()) % 4 == 0:
        from . import ops
        return ops.gather_rows(val, index, unique_index=unique)     # differentiable in val
    return val.index_select(dim, index.to(val.device))


def zigzag_slice(val: torch.Tensor, cp_size: int, cp_rank: int, seq_dim: int) -> torch.Tensor:
    """utils.py:351-362: view [.., 2CP, s/2CP, ..] -> index_select([r, 2CP-1-r]) -> flatten."""
    idx = zigzag_index(val.shape[seq_dim], cp_size, cp_rank, device=val.device)
    return _take_rows(val, idx, seq_dim, unique=True)


def _get_batch_on_this_cp_rank_in_megatron_cp(batch, cp_size=None, cp_rank=None, seq_length=None):
    args = state.get_args()
    cp_rank = state.mpu.get_context_parallel_rank() if cp_rank is None else cp_rank
    cp_size = state.mpu.get_context_parallel_world_size() if cp_size is None else cp_size
    seq_length = args.seq_length if seq_length is None else seq_length
    for key, val in list(batch.items()):
        if key == "external_images":
            if "external_indices" in batch:
                ind = batch["external_indices"]
                calib = zigzag_index(seq_length, cp_size, cp_rank, device=ind.device)
                mask = torch.isin(ind[1], calib)
                if mask.any():
                    batch["external_images"] = val[torch.any(mask, dim=1).to(val.device), ...]
            continue
        if key == "external_indices":
            calib = zigzag_index(seq_length, cp_size, cp_rank, device=val.device)
            indices_b, indices_s = val.unbind(dim=0)
            mask = torch.isin(indices_s, calib)
            if mask.any():
                selected = torch.any(mask, dim=1)
                n_img = int(selected.sum())
                n_tok = indices_s.shape[1]
                src_b = torch.arange(n_img, device=val.device).unsqueeze(1).repeat(1, n_tok)
                src_s = torch.arange(n_tok, device=val.device).unsqueeze(0).repeat(n_img, 1)
                msel = mask[selected]
                tgt_s = index_of_a_in_b(indices_s[mask], calib)
                batch["external_src_indices"] = torch.stack([src_b[msel], src_s[msel]])
                batch["external_tgt_indices"] = torch.stack([indices_b[mask], tgt_s])
            batch.pop(key)
            continue
        if key == "attention_mask":
            continue
        if val is not None:
            batch[key] = zigzag_slice(val, cp_size, cp_rank, 1)
    return batch


def get_batch_on_this_cp_rank(batch):
    args = state.get_args()
    if args.reset_position_ids:
        state.set_position_ids(batch["position_ids"].transpose(0, 1).contiguous())
    cp_size = args.context_parallel_size
    if not cp_size > 1:
        return batch
    if args.context_parallel_algo == "megatron_cp_algo" and args.cp_attention_mask_type != "general":
        return _get_batch_on_this_cp_rank_in_megatron_cp(batch)
    raise NotImplementedError(f"context_parallel_algo={args.context_parallel_algo!r} / "
                              f"cp_attention_mask_type={args.cp_attention_mask_type!r}: only the causal ring is built")


def get_pos_emb_on_this_cp_rank(pos_emb: torch.Tensor, seq_dim: int, cp_size=None, cp_rank=None) -> torch.Tensor:
    cp_size = state.mpu.get_context_parallel_world_size() if cp_size is None else cp_size
    cp_rank = state.mpu.get_context_parallel_rank() if cp_rank is None else cp_rank
    return zigzag_slice(pos_emb, cp_size, cp_rank, seq_dim)


def gather_rotary_by_position_ids(emb: torch.Tensor, position_ids: torch.Tensor) -> torch.Tensor:
    """emb [S,1,1,d], position_ids [s,b] -> [s,b,1,d]  (rotary_pos_embedding.py:114-117)."""
    s, b = position_ids.shape
    rows = _take_rows(emb.reshape(emb.shape[0], -1), position_ids.reshape(-1), 0)
    return rows.reshape(s, b, 1, -1)


def unzigzag_gathered(output_list, cp_size: int, seq_dim: int = 1) -> torch.Tensor:
    """The reordering half of ``sync_output``: ``output_list[r]`` is rank r's tensor after
    the all_gather; split each in two along the sequence, order the halves by chunk id."""
    halves = [h for x in output_list for h in x.chunk(2, dim=seq_dim)]
    ids = torch.tensor([c for r in range(cp_size) for c in (r, 2 * cp_size - r - 1)])
    order = torch.sort(ids, dim=0, descending=False)[1].tolist()
    return torch.cat([halves[i] for i in order], dim=seq_dim)


def sync_output(output: torch.Tensor) -> torch.Tensor:
    mpu = state.mpu
    cp_size, cp_rank = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
    if cp_size == 1:
        return output
    import torch.distributed as dist
    output_list = [torch.empty_like(output) for _ in range(cp_size)]
    dist.all_gather(output_list, output.contiguous(), group=mpu.get_context_parallel_group())
    output_list[cp_rank] = output
    return unzigzag_gathered(output_list, cp_size, 1)
