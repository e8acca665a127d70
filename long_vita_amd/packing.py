"""Sample packing: the integer bookkeeping that produces ``actual_seq_len``
(cu_seqlens), per-segment position ids and image-token indices.

Mirrors cognitron_vl/data/dataset_qwen2.py:132-166 (``add_ret``), :168-254
(``process_ret``, integer side), :596-616 / :676-697 (image / video frame token
layout) and lcvlm_modellink/legacy/data/data_samplers.py:16-50 (``collate_fn``).
Tokenisation, chat templates and image decoding are outside this path.
"""
from __future__ import annotations

import torch

IGNORE_TOKEN_ID = -100


def frame_token_span(cur_len: int, image_token_length: int = 256):
    """One image / video frame appended at length ``cur_len``: [START] + N x CONTEXT + [END].
    Returns (image_indices [2,1,N] with row 0 zero and row 1 the context positions, new length)."""
    start = cur_len + 1
    s = torch.arange(start, start + image_token_length, dtype=torch.int64).unsqueeze(0)
    return torch.stack([torch.zeros_like(s), s], dim=0), cur_len + image_token_length + 2


class PackedSample:
    """Accumulates samples into one row of ``max_len`` tokens (``add_ret``), then pads /
    truncates and emits tensors (``process_ret``)."""

    def __init__(self, max_len: int, pad_token_id: int = 0, reset_position_ids: bool = True,
                 reset_attention_mask: bool = True, shift_token: bool = False, variable_length: bool = False):
        self.max_len = max_len
        self.pad_token_id = pad_token_id
        self.reset_position_ids = reset_position_ids
        self.reset_attention_mask = reset_attention_mask
        self.shift_token = shift_token
        self.variable_length = variable_length
        self.tokens, self.labels, self.position_ids, self.actual_seq_len = [], [], [], []
        self.image_indices = None
        self.n_images = 0

    def __len__(self):
        return len(self.tokens)

    def add(self, input_ids, labels=None, image_indices=None):
        """``image_indices`` [2, n_img, n_tok] holds positions relative to this sample."""
        cur = len(input_ids)
        total = len(self.tokens)
        if image_indices is not None and image_indices.shape[1] > 0:
            ii = image_indices.clone()
            if self.image_indices is not None:
                ii[1, :, :] += total                                  # dataset_qwen2.py:147
                self.image_indices = torch.cat([self.image_indices, ii], dim=1)
            else:
                self.image_indices = ii                                # first images: sample starts at 0 (:152-153)
            self.n_images += ii.shape[1]
        self.position_ids += list(range(cur))                          # :162
        self.tokens += list(input_ids)
        self.labels += list(labels) if labels is not None else [IGNORE_TOKEN_ID] * cur
        self.actual_seq_len += [total + cur]                           # :166

    def finalize(self):
        tokens, labels = list(self.tokens), list(self.labels)
        pos, actual = list(self.position_ids), list(self.actual_seq_len)
        if not self.reset_position_ids:
            pos = list(range(len(tokens)))                             # :178
        if self.shift_token:                                           # :188-193
            tokens, labels, pos = tokens[:-1], labels[1:], pos[:-1]
            actual[-1] -= 1
        if not self.variable_length and self.max_len > len(tokens):    # :204-213
            pad = self.max_len - len(tokens)
            tokens += [self.pad_token_id] * pad
            labels += [IGNORE_TOKEN_ID] * pad
            actual[-1] = self.max_len
            pos += pos[-1:] * pad
        tokens, labels, pos = tokens[:self.max_len], labels[:self.max_len], pos[:self.max_len]
        actual[-1] = self.max_len                                      # :219
        out = {
            "tokens": torch.tensor(tokens, dtype=torch.int64),
            "labels": torch.tensor(labels, dtype=torch.int64),
            "position_ids": torch.tensor(pos, dtype=torch.int64),
        }
        out["loss_mask"] = torch.where(out["labels"] == IGNORE_TOKEN_ID, 0, 1).to(torch.float32)   # :242-244
        if self.reset_position_ids or self.reset_attention_mask:       # popped otherwise (:246-247)
            out["actual_seq_len"] = torch.tensor(actual, dtype=torch.int64)
        if self.image_indices is not None:
            out["image_indices"] = self.image_indices
        out["input_ids"] = out["tokens"]
        return out


def collate_fn(batch):
    """Micro-batch collation of packed rows (data_samplers.py:16-50): tensors are stacked,
    ``images`` concatenated on dim 0, ``image_indices`` on dim 1 with the row number written
    into index row 0, and ``actual_seq_len`` of row i offset by i * seq_len."""
    keys = list(set().union(*[set(x.keys()) for x in batch]))
    special = [k for k in keys if "images" in k or k == "image_indices"]
    out = {}
    for k in keys:
        if k in special or k == "actual_seq_len":
            continue
        out[k] = torch.stack([x[k] for x in batch], 0)
    for k in special:
        parts = []
        for row, x in enumerate(batch):
            if k in x:
                t = x[k]
                if k == "image_indices":
                    t = t.clone()
                    t[0] = row
                parts.append(t)
        out[k] = torch.cat(parts, dim=1 if k == "image_indices" else 0)
    if "actual_seq_len" in batch[0]:
        seq_len = batch[0]["actual_seq_len"][-1]
        out["actual_seq_len"] = torch.cat([x["actual_seq_len"] + i * seq_len for i, x in enumerate(batch)])
    return out
