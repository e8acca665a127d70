"""Logits for the positions that are needed only: the ``--logit-mask`` path of the reference.

At long context the vocabulary projection of every position is the largest tensor of the step
(1M x 152K logits); the reference passes a boolean ``logit_mask`` [b, s] down to its output layer, which
keeps the selected rows of the hidden states only (``torch.masked_select`` form at
lcvlm_modellink/core/models/multimodal/gpt_vl_model.py:306-312; labels selected the same way at :346-357),
and the generation loop builds the mask for the one position it samples from
(tasks/inference/text_generation/utils.py:357-366, then :378-398 with ``sync_output`` under context
parallelism).  Here the row selection is a HIP row gather (differentiable: the gradient of the selected rows is
scattered back, every other position gets zero, as ``masked_select`` does); the projection itself is a library GEMM.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


def logit_mask_for_context(input_tokens: torch.Tensor, context_length: int, cp_size: int) -> torch.Tensor:
    """tasks/inference/text_generation/utils.py:358-363: one True per row, at the position whose logits are
    sampled; under context parallelism the column is taken modulo the local length."""
    logit_mask = torch.zeros_like(input_tokens).bool()
    if cp_size == 1:
        logit_mask[:, context_length - 1] = 1
    else:
        logit_mask[:, context_length % input_tokens.size(1) - 1] = 1
    return logit_mask


def select_hidden_states(hidden_states: torch.Tensor, logit_mask: torch.Tensor) -> torch.Tensor:
    """gpt_vl_model.py:306-312: hidden_states [s, b, c], logit_mask [b, s] with b == 1 -> [n_selected, b, c],
    rows in sequence order."""
    s, b, c = hidden_states.shape
    assert b == 1
    index = torch.nonzero(logit_mask.transpose(0, 1).reshape(-1), as_tuple=False).reshape(-1)
    return ops.gather_rows(hidden_states.reshape(s * b, c), index, unique_index=True).reshape(-1, b, c)


def select_labels(labels: torch.Tensor, logit_mask: torch.Tensor) -> torch.Tensor:
    """gpt_vl_model.py:346-357: labels [b, s] -> [b, n_selected]."""
    b = logit_mask.size(0)
    assert b == 1
    index = torch.nonzero(logit_mask.reshape(-1), as_tuple=False).reshape(-1)
    return ops.gather_rows(labels.reshape(-1, 1), index).reshape(b, -1)


def masked_logits(hidden_states: torch.Tensor, weight: torch.Tensor, logit_mask: torch.Tensor | None) -> torch.Tensor:
    """The output layer with a logit mask (gpt_vl_model.py:321): [s, b, c] x [vocab, c] -> [n_selected, b, vocab]
    (every position when the mask is None)."""
    if logit_mask is not None:
        hidden_states = select_hidden_states(hidden_states, logit_mask)
    return F.linear(hidden_states, weight)
