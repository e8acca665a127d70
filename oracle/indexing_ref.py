"""CPU oracle for the integer / data-movement side of the attention hot path.
TEST INFRASTRUCTURE ONLY: nothing in the shipped package imports this module.

Plain numpy (and Python loops for the tiny cases) restating, line by line, the
reference's indexing arithmetic.  Every function cites the reference lines it
follows.  Pinned by ``tests/golden/indexing.npz`` (outputs of the reference's
own functions, produced by tests/golden/make_golden.py).  All results here are
exact integers / exact copies, so parity is bit-exact.
"""
from __future__ import annotations

import numpy as np


# ---- zig-zag context-parallel partition ----------------------------------
def zigzag_index(seq_len: int, cp_size: int, cp_rank: int) -> np.ndarray:
    """Global token ids owned by ``cp_rank``: chunks (r, 2CP-1-r) of 2*CP equal
    chunks, concatenated in that order.  lcvlm_modellink/training/utils.py:273
    (``calibration_index``) and :352-361 (the tensor slicing)."""
    assert seq_len % (2 * cp_size) == 0
    c = seq_len // (2 * cp_size)
    a, b = cp_rank, 2 * cp_size - 1 - cp_rank
    return np.concatenate([np.arange(a * c, (a + 1) * c), np.arange(b * c, (b + 1) * c)]).astype(np.int64)


def zigzag_slice(val: np.ndarray, cp_size: int, cp_rank: int, seq_dim: int = 1) -> np.ndarray:
    """training/utils.py:351-362: view [.., 2CP, s/2CP, ..], index_select
    [r, 2CP-1-r], flatten back.  Same for the RoPE table with seq_dim=0
    (rotary_pos_embedding.py:36-47)."""
    idx = zigzag_index(val.shape[seq_dim], cp_size, cp_rank)
    return np.take(val, idx, axis=seq_dim)


def unzigzag_concat(per_rank: list, cp_size: int, seq_dim: int = 1) -> np.ndarray:
    """tasks/inference/text_generation/utils.py:557-581 (``sync_output``):
    all_gather over CP, split every rank's tensor in two along the sequence,
    order the 2*CP halves by their chunk id, concatenate."""
    halves = []
    ids = []
    for r, x in enumerate(per_rank):
        n = x.shape[seq_dim] // 2
        halves += [np.take(x, np.arange(0, n), axis=seq_dim), np.take(x, np.arange(n, 2 * n), axis=seq_dim)]
        ids += [r, 2 * cp_size - 1 - r]
    order = np.argsort(np.asarray(ids), kind="stable")
    return np.concatenate([halves[i] for i in order], axis=seq_dim)


# ---- index_of_a_in_b and the image-index remap under CP -------------------
def index_of_a_in_b(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """training/utils.py:539-542.  For every element of ``a`` (all present in
    ``b``, values of ``b`` distinct) the position of that value in ``b``."""
    pos = {int(v): i for i, v in enumerate(b.tolist())}
    return np.asarray([pos[int(x)] for x in a.tolist()], dtype=np.int64)


def cp_image_remap(external_indices: np.ndarray, seq_length: int, cp_size: int, cp_rank: int):
    """training/utils.py:271-281 and :314-333.

    ``external_indices`` is [2, n_img, n_tok] int64: row 0 = batch index, row 1 =
    global sequence position of every visual token.  Returns
    (selected_images bool [n_img], src_indices [2,n], tgt_indices [2,n]) or
    (None, None, None) when no visual token falls on this rank (the reference
    then leaves the batch without src/tgt indices, :317)."""
    calib = zigzag_index(seq_length, cp_size, cp_rank)
    ind_b, ind_s = external_indices[0], external_indices[1]
    mask = np.isin(ind_s, calib)
    if not mask.any():
        return None, None, None
    selected = mask.any(axis=1)
    n_sel = int(selected.sum())
    n_tok = ind_s.shape[1]
    src_b = np.repeat(np.arange(n_sel, dtype=np.int64)[:, None], n_tok, axis=1)
    src_s = np.repeat(np.arange(n_tok, dtype=np.int64)[None, :], n_sel, axis=0)
    msel = mask[selected]
    src_b, src_s = src_b[msel], src_s[msel]
    tgt_b = ind_b[mask]
    tgt_s = index_of_a_in_b(ind_s[mask], calib)
    return selected, np.stack([src_b, src_s]), np.stack([tgt_b, tgt_s])


# ---- sample packing -> actual_seq_len (cu_seqlens) -------------------------
def pack_samples(sample_lengths, image_positions, max_len: int, shift_token: bool = False):
    """cognitron_vl/data/dataset_qwen2.py:132-166 (``add_ret``) followed by
    :188-227 (``process_ret``), integer side only.

    ``sample_lengths[i]`` tokens per sample; ``image_positions[i]`` = int64
    array [n_img_i, n_tok] of visual-token positions inside sample i (may be
    empty).  Returns actual_seq_len (cumulative END offsets, last := max_len),
    position_ids (restart per sample, padded with the last id), and
    image_indices [2, n_img, n_tok] with row 0 zero and row 1 shifted by the running
    length (:147) -- but, exactly as the reference does, ONLY when the pack already
    holds images: the first image-bearing sample is stored unshifted (:151-153)."""
    actual = []
    pos_ids = []
    img = []
    total = 0
    for L, ip in zip(sample_lengths, image_positions):
        if ip is not None and len(ip) > 0:
            img.append(np.asarray(ip, dtype=np.int64) + (total if img else 0))   # :142-153
        pos_ids += list(range(L))                                       # :162
        actual.append(total + L)                                        # :166
        total += L
    if shift_token:                                                     # :188-193
        total -= 1
        actual[-1] -= 1
        pos_ids = pos_ids[:-1]
    if max_len > total:                                                 # :204-213
        pos_ids += pos_ids[-1:] * (max_len - total)
    pos_ids = pos_ids[:max_len]                                         # :221
    actual[-1] = max_len                                                # :209,219
    if img:
        s = np.concatenate(img, axis=0)
        image_indices = np.stack([np.zeros_like(s), s])
    else:
        image_indices = np.zeros((2, 0, 0), dtype=np.int64)
    return (np.asarray(actual, dtype=np.int64), np.asarray(pos_ids, dtype=np.int64), image_indices)


def collate_actual_seq_len(per_row: list) -> np.ndarray:
    """lcvlm_modellink/legacy/data/data_samplers.py:45-48: row i of the
    micro-batch is offset by i * seq_len (seq_len = last entry of row 0)."""
    seq_len = int(per_row[0][-1])
    return np.concatenate([np.asarray(x, dtype=np.int64) + i * seq_len for i, x in enumerate(per_row)])


def collate_image_indices(per_row: list) -> np.ndarray:
    """data_samplers.py:33-42: write the micro-batch row number into
    image_indices[0] of every row that has images (rows without images still
    advance the counter) and concatenate along dim 1."""
    out = []
    for i, x in enumerate(per_row):
        if x is None:
            continue
        y = np.array(x, dtype=np.int64, copy=True)
        y[0] = i
        out.append(y)
    return np.concatenate(out, axis=1)


def cu_seqlens_from_actual(actual_seq_len) -> np.ndarray:
    """[0] + actual_seq_len as int32: what a flash-style kernel consumes
    (the reference passes ``actual_seq_len`` itself as actual_seq_qlen,
    dot_product_attention.py:363-364)."""
    return np.concatenate([[0], np.asarray(actual_seq_len, dtype=np.int64)]).astype(np.int32)


def image_token_layout(start_len: int, n_frames: int, tokens_per_frame: int = 256):
    """dataset_qwen2.py:677-697 / tasks/inference/text_generation/module.py:785-803:
    every frame contributes [START] + 256 x CONTEXT + [END]; the visual-token
    positions of frame j are arange(len+1, len+1+256) at the time of insertion.
    Returns (positions [n_frames, 256], total length after the frames)."""
    pos = []
    cur = start_len
    for _ in range(n_frames):
        cur += 1
        pos.append(np.arange(cur, cur + tokens_per_frame, dtype=np.int64))
        cur += tokens_per_frame + 1
    return (np.stack(pos) if pos else np.zeros((0, tokens_per_frame), np.int64)), cur


# ---- vision token compression and feature scatter --------------------------
def pixel_shuffle(x: np.ndarray, scale_factor: float = 0.5) -> np.ndarray:
    """lcvlm_modellink/pretrain_lcvlm.py:546-556.  [N,W,H,C] ->
    [N, W*s, H*s, C/s^2] through view / permute(0,2,1,3) / view / permute(0,2,1,3)."""
    n, w, h, c = x.shape
    x = x.reshape(n, w, int(h * scale_factor), int(c / scale_factor))
    x = np.ascontiguousarray(x.transpose(0, 2, 1, 3))
    x = x.reshape(n, int(h * scale_factor), int(w * scale_factor), int(c / (scale_factor * scale_factor)))
    return np.ascontiguousarray(x.transpose(0, 2, 1, 3))


def pixel_shuffle_v2(x: np.ndarray, scale_stride: int = 2) -> np.ndarray:
    """lcvlm_modellink/pretrain_lcvlm.py:558-570.  [N,W,H,C] (W == H) -> [N, ceil(W/s)*ceil(H/s), s*s*C]: zero padding on
    the right / bottom up to a multiple of s, then every s x s patch becomes one token, its tokens in (row, column) order."""
    n, w, h, c = x.shape
    assert w == h
    pl = (scale_stride - (h % scale_stride)) % scale_stride
    x = np.pad(x, ((0, 0), (0, pl), (0, pl), (0, 0)))
    w, h = w + pl, h + pl
    x = x.reshape(n, w // scale_stride, scale_stride, h // scale_stride, scale_stride, c)
    x = x.transpose(0, 1, 3, 2, 4, 5)
    return np.ascontiguousarray(x).reshape(n, -1, scale_stride * scale_stride * c)


def vit_tokens_to_llm_tokens(vit_output: np.ndarray, scale_factor: float = 0.5) -> np.ndarray:
    """pretrain_lcvlm.py:463-470: drop the class token, reshape to a square
    grid, pixel-shuffle, flatten to [N, 256, 4096]."""
    x = vit_output[:, 1:, :]
    h = w = int(x.shape[1] ** 0.5)
    x = x.reshape(x.shape[0], h, w, -1)
    x = pixel_shuffle(x, scale_factor)
    return x.reshape(x.shape[0], -1, x.shape[-1])


def scatter_features(word_embeddings: np.ndarray, features: np.ndarray, indices=None,
                     src_indices=None, tgt_indices=None, pre_len=None) -> np.ndarray:
    """language_model_embedding.py:114-134.  ``word_embeddings`` [b,s,h];
    with ``indices`` [2,n_img,n_tok]: rows (b_idx, s_idx) := features.view(-1,h) (:123);
    with ``pre_len``: rows [:, pre_len:pre_len+n] := features (:126);
    with src/tgt: rows tgt := features[src_b, src_s] (:131); with none of them the values stay (:134 adds 0).
    Later writes win on duplicates (index_put without accumulate)."""
    out = word_embeddings.copy()
    if indices is not None:
        b = indices[0].reshape(-1)
        s = indices[1].reshape(-1)
        f = features.reshape(-1, features.shape[-1])
        for i in range(b.shape[0]):
            out[b[i], s[i]] = f[i]
    elif pre_len is not None:
        out[:, pre_len:pre_len + features.shape[1]] = features
    elif src_indices is not None:
        for i in range(src_indices.shape[1]):
            out[tgt_indices[0, i], tgt_indices[1, i]] = features[src_indices[0, i], src_indices[1, i]]
    return out


def scatter_features_grad(dout: np.ndarray, features_shape, indices=None, src_indices=None, tgt_indices=None,
                          pre_len=None):
    """Gradients torch autograd derives for the assignments above, given ``dout`` [b,s,h] (the gradient of the
    scattered embeddings): (d_word_embeddings, d_features).  An overwritten row passes nothing back to the word
    embedding; a feature row receives the gradient of the row it was written to (summed if written twice); with no
    index at all ``features.mean() * 0`` hands every feature element a zero."""
    dwe = dout.copy()
    df = np.zeros(features_shape, dtype=dout.dtype)
    h = dout.shape[-1]
    if indices is not None:
        b = indices[0].reshape(-1)
        s = indices[1].reshape(-1)
        f = df.reshape(-1, h)
        for i in range(b.shape[0]):
            f[i] += dout[b[i], s[i]]
            dwe[b[i], s[i]] = 0
    elif pre_len is not None:
        n = features_shape[1]
        df[...] = dout[:, pre_len:pre_len + n]
        dwe[:, pre_len:pre_len + n] = 0
    elif src_indices is not None:
        for i in range(src_indices.shape[1]):
            df[src_indices[0, i], src_indices[1, i]] += dout[tgt_indices[0, i], tgt_indices[1, i]]
            dwe[tgt_indices[0, i], tgt_indices[1, i]] = 0
    return dwe, df


def rope_position_gather(table: np.ndarray, position_ids: np.ndarray) -> np.ndarray:
    """rotary_pos_embedding.py:114-117: table [S,1,1,d], position_ids [s,b] ->
    emb[position_ids.view(-1)].squeeze(1).reshape(s, b, 1, d)."""
    s, b = position_ids.shape
    return table[position_ids.reshape(-1)].reshape(s, b, 1, table.shape[-1])
