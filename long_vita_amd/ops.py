"""Tensor-level wrappers over the C ABI (include/lvattn.h).

PyTorch is used here for device memory, streams and autograd bookkeeping only:
every arithmetic step runs in the HIP kernels of liblvattn.so.  Tensors are
passed as raw pointers + element strides; nothing is copied unless a tensor's
last dimension is not contiguous.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Sequence

import numpy as np
import torch

from . import _lib
from ._lib import LV_BWD_BLOCK_N, LV_DESC_INTS, LvAttnError, check

_I32P = C.POINTER(C.c_int32)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: torch.Tensor | None) -> int:
    return 0 if t is None else t.data_ptr()


def _require_gpu(*tensors):
    """Every tensor on the GPU, and on the CURRENT device: the library launches on the current device's current stream
    with raw pointers, so a tensor of another GPU would be addressed from the wrong one."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise LvAttnError("liblvattn kernels need tensors on the GPU (there is no CPU fallback)")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise LvAttnError(f"tensor on cuda:{t.device.index} while the current device is cuda:{cur}: the library launches on "
                              "the current device (torch.cuda.set_device / torch.cuda.device first)")


def _last_contig(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


# ---------------------------------------------------------------------------
# block descriptors
# ---------------------------------------------------------------------------
def cu_seqlens_from_actual_seq_len(actual_seq_len: Sequence[int]) -> list:
    """[0] + cumulative end offsets.  The reference hands ``actual_seq_len`` (end
    offsets without the leading 0) to the kernel as actual_seq_qlen/kvlen
    (lcvlm_modellink/core/transformer/dot_product_attention.py:333,363-364)."""
    return [0] + [int(x) for x in actual_seq_len]


def build_blocks_host(cu_seqlens: Sequence[int], causal: bool, backward: bool = False) -> np.ndarray:
    """Descriptors [n, 8] int32 (host) for packed segments, heaviest first."""
    cu = np.ascontiguousarray(np.asarray(cu_seqlens, dtype=np.int32))
    n_seg = cu.shape[0] - 1
    if n_seg < 0:
        raise LvAttnError("cu_seqlens must hold at least one entry")
    blk = LV_BWD_BLOCK_N if backward else _lib.lib().lvattn_fwd_block_m()
    lens = np.diff(cu.astype(np.int64)) if n_seg else np.zeros(0, np.int64)
    max_blocks = int(np.sum((np.maximum(lens, 0) + blk - 1) // blk)) + 1
    out = np.zeros((max_blocks, LV_DESC_INTS), dtype=np.int32)
    fn = _lib.lib().lv_build_bwd_blocks if backward else _lib.lib().lv_build_fwd_blocks
    n = check(fn(cu.ctypes.data_as(_I32P), n_seg, int(causal), out.ctypes.data_as(_I32P), max_blocks), "lv_build_blocks")
    return out[:n]


def build_rect_blocks_host(q_row0: int, q_len: int, k_row0: int, k_len: int, causal: bool, shift: int = 0,
                           backward: bool = False) -> np.ndarray:
    """Descriptors for one rectangle of the score matrix (a ring step)."""
    blk = LV_BWD_BLOCK_N if backward else _lib.lib().lvattn_fwd_block_m()
    span = k_len if backward else q_len
    max_blocks = (span + blk - 1) // blk + 1
    out = np.zeros((max_blocks, LV_DESC_INTS), dtype=np.int32)
    fn = _lib.lib().lv_build_bwd_blocks_rect if backward else _lib.lib().lv_build_fwd_blocks_rect
    n = check(fn(q_row0, q_len, k_row0, k_len, int(causal), shift, out.ctypes.data_as(_I32P), max_blocks), "lv_build_blocks_rect")
    return out[:n]


class _BlockCache:
    """Device copies of descriptor tables, keyed by (segments, causal, direction, device)."""

    def __init__(self, capacity: int = 1024):   # a packed-segment ring step set is 4 pairs x CP steps x 3 tables
        self.capacity = capacity
        self.store: OrderedDict = OrderedDict()

    def get(self, key, make):
        hit = self.store.get(key)
        if hit is not None:
            self.store.move_to_end(key)
            return hit
        val = make()
        self.store[key] = val
        if len(self.store) > self.capacity:
            self.store.popitem(last=False)
        return val


_blocks = _BlockCache()


def _block_rows(backward: bool) -> int:
    """Rows per work item of the kernel that will run: part of every cache key, since the forward value follows the
    forward kernel selected at run time (lvattn_select_fwd_kernel)."""
    return LV_BWD_BLOCK_N if backward else _lib.lib().lvattn_fwd_block_m()


def _to_device(desc_host: np.ndarray, device) -> torch.Tensor:
    """Device copy of a descriptor table.  It carries the table's mean key count per work item (``lv_mean_keys``), which
    the forward launch hands to the library: the choice between the two head-dim-128 forward kernels follows the length
    of a work item's key sweep, and the library cannot read the table once it is in device memory."""
    desc_host = np.ascontiguousarray(desc_host)
    t = torch.from_numpy(desc_host).to(device)
    t.lv_mean_keys = int(_lib.lib().lv_mean_keys_per_item(desc_host.ctypes.data_as(_I32P), desc_host.shape[0]))
    return t


def blocks_on_device(cu_seqlens: Sequence[int], causal: bool, backward: bool, device) -> torch.Tensor:
    key = ("seg", tuple(int(x) for x in cu_seqlens), bool(causal), bool(backward), str(device), _block_rows(backward))
    return _blocks.get(key, lambda: _to_device(build_blocks_host(cu_seqlens, causal, backward), device))


def rect_blocks_on_device(q_row0, q_len, k_row0, k_len, causal, shift, backward, device) -> torch.Tensor:
    key = ("rect", q_row0, q_len, k_row0, k_len, bool(causal), shift, bool(backward), str(device), _block_rows(backward))
    return _blocks.get(key, lambda: _to_device(build_rect_blocks_host(q_row0, q_len, k_row0, k_len, causal, shift, backward), device))


def band_blocks_on_device(cu_seqlens: Sequence[int], band: int, backward: bool, device) -> torch.Tensor:
    """Descriptors of packed segments under a band (a query sees itself and the ``band`` keys before it): launch them with
    causal = 2."""
    key = ("band", tuple(int(x) for x in cu_seqlens), int(band), bool(backward), str(device), _block_rows(backward))

    def build():
        cu = np.ascontiguousarray(np.asarray(cu_seqlens, dtype=np.int32))
        n_seg = cu.shape[0] - 1
        blk = LV_BWD_BLOCK_N if backward else _lib.lib().lvattn_fwd_block_m()
        lens = np.diff(cu.astype(np.int64)) if n_seg else np.zeros(0, np.int64)
        max_blocks = int(np.sum((np.maximum(lens, 0) + blk - 1) // blk)) + 1
        out = np.zeros((max_blocks, LV_DESC_INTS), dtype=np.int32)
        fn = _lib.lib().lv_build_bwd_blocks_band if backward else _lib.lib().lv_build_fwd_blocks_band
        n = check(fn(cu.ctypes.data_as(_I32P), n_seg, int(band), out.ctypes.data_as(_I32P), max_blocks), "lv_build_blocks_band")
        return _to_device(out[:n], device)
    return _blocks.get(key, build)


def rect_band_blocks_on_device(q_row0, q_len, k_row0, k_len, shift, band, device) -> torch.Tensor:
    """Forward descriptors of one rectangle under a band (the prompt pass over a KV cache)."""
    key = ("rect_band", q_row0, q_len, k_row0, k_len, shift, int(band), str(device), _block_rows(False))

    def build():
        blk = _lib.lib().lvattn_fwd_block_m()
        max_blocks = (q_len + blk - 1) // blk + 1
        out = np.zeros((max_blocks, LV_DESC_INTS), dtype=np.int32)
        n = check(_lib.lib().lv_build_fwd_blocks_rect_band(q_row0, q_len, k_row0, k_len, shift, int(band), out.ctypes.data_as(_I32P), max_blocks),
                  "lv_build_fwd_blocks_rect_band")
        return _to_device(out[:n], device)
    return _blocks.get(key, build)


def rects_blocks_on_device(rects, backward: bool, device) -> torch.Tensor:
    """Descriptors for several rectangles of one launch: ``rects`` = tuples (q_row0, q_len, k_row0, k_len, shift), all
    causal with their own shift (key j of the rectangle visible to its row i iff j <= i + shift; a shift >= k_len makes
    the rectangle dense).  The rectangles of one launch must not share query rows (forward, dQ) or key rows (dK/dV).
    Heaviest work items first, as for a single rectangle."""
    rects = tuple(tuple(int(x) for x in r) for r in rects)
    key = ("rects", rects, bool(backward), str(device), _block_rows(backward))

    def build():
        parts = [build_rect_blocks_host(q0, ql, k0, kl, True, sh, backward) for q0, ql, k0, kl, sh in rects]
        d = np.concatenate(parts) if parts else np.zeros((0, LV_DESC_INTS), dtype=np.int32)
        order = np.argsort(-(d[:, 1].astype(np.int64) * d[:, 3].astype(np.int64)), kind="stable")
        return _to_device(d[order], device)
    return _blocks.get(key, build)


# ---------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------
def _check_qkv(q, k, v):
    _require_gpu(q, k, v)
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise LvAttnError("q, k, v must be bfloat16")
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3:
        raise LvAttnError("q, k, v must be [rows, heads, head_dim]")
    if k.shape != v.shape or q.shape[2] != k.shape[2]:
        raise LvAttnError(f"shape mismatch q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")


def attn_fwd_launch(q, k, v, desc: torch.Tensor, scale: float, causal: bool, out=None, lse=None):
    """q [Rq,Hq,D], k/v [Rk,Hkv,D] bf16 (any row/head stride), desc int32 [n,8] on
    the device.  Returns (out [Rq,Hq,D] bf16, lse [Hq,Rq] fp32).  Rows no descriptor
    covers are left untouched in ``out``/``lse`` when those are passed in."""
    q, k, v = _last_contig(q), _last_contig(k), _last_contig(v)
    _check_qkv(q, k, v)
    rq, hq, d = q.shape
    rk, hkv, _ = k.shape
    if out is None:
        out = torch.empty((rq, hq, d), dtype=torch.bfloat16, device=q.device)
    if lse is None:
        lse = torch.empty((hq, rq), dtype=torch.float32, device=q.device)
    check(_lib.lib().lvattn_fwd_hinted(
        _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _ptr(desc), desc.shape[0],
        rq, rk, hq, hkv, d,
        q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
        out.stride(0), out.stride(1), lse.stride(0), float(scale), int(causal), int(getattr(desc, "lv_mean_keys", -1)),
        _stream()), "lvattn_fwd")
    return out, lse


def attn_fwd_dense_launch(q, k, v, scale: float, causal: bool, out=None, lse=None):
    """attn_fwd_launch for ONE dense segment (all of q's rows against all of k's, every head) without descriptors:
    the library numbers the work items itself and sizes its workgroups to the row count (lvattn_fwd_dense).  What the
    vision tower's attention (dot_product_attention.py:314-325) runs through."""
    q, k, v = _last_contig(q), _last_contig(k), _last_contig(v)
    _check_qkv(q, k, v)
    rq, hq, d = q.shape
    rk, hkv, _ = k.shape
    if out is None:
        out = torch.empty((rq, hq, d), dtype=torch.bfloat16, device=q.device)
    if lse is None:
        lse = torch.empty((hq, rq), dtype=torch.float32, device=q.device)
    check(_lib.lib().lvattn_fwd_dense(
        _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), rq, rk, hq, hkv, d,
        q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
        out.stride(0), out.stride(1), lse.stride(0), float(scale), int(causal), _stream()), "lvattn_fwd_dense")
    return out, lse


def decode_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, n_split: int = 0, return_lse: bool = False,
                     out: torch.Tensor | None = None, lse: torch.Tensor | None = None, library_scratch: bool = False):
    """One new token per sequence against a KV cache: q [b, hq, d], k / v [b, k_len, hkv, d] bf16 (any batch / row /
    head strides, last dimension contiguous) -> out [b, hq, d] (and lse [b, hq] fp32).  With ``library_scratch`` the
    partial results live in the library's own fixed buffer (lvattn_decode_noalloc): given ``out`` (and ``lse``) the
    call then allocates nothing, at the price of a size limit and of calls on one GPU having to be serialised."""
    _require_gpu(q, k, v)
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise LvAttnError("q, k, v must be bfloat16")
    if q.dim() != 3 or k.dim() != 4 or v.shape != k.shape or k.shape[0] != q.shape[0] or k.shape[3] != q.shape[2]:
        raise LvAttnError("decode_attention expects q [b, hq, d] and k, v [b, k_len, hkv, d]")
    q, k, v = _last_contig(q), _last_contig(k), _last_contig(v)
    b, hq, d = q.shape
    _, k_len, hkv, _ = k.shape
    if out is None:
        out = torch.empty((b, hq, d), dtype=torch.bfloat16, device=q.device)
    elif out.shape != (b, hq, d) or out.dtype != torch.bfloat16 or out.device != q.device or out.stride(2) != 1:
        raise LvAttnError("decode_attention: out must be bf16 [b, hq, d] on q's device with a contiguous last dimension")
    if lse is None:
        lse = torch.empty((b, hq), dtype=torch.float32, device=q.device) if return_lse else None
    elif lse.shape != (b, hq) or lse.dtype != torch.float32 or lse.device != q.device or not lse.is_contiguous():
        raise LvAttnError("decode_attention: lse must be contiguous fp32 [b, hq] on q's device")
    if library_scratch:
        check(_lib.lib().lvattn_decode_noalloc(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), b, k_len, hq, hkv, d,
                                               q.stride(0), q.stride(1), k.stride(0), k.stride(1), k.stride(2),
                                               v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1),
                                               float(scale), int(n_split), _stream()), "lvattn_decode_noalloc")
        return (out, lse) if return_lse else out
    nbytes = _lib.lib().lvattn_decode_workspace_bytes(b, hq, hkv, d)
    ws = torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=q.device)
    check(_lib.lib().lvattn_decode(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _ptr(ws), nbytes, b, k_len, hq, hkv, d,
                                   q.stride(0), q.stride(1), k.stride(0), k.stride(1), k.stride(2),
                                   v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1),
                                   float(scale), int(n_split), _stream()), "lvattn_decode")
    return (out, lse) if return_lse else out


def attn_bwd_launch(q, k, v, out, dout, lse, desc_q: torch.Tensor, desc_k: torch.Tensor, scale: float, causal: bool,
                    dq=None, dk=None, dv=None, delta=None, accum_f32: bool | int = False):
    """Backward of ``attn_fwd_launch``.  ``desc_q`` are the forward's query-block descriptors
    (dQ kernel), ``desc_k`` the key-block descriptors (dK/dV kernel).  Returns (dq, dk, dv, delta):
    bf16 tensors that are written or, with ``accum_f32``, fp32 buffers that are added to (True / 1) or written (2).
    Rows no descriptor covers are left untouched (buffers allocated here start at zero)."""
    q, k, v, dout, out = (_last_contig(t) for t in (q, k, v, dout, out))
    _check_qkv(q, k, v)
    rq, hq, d = q.shape
    rk, hkv, _ = k.shape
    L = _lib.lib()
    if delta is None:
        delta = torch.empty((hq, lse.stride(0)), dtype=torch.float32, device=q.device)
        check(L.lvattn_bwd_delta(_ptr(out), _ptr(dout), _ptr(delta), rq, hq, d, out.stride(0), out.stride(1),
                                 dout.stride(0), dout.stride(1), delta.stride(0), _stream()), "lvattn_bwd_delta")
    dt = torch.float32 if accum_f32 else torch.bfloat16
    if dq is None:
        dq = torch.zeros((rq, hq, d), dtype=dt, device=q.device)
    if dk is None:
        dk = torch.zeros((rk, hkv, d), dtype=dt, device=q.device)
        dv = torch.zeros((rk, hkv, d), dtype=dt, device=q.device)
    check(L.lvattn_bwd(
        _ptr(q), _ptr(k), _ptr(v), _ptr(dout), _ptr(lse), _ptr(delta), _ptr(dq), _ptr(dk), _ptr(dv),
        _ptr(desc_q), desc_q.shape[0], _ptr(desc_k), desc_k.shape[0], rq, rk, hq, hkv, d,
        q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
        dout.stride(0), dout.stride(1), dq.stride(0), dq.stride(1), dk.stride(0), dk.stride(1), dv.stride(0), dv.stride(1),
        lse.stride(0), float(scale), int(causal), int(accum_f32), _stream()), "lvattn_bwd")
    return dq, dk, dv, delta


def cast_f32_to_bf16(src: torch.Tensor, dst: torch.Tensor | None = None) -> torch.Tensor:
    """src fp32 [rows, heads, d] contiguous -> bf16 (dst may be a strided view)."""
    rows, heads, d = src.shape
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(_lib.lib().lvattn_cast_f32_bf16(_ptr(src), _ptr(dst), rows, heads, d, dst.stride(0), dst.stride(1), _stream()),
          "lvattn_cast_f32_bf16")
    return dst


def merge_launch(out_acc, lse_acc, out_blk, lse_blk, row0: int, rows: int, init: bool = False):
    """In-place log-sum-exp merge of a block result into the running fp32 result; with ``init`` the block becomes
    the running result for those rows (the accumulators are not read)."""
    _, hq, d = out_acc.shape
    fn = _lib.lib().lvattn_merge_init if init else _lib.lib().lvattn_merge
    check(fn(_ptr(out_acc), _ptr(lse_acc), _ptr(out_blk), _ptr(lse_blk), row0, rows, hq, d,
                                  out_blk.stride(0), out_blk.stride(1), lse_acc.stride(0), _stream()), "lvattn_merge")


def key_pieces_rule(sq: int, sk: int, hq: int, n_cu: int) -> int:
    """How many pieces to cut the key range of a (few rows) x (long cache) rectangle into: a forward work item is 256
    query rows of one head sweeping ALL its keys, so a short query block against a long cache is a handful of very long
    work items on a chip of hundreds of CUs.  With P pieces the launch has items * P work items of 1/P the length and
    takes ceil(items * P / CUs) / P of the single sweep's time: the smallest P (pieces of at least 16,384 keys, at most
    64) that gets within 3 % of the best such figure, or 1 when even the best saves less than a fifth."""
    items = -(-sq // 256) * hq
    p_max = min(64, sk // 16384, 16384 // max(1, sq))      # the query rows are replicated once per piece: at most 16,384 rows
    if items >= n_cu or p_max < 2:
        return 1
    cost = [(-(-items * p // n_cu)) / p for p in range(1, p_max + 1)]
    best = min(cost)
    if best > 0.8:                                         # under a fifth to gain: not worth the replicated rows and the merges
        return 1
    return next(p for p, c in enumerate(cost, start=1) if c <= best * 1.03)


def _key_pieces_for(sq: int, sk: int, hq: int, device) -> int:
    return key_pieces_rule(sq, sk, hq, torch.cuda.get_device_properties(device).multi_processor_count)


def attn_fwd_rect_split_keys(q, k, v, scale: float, causal: bool, shift: int = 0, n_pieces: int | None = None):
    """One rectangle: every row of q [Rq,Hq,D] against every row of k / v [Rk,Hkv,D]; with ``causal`` key j is visible to
    row i iff j <= i + shift (the prompt pass over a KV cache: shift = Rk - Rq, dot_product_attention.py:440-451).
    Returns (out [Rq,Hq,D] bf16, lse [Hq,Rq] fp32) like ``attn_fwd_launch``.

    When the rectangle is a few query rows against a long key range (a follow-up turn of a few hundred tokens on a
    1M-token cache), its key range is cut into ``n_pieces`` (chosen from the shape when None): ONE launch computes every
    (query block, piece) pair as a work item of its own -- the query rows are replicated once per piece so that each
    pair has its own output rows -- and the pieces' (out, lse) are merged in log-sum-exp form by the ring's merge
    kernel.  Same result as the single sweep up to the bf16 rounding of the per-piece outputs."""
    q, k, v = _last_contig(q), _last_contig(k), _last_contig(v)
    _check_qkv(q, k, v)
    sq, hq, d = q.shape
    sk = k.shape[0]
    dev = q.device
    P = _key_pieces_for(sq, sk, hq, dev) if n_pieces is None else max(1, int(n_pieces))
    vis_end = min(sk, sq + shift) if causal else sk       # keys past this are visible to no row
    if P > 1 and vis_end > 0:
        L = -(-(-(-vis_end // P)) // 64) * 64
        starts = list(range(0, vis_end, L))
        P = len(starts)
    if P <= 1 or vis_end <= 0:
        desc = rect_blocks_on_device(0, sq, 0, sk, causal, shift, False, dev)
        return attn_fwd_launch(q, k, v, desc, scale, causal)
    dense = 1 << 29                                           # a shift no row / key pair can exceed: the piece is unmasked
    rects = [(p * sq, sq, k0, min(L, vis_end - k0), (shift - k0) if causal else dense) for p, k0 in enumerate(starts)]
    desc = rects_blocks_on_device(rects, False, dev)
    q_rep = q.unsqueeze(0).expand(P, sq, hq, d).reshape(P * sq, hq, d)
    out_rep, lse_rep = attn_fwd_launch(q_rep, k, v, desc, scale, True)
    out_acc = torch.empty((sq, hq, d), dtype=torch.float32, device=dev)
    lse_acc = torch.empty((hq, P * sq), dtype=torch.float32, device=dev)[:, :sq]     # the pieces' row stride
    for p in range(P):
        merge_launch(out_acc, lse_acc, out_rep[p * sq:(p + 1) * sq], lse_rep[:, p * sq:(p + 1) * sq], 0, sq, init=(p == 0))
    return cast_f32_to_bf16(out_acc), lse_acc.contiguous()


# ---------------------------------------------------------------------------
# split sweeps: filling the chip when a launch has fewer work items than compute units
# ---------------------------------------------------------------------------
class SplitTables:
    """Device tables of one split launch: ``desc`` (work items, one per (block, piece)), ``comb`` (combine entries),
    ``ws_rows`` workspace rows, ``block_rows`` (256 forward / dQ, 128 dK/dV), ``piece`` the piece length used."""

    def __init__(self, desc, comb, ws_rows, block_rows, piece):
        self.desc, self.comb, self.ws_rows, self.block_rows, self.piece = desc, comb, ws_rows, block_rows, piece


_split_override = {"fwd": None, "bwd": None}     # tests: force a piece length (0 = never split, None = the planner's choice)


def set_split_override(fwd=None, bwd=None):
    """Test hook: force the piece length of the forward / dQ sweeps (keys) and of the dK/dV sweeps (query rows);
    0 disables splitting, None gives the choice back to the planner."""
    _split_override["fwd"], _split_override["bwd"] = fwd, bwd


def _n_cu(device) -> int:
    return torch.cuda.get_device_properties(device).multi_processor_count


def plan_split(cu_seqlens, causal: bool, hq: int, hkv: int, device):
    """(piece_keys, piece_rows) for the forward / dQ and the dK/dV launch of a packed batch: 0 = leave that launch as it
    is.  The planner simulates a heaviest-first schedule of the launch's work items on the device's compute units and
    cuts the sweeps when the launch has too few items to fill them (lv_plan_fwd_split / lv_plan_bwd_split)."""
    pf = _split_override["fwd"]
    pb = _split_override["bwd"]
    if pf is not None and pb is not None:
        return int(pf), int(pb)
    # (the simulation costs the host 0.05 - 1.3 ms for batches of 3K - 32K tokens, and every layer of a step asks the same
    # question: the answer is cached like the tables)
    key = ("plan", tuple(int(x) for x in cu_seqlens), bool(causal), int(hq), int(hkv), str(device))

    def build():
        cu = np.ascontiguousarray(np.asarray(cu_seqlens, dtype=np.int32))
        n_seg = cu.shape[0] - 1
        L = _lib.lib()
        n_cu = _n_cu(device)
        return (check(L.lv_plan_fwd_split(cu.ctypes.data_as(_I32P), n_seg, int(causal), hq, n_cu), "lv_plan_fwd_split"),
                check(L.lv_plan_bwd_split(cu.ctypes.data_as(_I32P), n_seg, int(causal), hkv, hq // hkv, n_cu), "lv_plan_bwd_split"))
    planned = _blocks.get(key, build)
    return int(planned[0] if pf is None else pf), int(planned[1] if pb is None else pb)


def split_tables_on_device(cu_seqlens, causal: bool, backward: bool, piece: int, device) -> SplitTables:
    """Split tables of a packed batch (cached like the plain descriptor tables).  ``piece`` = keys per piece (forward /
    dQ) or query rows per piece (dK/dV); a huge value gives one piece per block (the launch then merely goes through
    the workspace)."""
    key = ("split", tuple(int(x) for x in cu_seqlens), bool(causal), bool(backward), int(piece), str(device))

    def build():
        import ctypes
        cu = np.ascontiguousarray(np.asarray(cu_seqlens, dtype=np.int32))
        n_seg = cu.shape[0] - 1
        fn = _lib.lib().lv_build_bwd_blocks_split if backward else _lib.lib().lv_build_fwd_blocks_split
        n_comb, ws_rows = ctypes.c_int(0), ctypes.c_int(0)
        n = check(fn(cu.ctypes.data_as(_I32P), n_seg, int(causal), int(piece), None, 0, None, 0, ctypes.byref(n_comb), ctypes.byref(ws_rows)),
                  "lv_build_blocks_split")
        desc = np.zeros((max(n, 1), LV_DESC_INTS), dtype=np.int32)
        comb = np.zeros((max(n_comb.value, 1), 4), dtype=np.int32)
        n = check(fn(cu.ctypes.data_as(_I32P), n_seg, int(causal), int(piece), desc.ctypes.data_as(_I32P), desc.shape[0],
                     comb.ctypes.data_as(_I32P), comb.shape[0], ctypes.byref(n_comb), ctypes.byref(ws_rows)), "lv_build_blocks_split")
        return SplitTables(_to_device(desc[:n], device), torch.from_numpy(comb[:n_comb.value]).to(device), int(ws_rows.value),
                           LV_BWD_BLOCK_N if backward else 256, int(piece))
    return _blocks.get(key, build)


_ONE_PIECE = 1 << 30     # a "piece length" that leaves every sweep whole


def combine_lse_launch(ws_out, ws_lse, out, lse, tables: SplitTables):
    hq, d = out.shape[1], out.shape[2]
    check(_lib.lib().lvattn_combine_lse(_ptr(ws_out), _ptr(ws_lse), _ptr(out), _ptr(lse), _ptr(tables.comb), tables.comb.shape[0],
                                        hq, d, ws_lse.stride(0), out.stride(0), out.stride(1), lse.stride(0), _stream()),
          "lvattn_combine_lse")


def combine_sum_launch(ws, out, tables: SplitTables):
    heads, d = out.shape[1], out.shape[2]
    check(_lib.lib().lvattn_combine_sum(_ptr(ws), _ptr(out), _ptr(tables.comb), tables.comb.shape[0], tables.block_rows, heads, d,
                                        out.stride(0), out.stride(1), _stream()), "lvattn_combine_sum")


def attn_fwd_split_launch(q, k, v, tables: SplitTables, scale: float, causal: bool, covered: bool = False):
    """Forward of a packed batch through split tables: one launch over every (block, piece) pair into a workspace, then
    the log-sum-exp combine.  Rows no segment covers come back as zeros / -inf (``covered``: there are none)."""
    rq, hq, d = q.shape
    dev = q.device
    ws_out = torch.empty((tables.ws_rows, hq, d), dtype=torch.bfloat16, device=dev)
    ws_lse = torch.empty((hq, tables.ws_rows), dtype=torch.float32, device=dev)
    attn_fwd_launch(q, k, v, tables.desc, scale, causal, ws_out, ws_lse)
    if covered:
        out = torch.empty((rq, hq, d), dtype=torch.bfloat16, device=dev)
        lse = torch.empty((hq, rq), dtype=torch.float32, device=dev)
    else:
        out = torch.zeros((rq, hq, d), dtype=torch.bfloat16, device=dev)
        lse = torch.full((hq, rq), float("-inf"), dtype=torch.float32, device=dev)
    combine_lse_launch(ws_out, ws_lse, out, lse, tables)
    return out, lse


_poison_workspaces = False     # test hook: fill the split path's fp32 workspaces with NaN, so that a row read without having been written shows


def _workspace(shape, dev):
    if _poison_workspaces:
        return torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
    return torch.empty(shape, dtype=torch.float32, device=dev)


def attn_bwd_split_launch(q, k, v, out, dout, lse, tq: SplitTables, tk: SplitTables, scale: float, causal: bool,
                          covered: bool = False):
    """Backward through split tables: fp32 partials of dq (forward-form table) and dk / dv (key-block table) stored to
    workspace rows of their own, summed per block in a fixed order.  Returns bf16 (dq, dk, dv)."""
    rq, hq, d = q.shape
    rk, hkv, _ = k.shape
    dev = q.device
    # (every work item writes all the rows of its piece, and the combine reads no others: no zero fill)
    dq_ws = _workspace((tq.ws_rows, hq, d), dev)
    dk_ws = _workspace((tk.ws_rows, hkv, d), dev)
    dv_ws = _workspace((tk.ws_rows, hkv, d), dev)
    attn_bwd_launch(q, k, v, out, dout, lse, tq.desc, tk.desc, scale, causal, dq_ws, dk_ws, dv_ws, accum_f32=2)
    make = torch.empty if covered else torch.zeros       # rows no segment covers get a zero gradient
    dq = make((rq, hq, d), dtype=torch.bfloat16, device=dev)
    dk = make((rk, hkv, d), dtype=torch.bfloat16, device=dev)
    dv = make((rk, hkv, d), dtype=torch.bfloat16, device=dev)
    combine_sum_launch(dq_ws, dq, tq)
    combine_sum_launch(dk_ws, dk, tk)
    combine_sum_launch(dv_ws, dv, tk)
    return dq, dk, dv


# ---------------------------------------------------------------------------
# autograd: packed variable-length attention
# ---------------------------------------------------------------------------
class _VarlenAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens, scale, causal, band=None):
        total = q.shape[0]
        ctx.band = band
        if band is not None:      # banded causal attention: its own tables, launched with causal = 2
            if not causal:
                raise LvAttnError("a band needs causal attention")
            desc = band_blocks_on_device(cu_seqlens, band, False, q.device)
            out = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
            lse = torch.full((q.shape[1], total), float("-inf"), dtype=torch.float32, device=q.device)
            out, lse = attn_fwd_launch(q, k, v, desc, scale, 2, out, lse)
            ctx.save_for_backward(q, k, v, out, lse)
            ctx.cu_seqlens, ctx.scale, ctx.causal, ctx.pieces = tuple(cu_seqlens), scale, causal, (0, 0)
            ctx.mark_non_differentiable(lse)
            return out, lse
        # few work items for the chip (small head counts, short batches): cut the sweeps into pieces
        ctx.pieces = plan_split(cu_seqlens, causal, q.shape[1], k.shape[1], q.device) if q.is_cuda else (0, 0)
        ctx.covered = len(cu_seqlens) > 1 and cu_seqlens[0] == 0 and cu_seqlens[-1] == total == k.shape[0]
        if ctx.pieces[0] > 0:
            out, lse = attn_fwd_split_launch(q, k, v, split_tables_on_device(cu_seqlens, causal, False, ctx.pieces[0], q.device),
                                             scale, causal, ctx.covered)
        elif len(cu_seqlens) == 2 and cu_seqlens[0] == 0 and cu_seqlens[1] == total == k.shape[0]:
            out, lse = attn_fwd_dense_launch(q, k, v, scale, causal)      # one segment over every row: no descriptors
        else:
            desc = blocks_on_device(cu_seqlens, causal, False, q.device)
            covered = len(cu_seqlens) > 1 and cu_seqlens[0] == 0 and cu_seqlens[-1] == total
            out = None if covered else torch.zeros(q.shape, dtype=q.dtype, device=q.device)
            lse = None if covered else torch.full((q.shape[1], total), float("-inf"), dtype=torch.float32, device=q.device)
            out, lse = attn_fwd_launch(q, k, v, desc, scale, causal, out, lse)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.cu_seqlens = tuple(cu_seqlens)
        ctx.scale = scale
        ctx.causal = causal
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        q, k, v, out, lse = ctx.saved_tensors
        if ctx.band is not None:
            desc_q = band_blocks_on_device(ctx.cu_seqlens, ctx.band, False, q.device)
            desc_k = band_blocks_on_device(ctx.cu_seqlens, ctx.band, True, q.device)
            dq, dk, dv, _ = attn_bwd_launch(q, k, v, out, dout, lse, desc_q, desc_k, ctx.scale, 2)
            return dq, dk, dv, None, None, None, None
        pf, pb = ctx.pieces
        if pf > 0 or pb > 0:      # one launch takes one kind of output buffer: when either sweep is cut, both go through workspaces
            tq = split_tables_on_device(ctx.cu_seqlens, ctx.causal, False, pf if pf > 0 else _ONE_PIECE, q.device)
            tk = split_tables_on_device(ctx.cu_seqlens, ctx.causal, True, pb if pb > 0 else _ONE_PIECE, q.device)
            dq, dk, dv = attn_bwd_split_launch(q, k, v, out, dout, lse, tq, tk, ctx.scale, ctx.causal, ctx.covered)
            return dq, dk, dv, None, None, None, None
        desc_q = blocks_on_device(ctx.cu_seqlens, ctx.causal, False, q.device)
        desc_k = blocks_on_device(ctx.cu_seqlens, ctx.causal, True, q.device)
        dq, dk, dv, _ = attn_bwd_launch(q, k, v, out, dout, lse, desc_q, desc_k, ctx.scale, ctx.causal)
        return dq, dk, dv, None, None, None, None


def varlen_attention(q, k, v, cu_seqlens: Sequence[int], scale: float | None = None, causal: bool = True, band: int | None = None):
    """Packed variable-length attention, one block per segment of ``cu_seqlens``.
    q [T,Hq,D], k/v [T,Hkv,D] bf16 -> (out [T,Hq,D] bf16, lse [Hq,T] fp32); differentiable in q, k, v.
    ``band`` (causal only): a query sees itself and the ``band`` keys before it (the fused op's band modes,
    dot_product_attention.py:403-406, 454-467); a band no segment is longer than changes nothing and is dropped."""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    cu = tuple(int(x) for x in cu_seqlens)
    if band is not None:
        longest = max((b - a for a, b in zip(cu[:-1], cu[1:])), default=0)
        band = int(band) if int(band) < longest - 1 else None
    return _VarlenAttention.apply(q, k, v, cu, float(scale), bool(causal), band)


# ---------------------------------------------------------------------------
# data movement
# ---------------------------------------------------------------------------
class _PixelShuffleTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vit_output, skip):
        ctx.skip, ctx.in_shape = skip, vit_output.shape
        return pixel_shuffle_tokens_launch(vit_output, skip)

    @staticmethod
    def backward(ctx, dy):                          # the permutation run backwards, zeros for the skipped tokens
        n, t, c = ctx.in_shape
        side = int(round(math.sqrt(t - ctx.skip)))
        g = dy.reshape(n, side // 2, side // 2, 4 * c).permute(0, 2, 1, 3).reshape(n, side // 2, side, 2 * c)
        g = g.permute(0, 2, 1, 3).reshape(n, side * side, c)
        dx = torch.zeros(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        dx[:, ctx.skip:] = g
        return dx, None


def pixel_shuffle_tokens(vit_output: torch.Tensor, skip: int = 1) -> torch.Tensor:
    """Differentiable form of ``pixel_shuffle_tokens_launch``."""
    if torch.is_grad_enabled() and vit_output.requires_grad:
        return _PixelShuffleTokens.apply(vit_output, skip)
    return pixel_shuffle_tokens_launch(vit_output, skip)


def pixel_shuffle_tokens_launch(vit_output: torch.Tensor, skip: int = 1) -> torch.Tensor:
    """[N, skip + side*side, C] bf16 -> [N, (side/2)^2, 4C]: class-token drop + pixel
    shuffle with scale 0.5 (lcvlm_modellink/pretrain_lcvlm.py:463-470, 546-556)."""
    _require_gpu(vit_output)
    if vit_output.dtype != torch.bfloat16:
        raise LvAttnError("pixel_shuffle_tokens expects bfloat16")
    x = vit_output if (vit_output.stride(2) == 1 and vit_output.stride(1) == vit_output.shape[2]) else vit_output.contiguous()
    n, t, c = x.shape
    side = int(round(math.sqrt(t - skip)))
    if side * side != t - skip:
        raise LvAttnError(f"{t - skip} tokens do not form a square grid")
    out = torch.empty((n, (side // 2) ** 2, 4 * c), dtype=x.dtype, device=x.device)
    check(_lib.lib().lv_pixel_shuffle(_ptr(x), _ptr(out), n, side, c, skip, x.stride(0), _stream()), "lv_pixel_shuffle")
    return out


class _PixelShuffleV2Tokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vit_output, skip, scale_stride):
        ctx.skip, ctx.st, ctx.in_shape = skip, scale_stride, vit_output.shape
        return pixel_shuffle_v2_tokens_launch(vit_output, skip, scale_stride)

    @staticmethod
    def backward(ctx, dy):      # every input token went to exactly one place: read it back from there (the padding has no source)
        n, t, c = ctx.in_shape
        st = ctx.st
        side = int(round(math.sqrt(t - ctx.skip)))
        os_ = (side + st - 1) // st
        g = dy.reshape(n, os_, os_, st, st, c).permute(0, 1, 3, 2, 4, 5).reshape(n, os_ * st, os_ * st, c)
        dx = torch.zeros(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        dx[:, ctx.skip:] = g[:, :side, :side].reshape(n, side * side, c)
        return dx, None, None


def pixel_shuffle_v2_tokens(vit_output: torch.Tensor, skip: int = 1, scale_stride: int = 2) -> torch.Tensor:
    """Differentiable form of ``pixel_shuffle_v2_tokens_launch``."""
    if torch.is_grad_enabled() and vit_output.requires_grad:
        return _PixelShuffleV2Tokens.apply(vit_output, skip, scale_stride)
    return pixel_shuffle_v2_tokens_launch(vit_output, skip, scale_stride)


def pixel_shuffle_v2_tokens_launch(vit_output: torch.Tensor, skip: int = 1, scale_stride: int = 2) -> torch.Tensor:
    """[N, skip + side*side, C] bf16 -> [N, ceil(side/st)^2, st*st*C]: class-token drop + the stride form of the token
    compression (lcvlm_modellink/pretrain_lcvlm.py:472-476, 558-570; zero padding on the right / bottom of an odd grid)."""
    _require_gpu(vit_output)
    if vit_output.dtype != torch.bfloat16:
        raise LvAttnError("pixel_shuffle_v2_tokens expects bfloat16")
    if int(scale_stride) < 1:
        raise LvAttnError("scale_stride must be a positive integer")
    x = vit_output if (vit_output.stride(2) == 1 and vit_output.stride(1) == vit_output.shape[2]) else vit_output.contiguous()
    n, t, c = x.shape
    side = int(round(math.sqrt(t - skip)))
    if side * side != t - skip:
        raise LvAttnError(f"{t - skip} tokens do not form a square grid")
    st = int(scale_stride)
    os_ = (side + st - 1) // st
    out = torch.empty((n, os_ * os_, st * st * c), dtype=x.dtype, device=x.device)
    check(_lib.lib().lv_pixel_shuffle_v2(_ptr(x), _ptr(out), n, side, c, skip, st, x.stride(0), _stream()), "lv_pixel_shuffle_v2")
    return out


def scatter_rows_(dst: torch.Tensor, dst_index: torch.Tensor, src: torch.Tensor, src_index: torch.Tensor | None = None):
    """dst[dst_index[i]] = src[src_index[i]] over 2-D row views (bf16, contiguous rows); a raw launch, no autograd:
    it refuses tensors that are part of a graph (use ``scatter_rows`` for those)."""
    if torch.is_grad_enabled() and (dst.requires_grad or src.requires_grad):
        raise LvAttnError("scatter_rows_ is a raw launch without a gradient; call ops.scatter_rows on tensors that need one")
    return _scatter_rows_launch(dst, dst_index, src, src_index)


def _scatter_rows_launch(dst, dst_index, src, src_index=None):
    _require_gpu(dst, src, dst_index)
    if dst.dtype != torch.bfloat16 or src.dtype != torch.bfloat16:
        raise LvAttnError("scatter_rows expects bfloat16")
    if not dst.is_contiguous():
        raise LvAttnError("scatter_rows needs a contiguous destination")
    src = src.contiguous()
    dst_index = dst_index.to(torch.int64).contiguous()
    if src_index is not None:
        src_index = src_index.to(torch.int64).contiguous()
        if src_index.numel() != dst_index.numel():
            raise LvAttnError("scatter_rows: source and destination index lists differ in length")
    n = dst_index.numel()
    width = dst.shape[-1]
    if src.shape[-1] != width:
        raise LvAttnError("scatter_rows: source and destination rows differ in width")
    if src_index is None and src.numel() // width < n:
        raise LvAttnError("scatter_rows: fewer source rows than destination indices")
    if n:
        check(_lib.lib().lv_scatter_rows(_ptr(src), _ptr(dst), _ptr(src_index), _ptr(dst_index), n, width,
                                         src.numel() // width, dst.numel() // width, _stream()), "lv_scatter_rows")
    return dst


class _ScatterRows(torch.autograd.Function):
    """In-place row scatter with the gradient torch derives for ``dst[index] = src[src_index]``
    (language_model_embedding.py:123,126,131): the overwritten rows of ``dst`` pass nothing back, the rows of
    ``src`` that were written somewhere receive the gradient of the row they landed in."""

    @staticmethod
    def forward(ctx, dst, src, dst_index, src_index):
        ctx.src_shape = src.shape
        ctx.has_src_index = src_index is not None
        ctx.save_for_backward(dst_index, src_index if src_index is not None else dst_index)
        ctx.mark_dirty(dst)
        _scatter_rows_launch(dst, dst_index, src, src_index)
        return dst

    @staticmethod
    def backward(ctx, g):
        dst_index, src_index = ctx.saved_tensors
        g = g.contiguous()
        width = g.shape[-1]
        g2 = g.view(-1, width)
        n = dst_index.numel()
        d_src = d_dst = None
        if ctx.needs_input_grad[1]:
            picked = gather_rows_launch(g2, dst_index)                       # gradient of the rows that were written
            n_src_rows = 1
            for x in ctx.src_shape[:-1]:
                n_src_rows *= x
            if not ctx.has_src_index and n_src_rows == n:
                d_src = picked.view(ctx.src_shape)
            else:                                                            # rows never written keep a zero gradient
                d_src = torch.zeros((n_src_rows, width), dtype=g.dtype, device=g.device)
                _scatter_rows_launch(d_src, src_index if ctx.has_src_index else torch.arange(n, device=g.device), picked)
                d_src = d_src.view(ctx.src_shape)
        if ctx.needs_input_grad[0]:
            d_dst = g2.clone()
            zero_row = torch.zeros((1, width), dtype=g.dtype, device=g.device)
            _scatter_rows_launch(d_dst, dst_index, zero_row, torch.zeros_like(dst_index))
            d_dst = d_dst.view(g.shape)
        return d_dst, d_src, None, None


def scatter_rows(dst: torch.Tensor, dst_index: torch.Tensor, src: torch.Tensor, src_index: torch.Tensor | None = None):
    """Differentiable ``dst[dst_index[i]] = src[src_index[i]]`` (in place on ``dst``, which must not be a leaf that
    requires grad, exactly as for torch's own indexed assignment).  ``dst`` [rows, width] contiguous bf16; the
    destination indices must be distinct, and so must the source indices (both are by construction on this path:
    lcvlm_modellink/training/utils.py:314-333 builds them from a boolean mask)."""
    if torch.is_grad_enabled() and (dst.requires_grad or src.requires_grad):
        return _ScatterRows.apply(dst, src, dst_index, src_index)
    return _scatter_rows_launch(dst, dst_index, src, src_index)


class _KeepGraphAlive(torch.autograd.Function):
    """``word_embeddings += features.mean() * 0`` (language_model_embedding.py:134, "to avoid backward hanging"):
    the values stay as they are, ``features`` joins the graph and receives a zero gradient, so that its producer's
    parameters are used in every step."""

    @staticmethod
    def forward(ctx, dst, src):
        ctx.src_shape, ctx.src_dtype = src.shape, src.dtype
        ctx.mark_dirty(dst)
        return dst

    @staticmethod
    def backward(ctx, g):
        d_src = torch.zeros(ctx.src_shape, dtype=ctx.src_dtype, device=g.device) if ctx.needs_input_grad[1] else None
        return g, d_src


def keep_graph_alive_(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and (dst.requires_grad or src.requires_grad):
        return _KeepGraphAlive.apply(dst, src)
    return dst


def gather_rows_launch(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """out[i] = src[index[i]] along dim 0 (any dtype whose row size is a multiple of 4 bytes); no autograd."""
    _require_gpu(src, index)
    src = src.contiguous()
    index = index.to(torch.int64).contiguous()
    row_bytes = src[0].numel() * src.element_size() if src.shape[0] else 0
    out = torch.empty((index.numel(),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if index.numel() == 0 or row_bytes == 0:
        return out
    if row_bytes % 4:
        raise LvAttnError("gather_rows: rows must be a multiple of 4 bytes")
    check(_lib.lib().lv_gather_rows(_ptr(src), _ptr(out), _ptr(index), index.numel(), row_bytes, src.shape[0], _stream()),
          "lv_gather_rows")
    return out


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, unique):
        ctx.src_shape, ctx.unique = src.shape, unique
        ctx.save_for_backward(index)
        return gather_rows_launch(src, index)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        g = g.contiguous()
        d_src = torch.zeros(ctx.src_shape, dtype=g.dtype, device=g.device)
        if index.numel():
            row_bytes = g[0].numel() * g.element_size()
            if ctx.unique and row_bytes % 4 == 0:   # a byte move: any dtype goes through the row kernel as 2-byte units
                _scatter_rows_launch(d_src.view(d_src.shape[0], -1).view(torch.bfloat16), index,
                                     g.view(g.shape[0], -1).view(torch.bfloat16))
            else:                                   # repeated rows: their gradients add up
                d_src.index_add_(0, index.to(torch.int64), g)
        return d_src, None, None


def gather_rows(src: torch.Tensor, index: torch.Tensor, unique_index: bool = False) -> torch.Tensor:
    """out[i] = src[index[i]] along dim 0, differentiable in ``src``: the gradient of row j is the sum of the
    gradients of the output rows that picked it (``unique_index=True`` promises that no row is picked twice, which
    lets the backward be a plain row scatter)."""
    if torch.is_grad_enabled() and src.requires_grad:
        return _GatherRows.apply(src, index, bool(unique_index))
    return gather_rows_launch(src, index)


# ---------------------------------------------------------------------------
# rotary position embedding
# ---------------------------------------------------------------------------
def rope_launch(x: torch.Tensor, cos_tab: torch.Tensor, sin_tab: torch.Tensor, pos_ids: torch.Tensor | None = None,
                tab_div: int = 1, conj: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
    """x [rows, heads, d] bf16 (last dim contiguous) -> rotated copy (or in place when out is x).
    cos_tab / sin_tab [tab_rows, rot] bf16 contiguous; table row = pos_ids[row] (int32) or row // tab_div."""
    _require_gpu(x, cos_tab, sin_tab)
    if x.dtype != torch.bfloat16 or cos_tab.dtype != torch.bfloat16 or sin_tab.dtype != torch.bfloat16:
        raise LvAttnError("rope_launch expects bfloat16 tensors and tables")
    if x.dim() != 3:
        raise LvAttnError("rope_launch expects [rows, heads, d]")
    x = _last_contig(x)
    cos_tab, sin_tab = cos_tab.contiguous(), sin_tab.contiguous()
    if cos_tab.shape != sin_tab.shape or cos_tab.dim() != 2:
        raise LvAttnError("cos / sin tables must be [tab_rows, rot] of one shape")
    rows, heads, d = x.shape
    tab_rows, rot = cos_tab.shape
    if out is None:
        out = torch.empty((rows, heads, d), dtype=x.dtype, device=x.device)
    if pos_ids is not None:
        _require_gpu(pos_ids)
        if pos_ids.dtype != torch.int32 or pos_ids.numel() != rows or not pos_ids.is_contiguous():
            raise LvAttnError("pos_ids must be a contiguous int32 tensor with one entry per row")
    check(_lib.lib().lv_rope(_ptr(x), _ptr(out), _ptr(cos_tab), _ptr(sin_tab), _ptr(pos_ids), rows, heads, d, rot,
                             x.stride(0), x.stride(1), out.stride(0), out.stride(1), tab_rows, int(tab_div), int(bool(conj)),
                             _stream()), "lv_rope")
    return out


class _Rope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cos_tab, sin_tab, pos_ids, tab_div):
        ctx.save_for_backward(cos_tab, sin_tab, pos_ids if pos_ids is not None else torch.empty(0))
        ctx.has_pos = pos_ids is not None
        ctx.tab_div = tab_div
        return rope_launch(x, cos_tab, sin_tab, pos_ids, tab_div, conj=False)

    @staticmethod
    def backward(ctx, dout):
        cos_tab, sin_tab, pos = ctx.saved_tensors
        dx = rope_launch(dout, cos_tab, sin_tab, pos if ctx.has_pos else None, ctx.tab_div, conj=True)
        return dx, None, None, None, None


def rope(x: torch.Tensor, cos_tab: torch.Tensor, sin_tab: torch.Tensor, pos_ids: torch.Tensor | None = None, tab_div: int = 1):
    """Differentiable rotary embedding of x [rows, heads, d] (see rope_launch)."""
    return _Rope.apply(x, cos_tab, sin_tab, pos_ids, int(tab_div))


def _check_rope_tables(cos_tab, sin_tab, pos_ids, rows: int, who: str):
    _require_gpu(cos_tab, sin_tab)
    if cos_tab.dtype != torch.bfloat16 or sin_tab.dtype != torch.bfloat16:
        raise LvAttnError(f"{who}: cos / sin tables must be bfloat16")
    if cos_tab.shape != sin_tab.shape or cos_tab.dim() != 2 or not cos_tab.is_contiguous() or not sin_tab.is_contiguous():
        raise LvAttnError(f"{who}: cos / sin tables must be contiguous [tab_rows, rot] of one shape")
    if pos_ids is not None:
        _require_gpu(pos_ids)
        if pos_ids.dtype != torch.int32 or pos_ids.numel() != rows or not pos_ids.is_contiguous():
            raise LvAttnError(f"{who}: pos_ids must be a contiguous int32 tensor with one entry per row")


def attn_fwd_rope_launch(q, k_rot, v, desc: torch.Tensor, scale: float, causal: bool, cos_tab, sin_tab, pos_ids=None,
                         tab_div: int = 1, out=None, lse=None, q_rot=None):
    """``attn_fwd_launch`` on UNROTATED queries: the kernel rotates each query row in registers as it loads it (tables
    and table-row rule of ``rope_launch``, rot == d == 128), so the result equals attn_fwd_launch(rope(q), k_rot, v) on the
    ping-pong kernel bit for bit without the rotated queries ever going through memory (a plain launch that takes the
    one-wave kernel -- the default for key sweeps of 32,768 and more -- differs by kernel rounding).  ``k_rot`` must already be rotated.  ``q_rot``
    (optional, bf16 [rows, hq, d]) receives the rotated queries (the backward kernels read them)."""
    q, k_rot, v = _last_contig(q), _last_contig(k_rot), _last_contig(v)
    _check_qkv(q, k_rot, v)
    rq, hq, d = q.shape
    rk, hkv, _ = k_rot.shape
    _check_rope_tables(cos_tab, sin_tab, pos_ids, rq, "attn_fwd_rope_launch")
    if d != 128 or cos_tab.shape[1] != d:
        raise LvAttnError(f"attn_fwd_rope_launch: head dim {d} with rotary width {cos_tab.shape[1]} (the fused rotation needs 128 / 128)")
    if out is None:
        out = torch.empty((rq, hq, d), dtype=torch.bfloat16, device=q.device)
    if lse is None:
        lse = torch.empty((hq, rq), dtype=torch.float32, device=q.device)
    if q_rot is not None and (q_rot.shape != q.shape or q_rot.dtype != torch.bfloat16 or q_rot.device != q.device or q_rot.stride(2) != 1):
        raise LvAttnError("attn_fwd_rope_launch: q_rot must be bf16 of q's shape on q's device with a contiguous last dimension")
    check(_lib.lib().lvattn_fwd_rope(
        _ptr(q), _ptr(k_rot), _ptr(v), _ptr(out), _ptr(lse), _ptr(desc), desc.shape[0],
        rq, rk, hq, hkv, d,
        q.stride(0), q.stride(1), k_rot.stride(0), k_rot.stride(1), v.stride(0), v.stride(1),
        out.stride(0), out.stride(1), lse.stride(0), float(scale), int(causal),
        _ptr(cos_tab), _ptr(sin_tab), _ptr(pos_ids), cos_tab.shape[0], int(tab_div),
        _ptr(q_rot), q_rot.stride(0) if q_rot is not None else 0, q_rot.stride(1) if q_rot is not None else 0,
        _stream()), "lvattn_fwd_rope")
    return out, lse


def rope_pack_kv(k: torch.Tensor, v: torch.Tensor, cos_tab: torch.Tensor, sin_tab: torch.Tensor, pos_ids=None, tab_div: int = 1,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """k, v [rows, heads, d] bf16 -> kv [2, rows, heads, d] contiguous with kv[0] = rope(k), kv[1] = v, in one pass
    (the K/V message of a ring step, rotated on the way in)."""
    k, v = _last_contig(k), _last_contig(v)
    _require_gpu(k, v)
    if k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16 or k.dim() != 3 or k.shape != v.shape:
        raise LvAttnError("rope_pack_kv expects k, v bf16 [rows, heads, d] of one shape")
    rows, heads, d = k.shape
    _check_rope_tables(cos_tab, sin_tab, pos_ids, rows, "rope_pack_kv")
    if out is None:
        out = torch.empty((2, rows, heads, d), dtype=torch.bfloat16, device=k.device)
    elif out.shape != (2, rows, heads, d) or out.dtype != torch.bfloat16 or out.device != k.device or not out.is_contiguous():
        raise LvAttnError("rope_pack_kv: out must be contiguous bf16 [2, rows, heads, d] on k's device")
    check(_lib.lib().lv_rope_pack_kv(_ptr(k), _ptr(v), _ptr(out), _ptr(cos_tab), _ptr(sin_tab), _ptr(pos_ids), rows, heads, d,
                                     cos_tab.shape[1], k.stride(0), k.stride(1), v.stride(0), v.stride(1), cos_tab.shape[0],
                                     int(tab_div), _stream()), "lv_rope_pack_kv")
    return out


class _RopeVarlenAttention(torch.autograd.Function):
    """Rotary embedding of q and k + packed attention as one differentiable op.  Forward: K is rotated by the
    bandwidth-bound kernel (every query block reads it), Q inside the attention kernel, which also writes the rotated
    queries out when a backward pass will need them.  Backward: the attention backward on the rotated q / k, then the
    transposed rotation of dq and dk in place."""

    @staticmethod
    def forward(ctx, q, k, v, cos_tab, sin_tab, pos_ids, tab_div, cu_seqlens, scale, causal):
        desc = blocks_on_device(cu_seqlens, causal, False, q.device)
        total = q.shape[0]
        covered = len(cu_seqlens) > 1 and cu_seqlens[0] == 0 and cu_seqlens[-1] == total
        out = None if covered else torch.zeros(q.shape, dtype=q.dtype, device=q.device)
        lse = None if covered else torch.full((q.shape[1], total), float("-inf"), dtype=torch.float32, device=q.device)
        need_bwd = any(ctx.needs_input_grad[:3])
        k_rot = rope_launch(k, cos_tab, sin_tab, pos_ids, tab_div)
        # rows outside every segment are not visited by the kernel: their q_rot is never read either
        q_rot = torch.empty(q.shape, dtype=q.dtype, device=q.device) if need_bwd else None
        out, lse = attn_fwd_rope_launch(q, k_rot, v, desc, scale, causal, cos_tab, sin_tab, pos_ids, tab_div, out, lse, q_rot)
        if need_bwd:
            ctx.save_for_backward(q_rot, k_rot, v, out, lse, cos_tab, sin_tab, pos_ids if pos_ids is not None else torch.empty(0))
        ctx.has_pos = pos_ids is not None
        ctx.tab_div, ctx.cu_seqlens, ctx.scale, ctx.causal = tab_div, tuple(cu_seqlens), scale, causal
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        q_rot, k_rot, v, out, lse, cos_tab, sin_tab, pos = ctx.saved_tensors
        pos = pos if ctx.has_pos else None
        desc_q = blocks_on_device(ctx.cu_seqlens, ctx.causal, False, q_rot.device)
        desc_k = blocks_on_device(ctx.cu_seqlens, ctx.causal, True, q_rot.device)
        dq, dk, dv, _ = attn_bwd_launch(q_rot, k_rot, v, out, dout, lse, desc_q, desc_k, ctx.scale, ctx.causal)
        rope_launch(dq, cos_tab, sin_tab, pos, ctx.tab_div, conj=True, out=dq)
        rope_launch(dk, cos_tab, sin_tab, pos, ctx.tab_div, conj=True, out=dk)
        return dq, dk, dv, None, None, None, None, None, None, None


def rope_varlen_attention(q, k, v, cos_tab, sin_tab, cu_seqlens: Sequence[int], scale: float | None = None, causal: bool = True,
                          pos_ids: torch.Tensor | None = None, tab_div: int = 1):
    """varlen_attention(rope(q), rope(k), v) with the query rotation folded into the attention kernel (head dim 128,
    rotary width 128).  q [T,Hq,128], k/v [T,Hkv,128] bf16 UNROTATED; cos_tab / sin_tab [tab_rows, 128] bf16; table row
    of token t = pos_ids[t] (int32) or t // tab_div.  Returns (out, lse); differentiable in q, k, v, and every result
    (out, lse, dq, dk, dv) is bit-identical to the unfused sequence rope -> varlen_attention run on the SAME forward kernel
    (the rotation lives in the ping-pong kernel; a plain launch whose work items sweep 32,768 keys or more takes the
    one-wave kernel by default and then differs from the fused result by kernel rounding, a few 1e-3 relative)."""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    return _RopeVarlenAttention.apply(q, k, v, cos_tab, sin_tab, pos_ids, int(tab_div), tuple(int(x) for x in cu_seqlens),
                                      float(scale), bool(causal))


# ---------------------------------------------------------------------------
# InternViT encoder pieces around the GEMMs
# ---------------------------------------------------------------------------
def _bf16_2d(t: torch.Tensor, what: str) -> torch.Tensor:
    _require_gpu(t)
    if t.dtype != torch.bfloat16:
        raise LvAttnError(f"{what} expects bfloat16")
    return t


def _vec(t: torch.Tensor | None, what: str, n: int, like: torch.Tensor, optional: bool = False):
    """A per-column operand (bias, gamma, LayerScale ...): the kernels read it through a raw pointer as ``n``
    contiguous bf16 values on ``like``'s device, so anything else must be refused here, not misread there."""
    if t is None:
        if optional:
            return None
        raise LvAttnError(f"{what} is required")
    if not t.is_cuda or t.device != like.device:
        raise LvAttnError(f"{what} must live on {like.device} (got {t.device})")
    if t.dtype != torch.bfloat16:
        raise LvAttnError(f"{what} must be bfloat16 (got {t.dtype}); cast the module with .to(torch.bfloat16)")
    if t.numel() != n:
        raise LvAttnError(f"{what} has {t.numel()} elements, expected {n}")
    return t.contiguous()


def layernorm_launch(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm over the last dimension of x [..., h] (bf16, fp32 statistics); no autograd."""
    _bf16_2d(x, "layernorm")
    h = x.shape[-1]
    gamma, beta = _vec(gamma, "layernorm weight", h, x), _vec(beta, "layernorm bias", h, x)
    x2 = x.reshape(-1, h)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], h), dtype=x.dtype, device=x.device)
    check(_lib.lib().lv_layernorm(_ptr(x2), _ptr(y), _ptr(gamma), _ptr(beta), x2.shape[0], h,
                                  x2.stride(0), y.stride(0), float(eps), _stream()), "lv_layernorm")
    return y.view(x.shape)


def layernorm_bwd_launch(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float):
    """Gradients of ``layernorm_launch``: (dx like x, dgamma [h] fp32, dbeta [h] fp32)."""
    _bf16_2d(x, "layernorm_bwd")
    _require_gpu(dy)
    h = x.shape[-1]
    gamma = _vec(gamma, "layernorm weight", h, x)
    if dy.shape != x.shape:
        raise LvAttnError(f"layernorm_bwd: dy {tuple(dy.shape)} does not match x {tuple(x.shape)}")
    x2, dy2 = x.reshape(-1, h), dy.reshape(-1, h)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    if dy2.stride(1) != 1 or dy2.dtype != x.dtype:
        dy2 = dy2.to(x.dtype).contiguous()
    rows = x2.shape[0]
    dx = torch.empty((rows, h), dtype=x.dtype, device=x.device)
    dgb = torch.empty((2, h), dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().lv_layernorm_bwd_workspace_bytes(rows, h)
    ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=x.device)
    check(_lib.lib().lv_layernorm_bwd(_ptr(x2), _ptr(dy2), _ptr(gamma), _ptr(dx), _ptr(dgb), _ptr(ws), nbytes, rows, h,
                                      x2.stride(0), dy2.stride(0), dx.stride(0), float(eps), _stream()), "lv_layernorm_bwd")
    return dx.view(x.shape), dgb[0], dgb[1]


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.param_dtypes = (gamma.dtype, beta.dtype)
        return layernorm_launch(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = layernorm_bwd_launch(x, dy, gamma, ctx.eps)
        return dx, dg.to(ctx.param_dtypes[0]), db.to(ctx.param_dtypes[1]), None


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm over the last dimension of x [..., h] (bf16 data and parameters, fp32 statistics), differentiable."""
    if torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad or beta.requires_grad):
        return _LayerNorm.apply(x, gamma, beta, eps)
    return layernorm_launch(x, gamma, beta, eps)


def bias_gelu_launch(x: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    _bf16_2d(x, "bias_gelu")
    x = x.contiguous()
    h = x.shape[-1]
    bias = _vec(bias, "bias_gelu bias", h, x, optional=True)
    y = torch.empty_like(x)
    check(_lib.lib().lv_bias_gelu(_ptr(x), _ptr(bias), _ptr(y), x.numel() // h, h,
                                  _stream()), "lv_bias_gelu")
    return y


def bias_gelu_bwd_launch(x: torch.Tensor, bias: torch.Tensor | None, dy: torch.Tensor) -> torch.Tensor:
    _bf16_2d(x, "bias_gelu_bwd")
    _require_gpu(dy)
    if dy.shape != x.shape:
        raise LvAttnError(f"bias_gelu_bwd: dy {tuple(dy.shape)} does not match x {tuple(x.shape)}")
    x, dy = x.contiguous(), dy.to(x.dtype).contiguous()
    h = x.shape[-1]
    bias = _vec(bias, "bias_gelu bias", h, x, optional=True)
    dx = torch.empty_like(x)
    check(_lib.lib().lv_bias_gelu_bwd(_ptr(x), _ptr(bias), _ptr(dy), _ptr(dx),
                                      x.numel() // h, h, _stream()), "lv_bias_gelu_bwd")
    return dx


class _BiasGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        ctx.save_for_backward(x, bias)
        return bias_gelu_launch(x, bias)

    @staticmethod
    def backward(ctx, dy):
        x, bias = ctx.saved_tensors
        dx = bias_gelu_bwd_launch(x, bias, dy)
        dbias = None
        if bias is not None and ctx.needs_input_grad[1]:
            dbias = dx.reshape(-1, dx.shape[-1]).sum(0, dtype=torch.float32).to(bias.dtype)
        return dx, dbias


def bias_gelu(x: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """gelu(x + bias), erf form, x [..., h] contiguous bf16; bias may be None; differentiable."""
    if torch.is_grad_enabled() and (x.requires_grad or (bias is not None and bias.requires_grad)):
        return _BiasGelu.apply(x, bias)
    return bias_gelu_launch(x, bias)


def bias_scale_residual_launch(x: torch.Tensor, bias: torch.Tensor | None, ls: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
    _bf16_2d(x, "bias_scale_residual")
    _bf16_2d(residual, "bias_scale_residual (residual)")
    if residual.shape != x.shape or residual.device != x.device:
        raise LvAttnError("bias_scale_residual: x and residual differ in shape or device")
    x, residual = x.contiguous(), residual.contiguous()
    h = x.shape[-1]
    bias, ls = _vec(bias, "bias_scale_residual bias", h, x, optional=True), _vec(ls, "LayerScale vector", h, x)
    out = torch.empty_like(x)
    check(_lib.lib().lv_bias_scale_residual(_ptr(x), _ptr(bias), _ptr(ls),
                                            _ptr(residual), _ptr(out), x.numel() // h, h, _stream()), "lv_bias_scale_residual")
    return out


def bias_scale_residual_layernorm_launch(x, bias, ls, residual, gamma, beta, eps: float):
    """(out, normed) with out = residual + (x + bias) * ls and normed = LayerNorm(out; gamma, beta) in one pass (no
    autograd): bit-identical to bias_scale_residual_launch followed by layernorm_launch, one read of out less."""
    _bf16_2d(x, "bias_scale_residual_layernorm")
    _bf16_2d(residual, "bias_scale_residual_layernorm (residual)")
    if residual.shape != x.shape or residual.device != x.device:
        raise LvAttnError("bias_scale_residual_layernorm: x and residual differ in shape or device")
    x, residual = x.contiguous(), residual.contiguous()
    h = x.shape[-1]
    if h > 8192:
        raise LvAttnError("bias_scale_residual_layernorm: rows wider than 8192 are not supported")
    bias, ls = _vec(bias, "bias_scale_residual_layernorm bias", h, x, optional=True), _vec(ls, "LayerScale vector", h, x)
    gamma, beta = _vec(gamma, "LayerNorm weight", h, x), _vec(beta, "LayerNorm bias", h, x)
    out, normed = torch.empty_like(x), torch.empty_like(x)
    check(_lib.lib().lv_bias_scale_residual_layernorm(_ptr(x), _ptr(bias), _ptr(ls), _ptr(residual), _ptr(out), _ptr(normed),
                                                      _ptr(gamma), _ptr(beta), x.numel() // h, h, float(eps), _stream()),
          "lv_bias_scale_residual_layernorm")
    return out, normed


class _BiasScaleResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, ls, residual):
        ctx.save_for_backward(x, bias, ls)
        return bias_scale_residual_launch(x, bias, ls, residual)

    @staticmethod
    def backward(ctx, dout):
        x, bias, ls = ctx.saved_tensors
        h = x.shape[-1]
        d2 = dout.reshape(-1, h)
        dx = (d2 * ls).view(x.shape) if ctx.needs_input_grad[0] else None
        col = d2.sum(0, dtype=torch.float32)
        dbias = (col * ls.float()).to(bias.dtype) if (bias is not None and ctx.needs_input_grad[1]) else None
        dls = None
        if ctx.needs_input_grad[2]:
            dls = (d2.float() * x.reshape(-1, h).float()).sum(0)
            if bias is not None:
                dls = dls + col * bias.float()
            dls = dls.to(ls.dtype)
        return dx, dbias, dls, (dout if ctx.needs_input_grad[3] else None)


def bias_scale_residual(x: torch.Tensor, bias: torch.Tensor | None, ls: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
    """residual + (x + bias) * ls with per-column bias / ls, all [..., h] contiguous bf16; differentiable."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, bias, ls, residual)):
        return _BiasScaleResidual.apply(x, bias, ls, residual)
    return bias_scale_residual_launch(x, bias, ls, residual)


def patchify(img: torch.Tensor, patch: int, kpad: int) -> torch.Tensor:
    """[n, c, H, W] bf16 -> [n * (H/patch) * (W/patch), kpad] rows in Conv2d-weight column order."""
    _bf16_2d(img, "patchify")
    img = img.contiguous()
    n, c, H, W = img.shape
    out = torch.empty((n * (H // patch) * (W // patch), kpad), dtype=img.dtype, device=img.device)
    check(_lib.lib().lv_patchify(_ptr(img), _ptr(out), n, c, H, W, patch, kpad, _stream()), "lv_patchify")
    return out


def embed_tokens_launch(proj, conv_bias, class_token, pos_emb) -> torch.Tensor:
    _bf16_2d(proj, "embed_tokens")
    proj = proj.contiguous()
    n, npatch, h = proj.shape
    ncls = 0 if class_token is None else class_token.numel() // h
    if pos_emb.dim() != 2 or pos_emb.shape[1] != h or pos_emb.shape[0] < npatch + ncls:
        raise LvAttnError("embed_tokens: position table must be [>= tokens, h]")
    conv_bias = _vec(conv_bias, "conv bias", h, proj)
    class_token = _vec(class_token, "class token", ncls * h, proj, optional=True)
    pos_emb = _vec(pos_emb, "position embedding", pos_emb.numel(), proj)
    out = torch.empty((npatch + ncls, n, h), dtype=proj.dtype, device=proj.device)
    check(_lib.lib().lv_embed_tokens(_ptr(proj), _ptr(conv_bias), _ptr(class_token),
                                     _ptr(pos_emb), _ptr(out), n, npatch, ncls, h, _stream()), "lv_embed_tokens")
    return out


class _EmbedTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj, conv_bias, class_token, pos_emb):
        ctx.shapes = (proj.shape, None if class_token is None else class_token.shape, pos_emb.shape)
        ctx.dtypes = (conv_bias.dtype, None if class_token is None else class_token.dtype, pos_emb.dtype)
        return embed_tokens_launch(proj, conv_bias, class_token, pos_emb)

    @staticmethod
    def backward(ctx, dtok):                      # dtok [s, n, h]
        (n, npatch, h), cls_shape, pos_shape = ctx.shapes
        ncls = dtok.shape[0] - npatch
        dproj = dtok[ncls:].transpose(0, 1).contiguous()
        per_pos = dtok.sum(1, dtype=torch.float32)             # [s, h]
        dbias = per_pos[ncls:].sum(0).to(ctx.dtypes[0])
        dcls = per_pos[:ncls].reshape(cls_shape).to(ctx.dtypes[1]) if cls_shape is not None else None
        dpos = torch.zeros(pos_shape, dtype=torch.float32, device=dtok.device)
        dpos[: dtok.shape[0]] = per_pos
        return dproj, dbias, dcls, dpos.to(ctx.dtypes[2])


def embed_tokens(proj: torch.Tensor, conv_bias: torch.Tensor, class_token: torch.Tensor | None, pos_emb: torch.Tensor) -> torch.Tensor:
    """proj [n, n_patches, h] -> tokens [n_class + n_patches, n, h] = cat(class_token, proj + conv_bias) + pos_emb;
    differentiable in all four."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (proj, conv_bias, class_token, pos_emb)):
        return _EmbedTokens.apply(proj, conv_bias, class_token, pos_emb)
    return embed_tokens_launch(proj, conv_bias, class_token, pos_emb)


# ---------------------------------------------------------------------------
# diagnostics
# ---------------------------------------------------------------------------
def mfma_clock_mhz(device=None, milliseconds: float = 40.0) -> dict:
    """Shader clock (MHz) the chip holds under a dense bf16 MFMA loop on random data, from in-kernel stamps of the
    shader-clock counter against the constant 100 MHz counter: median, min and max over all waves."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n_wg = 2 * torch.cuda.get_device_properties(device).multi_processor_count
    iters = max(1, int(milliseconds * 1e-3 * 2.0e9 / 128 / 2))      # 4 MFMAs of 32 cycles per iteration, two workgroups per CU
    seed = torch.randn(1024, device=device).to(torch.bfloat16)
    stamps = torch.zeros((n_wg * 4, 2), dtype=torch.int64, device=device)
    sink = torch.empty(n_wg * 4, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(_lib.lib().lv_clock_probe(_ptr(seed), _ptr(stamps), _ptr(sink), n_wg, iters, _stream()), "lv_clock_probe")
        torch.cuda.synchronize()
    st = stamps.cpu().double()
    mhz = 100.0 * st[:, 0] / st[:, 1].clamp(min=1)
    return {"median": float(mhz.median()), "min": float(mhz.min()), "max": float(mhz.max()), "waves": int(mhz.numel())}
