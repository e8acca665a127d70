"""The C-ABI library loads without a GPU, exports every symbol include/lvattn.h
declares, and its pure-host entry points (block builders, argument validation)
behave.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lvattn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lv(?:attn)?_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from long_vita_amd import _lib
    handle = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(handle, name), name
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    assert handle.lvattn_abi_version() == 1


def _blocks(cu, causal, backward=False):
    from long_vita_amd import ops
    return ops.build_blocks_host(cu, causal, backward)


def test_forward_blocks_cover_segments_exactly():
    cu = [0, 100, 356, 357, 700, 1290, 1290, 5000]
    d = _blocks(cu, True)
    seen = np.zeros(cu[-1], dtype=int)
    for q0, qn, k0, kl, sh, *_ in d.tolist():
        seen[q0:q0 + qn] += 1
        seg = max(i for i in range(len(cu) - 1) if cu[i] <= q0 and cu[i + 1] > q0)
        assert k0 == cu[seg] and sh == q0 - k0
        assert kl == min(q0 + qn, cu[seg + 1]) - k0          # causal: keys up to the block's last row
        assert 1 <= qn <= 256
    assert (seen == 1).all()
    work = [qn * kl for _, qn, _, kl, *_ in d.tolist()]
    assert work == sorted(work, reverse=True)                # heaviest first
    dn = _blocks(cu, False)
    for q0, qn, k0, kl, sh, *_ in dn.tolist():
        seg = max(i for i in range(len(cu) - 1) if cu[i] <= q0 and cu[i + 1] > q0)
        assert (k0, kl) == (cu[seg], cu[seg + 1] - cu[seg])


def test_mean_keys_per_item_of_a_table():
    """The figure the forward launch hands to the library (lvattn_fwd_hinted): keys swept per descriptor on average."""
    from long_vita_amd import _lib
    L = _lib.lib()
    I32P = C.POINTER(C.c_int32)
    one = _blocks([0, 131072], True)                       # one causal segment: block b sweeps 256 (b + 1) keys
    assert L.lv_mean_keys_per_item(one.ctypes.data_as(I32P), one.shape[0]) == int(one[:, 3].astype(np.int64).mean()) == 65664
    packed = _blocks(list(range(0, 131073, 8192)), True)   # sixteen 8K segments: 128.5 x 32 keys on average
    assert L.lv_mean_keys_per_item(packed.ctypes.data_as(I32P), packed.shape[0]) == 4224
    assert L.lv_mean_keys_per_item(packed.ctypes.data_as(I32P), 0) == -1
    assert L.lv_mean_keys_per_item(None, 4) == -1
    # the hinted launch validates like the plain one, before any HIP call
    rc = L.lvattn_fwd_hinted(None, None, None, None, None, None, 1, 1, 1, 1, 1, 128, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1.0, 1, 4224, None)
    assert rc == -1 and b"null" in L.lvattn_last_error()


def test_backward_blocks_visibility_rule():
    cu = [0, 130, 700]
    d = _blocks(cu, True, backward=True)
    seen = np.zeros(cu[-1], dtype=int)
    for k0, kn, q0, ql, sh, *_ in d.tolist():
        seen[k0:k0 + kn] += 1
        seg = 0 if k0 < 130 else 1
        assert q0 + ql == cu[seg + 1]
        assert q0 <= k0 and (k0 - q0) < 32 and (q0 - cu[seg]) % 32 == 0
        # key k0+j visible to query q0+i iff j <= i + sh  <=>  k0+j <= q0+i
        assert sh == q0 - k0
    assert (seen == 1).all()


def test_rect_blocks_and_errors():
    from long_vita_amd import ops, _lib
    d = ops.build_rect_blocks_host(512, 512, 0, 1024, False, 0)
    assert d[:, 0].tolist() == [512, 768] and (d[:, 3] == 1024).all()
    d = ops.build_rect_blocks_host(0, 600, 0, 600, True, 0)
    assert sorted(d[:, 0].tolist()) == [0, 256, 512]
    with pytest.raises(_lib.LvAttnError):
        ops.build_blocks_host([0, 10, 5], True)
    # argument validation happens before any HIP call
    L = _lib.lib()
    rc = L.lvattn_fwd(None, None, None, None, None, None, 1, 1, 1, 1, 1, 128, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1.0, 1, None)
    assert rc == -1 and b"null" in L.lvattn_last_error()
    buf = C.create_string_buffer(64)
    p = C.addressof(buf) + (16 - C.addressof(buf) % 16)
    rc = L.lvattn_fwd(p, p, p, p, p, p, 1, 1, 1, 3, 2, 128, 128, 128, 128, 128, 128, 128, 128, 128, 1, 1.0, 1, None)
    assert rc == -1 and b"multiple" in L.lvattn_last_error()
    rc = L.lvattn_fwd(p, p, p, p, p, p, 1, 1, 1, 2, 2, 96, 128, 128, 128, 128, 128, 128, 128, 128, 1, 1.0, 1, None)
    assert rc == -1 and b"head dim" in L.lvattn_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from long_vita_amd import ops, _lib
    q = torch.zeros(4, 1, 128, dtype=torch.bfloat16)
    with pytest.raises(_lib.LvAttnError):
        ops.varlen_attention(q, q, q, [0, 4])


def test_argument_validation_rejects_before_any_launch():
    """Entry points check their arguments on the host and return LV_EINVAL with a message; none of these calls
    reaches a kernel launch (the pointers are never dereferenced), so they run without a GPU."""
    from long_vita_amd import _lib
    L = _lib.lib()
    buf = (C.c_char * 4096)()
    p = C.addressof(buf)          # 16-byte alignment is not guaranteed by ctypes: align by hand
    p = (p + 255) & ~255
    err = lambda: L.lvattn_last_error().decode()
    f = C.c_float(0.1)
    # head dim 96
    rc = L.lvattn_fwd(p, p, p, p, p, p, 1, 64, 64, 4, 2, 96, 1024, 96, 512, 96, 512, 96, 1024, 96, 64, f, 1, None)
    assert rc != 0 and "head dim" in err()
    # 5 query heads over 2 KV heads
    rc = L.lvattn_fwd(p, p, p, p, p, p, 1, 64, 64, 5, 2, 128, 1024, 128, 512, 128, 512, 128, 1024, 128, 64, f, 1, None)
    assert rc != 0 and "multiple" in err()
    # misaligned q
    rc = L.lvattn_fwd(p + 2, p, p, p, p, p, 1, 64, 64, 4, 2, 128, 1024, 128, 512, 128, 512, 128, 1024, 128, 64, f, 1, None)
    assert rc != 0 and "aligned" in err()
    # lse row shorter than the rows
    rc = L.lvattn_fwd(p, p, p, p, p, p, 1, 64, 64, 4, 2, 128, 1024, 128, 512, 128, 512, 128, 1024, 128, 32, f, 1, None)
    assert rc != 0 and "lse_stride" in err()
    # zero work items: accepted, nothing launched
    assert L.lvattn_fwd(p, p, p, p, p, p, 0, 64, 64, 4, 2, 128, 1024, 128, 512, 128, 512, 128, 1024, 128, 64, f, 1, None) == 0
    # a row stride the kernels' 32-bit tile offsets cannot span
    rc = L.lvattn_fwd(p, p, p, p, p, p, 1, 64, 64, 4, 2, 128, 1 << 23, 128, 512, 128, 512, 128, 1024, 128, 64, f, 1, None)
    assert rc != 0 and "row stride" in err()
    # the descriptor-free form: same checks; causal needs at least as many keys as rows
    rc = L.lvattn_fwd_dense(p, p, p, p, p, 64, 64, 4, 2, 96, 1024, 96, 512, 96, 512, 96, 1024, 96, 64, f, 0, None)
    assert rc != 0 and "head dim" in err()
    rc = L.lvattn_fwd_dense(p, p, p, p, p, 64, 32, 4, 2, 128, 1024, 128, 512, 128, 512, 128, 1024, 128, 64, f, 1, None)
    assert rc != 0 and "k_rows >= q_rows" in err()
    rc = L.lvattn_fwd_dense(p, p, p, p, None, 64, 64, 4, 2, 128, 1024, 128, 512, 128, 512, 128, 1024, 128, 64, f, 1, None)
    assert rc != 0 and "null" in err()
    # incremental attention: 33 query heads per KV head, empty cache, short workspace
    ws = L.lvattn_decode_workspace_bytes(1, 40, 8, 128)
    assert ws > 0 and L.lvattn_decode_workspace_bytes(1, 5, 2, 128) == 0
    rc = L.lvattn_decode(p, p, p, p, None, p, ws, 1, 100, 66, 2, 128, 8448, 128, 25600, 256, 128, 25600, 256, 128, 8448, 128, f, 0, None)
    assert rc != 0 and "at most 32" in err()
    rc = L.lvattn_decode(p, p, p, p, None, p, ws, 1, 0, 40, 8, 128, 5120, 128, 1024, 1024, 128, 1024, 1024, 128, 5120, 128, f, 0, None)
    assert rc != 0 and "k_len" in err()
    rc = L.lvattn_decode(p, p, p, p, None, p, 16, 1, 100000, 40, 8, 128, 5120, 128, 1024, 1024, 128, 1024, 1024, 128, 5120, 128, f, 0, None)
    assert rc == -3 and "bytes of partial results needed" in err()        # LV_ESIZE
    # the allocation-free form: same checks, and its fixed buffer bounds the shape (nothing is launched for either)
    assert L.lvattn_decode_noalloc_max_bytes() == 8 << 20
    rc = L.lvattn_decode_noalloc(p, p, p, p, None, 1, 0, 40, 8, 128, 5120, 128, 1024, 1024, 128, 1024, 1024, 128, 5120, 128, f, 0, None)
    assert rc != 0 and "k_len" in err()
    # LayerNorm backward: rows wider than it supports
    rc = L.lv_layernorm_bwd(p, p, p, p, p, p, 1 << 20, 4, 8192, 8192, 8192, 8192, C.c_float(1e-5), None)
    assert rc != 0 and "4096" in err()
    # unknown kernel name
    assert L.lvattn_select_fwd_kernel(b"nope") != 0 and "unknown forward kernel" in err()
    assert L.lvattn_select_fwd_kernel(b"default") == 0


def _split_tables(fn, cu, causal, piece):
    import ctypes
    cu_a = np.asarray(cu, dtype=np.int32)
    p = cu_a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    n_comb, ws_rows = ctypes.c_int(0), ctypes.c_int(0)
    n = fn(p, len(cu) - 1, causal, piece, None, 0, None, 0, ctypes.byref(n_comb), ctypes.byref(ws_rows))
    assert n >= 0
    desc = np.zeros((max(n, 1), 8), dtype=np.int32)
    comb = np.zeros((max(n_comb.value, 1), 4), dtype=np.int32)
    n2 = fn(p, len(cu) - 1, causal, piece, desc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), desc.shape[0],
            comb.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), comb.shape[0], ctypes.byref(n_comb), ctypes.byref(ws_rows))
    assert n2 == n
    return desc[:n], comb[:n_comb.value], ws_rows.value


@pytest.mark.parametrize("cu,causal", [([0, 8192], 1), ([0, 300, 301, 2000, 2000, 5000], 1), ([0, 1000, 3000], 0), ([0, 77], 1)])
def test_split_tables_cover_every_visible_pair_exactly_once(cu, causal):
    """The split-sweep tables (host side, no GPU): the pieces of the forward table and of the dK/dV table each cover every
    (query row, key) pair the packed causal (or full) mask allows exactly once and nothing else; every work item writes
    to workspace rows of its own; the combine entries list each block's pieces back to back."""
    from long_vita_amd import _lib
    L = _lib.lib()
    T = cu[-1]
    seg = np.full(T, -1)
    for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
        seg[a:b] = i
    want = (seg[:, None] == seg[None, :]) & (seg[:, None] >= 0)
    if causal:
        want &= np.arange(T)[None, :] <= np.arange(T)[:, None]
    for backward, fn, blk, piece in ((False, L.lv_build_fwd_blocks_split, 256, 512), (True, L.lv_build_bwd_blocks_split, 128, 256),
                                     (False, L.lv_build_fwd_blocks_split, 256, 1 << 30)):
        desc, comb, ws_rows = _split_tables(fn, cu, causal, piece)
        cover = np.zeros((T, T), dtype=np.int32)
        ws_used = np.zeros(ws_rows, dtype=np.int32)
        for a0, an, b0, bn, sh, od, _, _ in desc:
            assert 0 < an <= blk and bn >= 0
            ws_used[a0 + od:a0 + od + an] += 1
            ai = np.arange(an)[:, None]
            bj = np.arange(bn)[None, :]
            if backward:      # a = keys (jj), b = query rows (ii): jj <= ii + shift
                vis = (ai <= bj + sh) if causal else np.ones((an, bn), bool)
                cover[b0:b0 + bn, a0:a0 + an] += vis.T
            else:             # a = query rows (i), b = keys (j): j <= i + shift
                vis = (bj <= ai + sh) if causal else np.ones((an, bn), bool)
                cover[a0:a0 + an, b0:b0 + bn] += vis
        assert (cover == want).all(), (backward, piece)
        assert ws_used.max() <= 1
        assert int(comb[:, 3].sum()) == len(desc) and ws_rows == blk * len(desc)
        rows_covered = np.zeros(T, dtype=np.int32)
        for out_row0, rows, ws_row0, n_pieces in comb:
            rows_covered[out_row0:out_row0 + rows] += 1
            for pc in range(n_pieces):
                assert (ws_used[ws_row0 + pc * blk: ws_row0 + pc * blk + rows] == 1).all()
        assert (rows_covered == (seg >= 0)).all()
        if backward and piece < (1 << 30):
            assert all(bn % 32 == 0 or b0 + bn == cu[np.searchsorted(cu, b0, side="right")] for _, _, b0, bn, *_ in desc)


def test_split_planner_rule():
    """lv_plan_*_split: enough work items -> 0; one tensor-parallel rank's head count on a short batch -> a piece length."""
    import ctypes
    from long_vita_amd import _lib
    L = _lib.lib()

    def plan(cu, heads_q, heads_kv, n_cu=256):
        a = np.asarray(cu, dtype=np.int32)
        p = a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        return (L.lv_plan_fwd_split(p, len(cu) - 1, 1, heads_q, n_cu), L.lv_plan_bwd_split(p, len(cu) - 1, 1, heads_kv, heads_q // heads_kv, n_cu))
    assert plan([0, 131072], 40, 8) == (0, 0)                  # 20,480 forward items: nothing to gain
    pf, pb = plan([0, 8192], 5, 1)                             # 160 forward items / 64 key blocks on 256 compute units
    assert pf >= 256 and pf % 256 == 0 and pb >= 256 and pb % 32 == 0
    assert plan([0, 8192], 40, 8)[0] == 0 or plan([0, 8192], 40, 8)[0] >= 256
    assert plan(list(range(0, 4096 + 1, 256)), 5, 1)[0] == 0   # 16 one-block segments: no sweep longer than a piece
    assert L.lv_plan_fwd_split(None, 1, 1, 5, 256) < 0
