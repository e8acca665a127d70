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
