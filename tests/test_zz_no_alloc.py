"""The allocation-free variant of the incremental-attention entry point (lvattn_decode_noalloc: partial results in a
fixed buffer inside the library image) against lvattn_decode, through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(b, k_len, hq, hkv, d, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *shape: torch.randn(*shape, generator=g, device="cuda").to(torch.bfloat16)
    return rnd(b, hq, d), rnd(b, k_len, hkv, d), rnd(b, k_len, hkv, d)


def test_decode_noalloc_equals_decode_two_calls_in_a_row():
    """Two calls in a row on the shared buffer, the second with a smaller input (fewer key ranges, fewer heads): each
    must equal the workspace-taking entry point bit for bit (same kernels, same split), and no device memory may be
    allocated by the call when the outputs are passed in."""
    from long_vita_amd import _lib, ops
    for b, k_len, hq, hkv, d, seed in ((2, 70001, 40, 8, 128, 1), (1, 3000, 8, 2, 64, 2)):
        q, k, v = _case(b, k_len, hq, hkv, d, seed)
        scale = d ** -0.5
        want, want_lse = ops.decode_attention(q, k, v, scale, return_lse=True)
        out = torch.full((b, hq, d), float("nan"), dtype=torch.bfloat16, device="cuda")
        lse = torch.full((b, hq), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats()["allocation.all.allocated"]
        ops.decode_attention(q, k, v, scale, return_lse=True, out=out, lse=lse, library_scratch=True)
        after = torch.cuda.memory_stats()["allocation.all.allocated"]
        torch.cuda.synchronize()
        assert after == before                      # nothing allocated through torch; the library has no hipMalloc at all
        assert torch.equal(out, want) and torch.equal(lse, want_lse), (b, k_len)
    # beyond the buffer: a clear error, nothing launched
    assert _lib.lib().lvattn_decode_noalloc_max_bytes() == 8 << 20
    q, k, v = _case(64, 4096, 40, 8, 128, 3)
    with pytest.raises(_lib.LvAttnError, match="bytes of partial results"):
        ops.decode_attention(q, k, v, 0.1, n_split=32, library_scratch=True)
    ops.decode_attention(q, k, v, 0.1, n_split=32)              # the workspace form takes the same shape


def test_decode_noalloc_under_graph_capture():
    """What the entry point is for: a caller under HIP graph capture cannot allocate.  One incremental-attention call is
    captured into a graph (after one plain call on the device, which takes the buffer's address), then replayed on new
    query / cache CONTENTS in the same tensors: each replay equals the workspace-taking entry point on those contents."""
    from long_vita_amd import ops
    b, k_len, hq, hkv, d = 1, 20000, 40, 8, 128
    scale = d ** -0.5
    q, k, v = _case(b, k_len, hq, hkv, d, 11)
    out = torch.empty((b, hq, d), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty((b, hq), dtype=torch.float32, device="cuda")
    ops.decode_attention(q, k, v, scale, return_lse=True, out=out, lse=lse, library_scratch=True)   # not captured
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.decode_attention(q, k, v, scale, return_lse=True, out=out, lse=lse, library_scratch=True)
    for seed in (12, 13):
        q2, k2, v2 = _case(b, k_len, hq, hkv, d, seed)
        q.copy_(q2); k.copy_(k2); v.copy_(v2)
        out.fill_(float("nan")); lse.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        want, want_lse = ops.decode_attention(q2, k2, v2, scale, return_lse=True)
        assert torch.equal(out, want) and torch.equal(lse, want_lse), seed
