"""Incremental attention, 48 layers' worth of calls per token: plain launches against one captured HIP graph
(lvattn_decode_noalloc allocates nothing, so it can be captured).  python tools_diag/decode_graph.py [cache_len ...]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops  # noqa: E402

HQ, HKV, D, LAYERS = 40, 8, 128, 48
dev = torch.device("cuda", 0)
for S in [int(x) for x in sys.argv[1:]] or [8192, 32768, 131072, 1048576]:
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(1, HQ, D, generator=g, device=dev).to(torch.bfloat16)
    n_cache = 4 if S >= 1 << 20 else LAYERS        # the 1M caches of 48 layers do not fit beside each other: four, cycled
    kc = [torch.randn(1, S, HKV, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(n_cache)]
    vc = [torch.randn(1, S, HKV, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(n_cache)]
    out = torch.empty((LAYERS, 1, HQ, D), dtype=torch.bfloat16, device=dev)
    scale = D ** -0.5

    def token():
        for l in range(LAYERS):
            ops.decode_attention(q, kc[l % n_cache], vc[l % n_cache], scale, out=out[l], library_scratch=True)
    token()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        token()

    def timed(fn, reps=20):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    a, b = timed(token), timed(graph.replay)
    gb = LAYERS * 2.0 * S * HKV * D * 2 / 1e9
    print(f"cache {S:8d} tokens x {LAYERS} layers: plain launches {a:8.3f} ms ({gb / a:5.2f} TB/s)   one graph {b:8.3f} ms ({gb / b:5.2f} TB/s)", flush=True)
    del kc, vc
