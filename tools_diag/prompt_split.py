"""A short block of new tokens against a long KV cache (the prompt pass of the kv-cache branch): one sweep per query block
against the key range cut into pieces (ops.attn_fwd_rect_split_keys).  python tools_diag/prompt_split.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops  # noqa: E402

HQ, HKV, D = 40, 8, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for sk in (131072, 1048576):
    k = (torch.randn(sk, HKV, D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    v = (torch.randn(sk, HKV, D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    for sq in (64, 256, 300, 1024, 4096):
        q = (torch.randn(sq, HQ, D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        res = {}
        for name, pieces in (("one sweep", 1), ("split", None)):
            f = lambda: ops.attn_fwd_rect_split_keys(q, k, v, D ** -0.5, True, sk - sq, n_pieces=pieces)
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 5
        P = ops._key_pieces_for(sq, sk, HQ, dev)
        flops = 4.0 * D * HQ * sq * (sk - sq / 2)
        print(f"cache {sk:8d}  new tokens {sq:5d}: one sweep {res['one sweep']:8.3f} ms   {P:2d} pieces {res['split']:8.3f} ms "
              f"({flops / res['split'] / 1e9:6.0f} TFLOP/s)   x{res['one sweep'] / res['split']:.2f}", flush=True)
