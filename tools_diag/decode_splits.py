"""Diagnostic: decode call time against the number of key ranges (n_split), 1M-token and 128K-token caches, 40 / 8 heads."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops
dev = torch.device("cuda")
for klen in [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1 << 20, 1 << 17, 8192):
    g = torch.Generator(device=dev).manual_seed(1)
    q = (torch.randn(1, 40, 128, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    k = (torch.randn(1, klen, 8, 128, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    v = (torch.randn(1, klen, 8, 128, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    for ns in [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 16, 32, 48, 64, 96, 128, 192, 256):
        for _ in range(3):
            ops.decode_attention(q, k, v, 128 ** -0.5, ns)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.decode_attention(q, k, v, 128 ** -0.5, ns)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"k_len {klen:8d} n_split {ns:3d}  {ms * 1000:8.1f} us  {2 * klen * 8 * 128 * 2 / ms / 1e9:7.2f} TB/s", flush=True)
