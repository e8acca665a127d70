#!/usr/bin/env python3
"""Per-rank compute time of the ring forward on ONE GPU: the rank's block kernels and merges run as they
would in a CP group, the exchange is replaced by a local copy on the side stream (same dependency
structure, no fabric).  Usage: python tools_ring_rank.py [--cp 8] [--seq 1048576] [--ranks 0,7] [--bwd]"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from long_vita_amd.ring_context_parallel import RingAttention  # noqa: E402

HQ, HKV, D = 40, 8, 128


class LocalCopyComm:
    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)

    def exchange(self, send_bufs, recv_bufs, dst, src):
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for s, r in zip(send_bufs, recv_bufs):
                r.copy_(s)
            done = torch.cuda.Event()
            done.record(self.stream)
        return done

    def wait(self, handle):
        torch.cuda.current_stream().wait_event(handle)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cp", type=int, default=8)
    ap.add_argument("--seq", type=int, default=1048576)
    ap.add_argument("--ranks", default="0,7")
    ap.add_argument("--bwd", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    s_loc = a.seq // a.cp
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v = rnd(s_loc, 1, HQ * D), rnd(s_loc, 1, HKV * D), rnd(s_loc, 1, HKV * D)
    scale = 1.0 / math.sqrt(D)
    ideal = 2.0 * D * HQ * float(a.seq) ** 2 / a.cp       # causal forward FLOPs per rank
    for r in (int(x) for x in a.ranks.split(",")):
        eng = RingAttention(None, a.cp, r, causal=True, comm=LocalCopyComm(dev))
        out, lse = eng.forward(q, k, v, HQ, scale, return_state=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, lse = eng.forward(q, k, v, HQ, scale, return_state=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"cp {a.cp} rank {r} forward {dt * 1e3:.1f} ms  {ideal / dt / 1e12:.0f} TFLOP/s per GPU", flush=True)
        if a.bwd:
            do = rnd(s_loc, 1, HQ * D)
            eng.backward(q, k, v, out, lse, do, HQ, scale)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.backward(q, k, v, out, lse, do, HQ, scale)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"cp {a.cp} rank {r} backward {dt * 1e3:.1f} ms  {2.5 * ideal / dt / 1e12:.0f} TFLOP/s per GPU (algorithmic)", flush=True)


if __name__ == "__main__":
    main()
