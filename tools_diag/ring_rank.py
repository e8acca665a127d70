#!/usr/bin/env python3
"""Per-rank compute time of the ring forward on ONE GPU: the rank's block kernels and merges run as they
would in a CP group, the exchange is replaced by a local copy on the side stream (same dependency
structure, no fabric).  With --rccl the copy is a real RCCL send + receive of the block from this one rank to itself
(process group "nccl", world size 1), posted by the shipped transport class: the RCCL kernel then competes with the
block kernel for the GPU as it would on a node, only its bytes stay in local memory.
Usage: python tools_diag/ring_rank.py [--cp 8] [--seq 1048576] [--ranks 0,7] [--bwd] [--rccl]"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def self_rccl_comm(dev):
    import torch.distributed as dist
    from long_vita_amd.ring_context_parallel import TorchDistComm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    class SelfComm(TorchDistComm):      # every virtual neighbour is this rank
        def exchange(self, send_bufs, recv_bufs, dst, src, lane=0):
            return super().exchange(send_bufs, recv_bufs, 0, 0, lane)
    return SelfComm(dist.group.WORLD, dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cp", type=int, default=8)
    ap.add_argument("--seq", type=int, default=1048576)
    ap.add_argument("--ranks", default="0,7")
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--rccl", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = self_rccl_comm(dev) if a.rccl else LocalCopyComm(dev)
    s_loc = a.seq // a.cp
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v = rnd(s_loc, 1, HQ * D), rnd(s_loc, 1, HKV * D), rnd(s_loc, 1, HKV * D)
    scale = 1.0 / math.sqrt(D)
    ideal = 2.0 * D * HQ * float(a.seq) ** 2 / a.cp       # causal forward FLOPs per rank
    for r in (int(x) for x in a.ranks.split(",")):
        eng = RingAttention(None, a.cp, r, causal=True, comm=comm)
        out, lse = eng.forward(q, k, v, HQ, scale, return_state=True)
        torch.cuda.synchronize()
        eng.profile = True
        t0 = time.perf_counter()
        out, lse = eng.forward(q, k, v, HQ, scale, return_state=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.profile = False
        steps = " ".join(f"{c:.1f}+{w:.1f}" for c, w in eng.step_stats())
        print(f"cp {a.cp} rank {r} forward {dt * 1e3:.1f} ms  {ideal / dt / 1e12:.0f} TFLOP/s per GPU  "
              f"[{'rccl self send/recv' if a.rccl else 'local copy'}; per step compute+wait ms: {steps}]", flush=True)
        if a.bwd:
            do = rnd(s_loc, 1, HQ * D)
            eng.backward(q, k, v, out, lse, do, HQ, scale)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.backward(q, k, v, out, lse, do, HQ, scale)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"cp {a.cp} rank {r} backward {dt * 1e3:.1f} ms  {2.5 * ideal / dt / 1e12:.0f} TFLOP/s per GPU (algorithmic)", flush=True)
    if a.rccl:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
