#!/usr/bin/env python3
"""Benchmark of the long-context attention hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--seq S]

With N > 1 either launch it under torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or run it plainly: without WORLD_SIZE it starts its own N ranks and relays rank 0's line.

Default workload ``prefill_1m``: one layer's causal attention prefill over a
1,048,576-token packed context with Qwen2.5-14B heads (40 query / 8 KV heads,
d = 128, bf16), single causal segment (the reference's stage-3/4 semantics:
joint packing, no mask reset).  With N > 1 the sequence is zig-zag sharded over
N ranks (context parallel, CP = N) and the KV blocks travel round a ring of
RCCL send/recv; the total work is fixed, so scaling is "strong".
A "step" is one full attention forward over the whole context.

Rank 0 prints ONE JSON line: metric tokens/s for the whole job, plus
  roofline     achieved TFLOP/s of the forward kernel (algorithmic FLOPs
               2*d*Hq*S^2 for a causal segment / mean kernel time from HIP
               events on the launch stream) against the 2.5 PFLOP/s dense
               bf16 MFMA peak,
  cpu_baseline the reference's eager PyTorch attention (oracle restatement)
               timed on the host cores on a bounded row sample, extrapolated.

Other workloads (``cfg3_fwd_bwd``: 128K packed tokens forward+backward; ``rope_qk``: rotary embedding of Q and K;
``cfg2_vit``: InternViT-300M attention on 64 frames; ``decode``: one token against a 1M-token KV cache; ``vit_tower``: the whole InternViT-300M encoder forward on 64 tiles) are for profiling; the
judged line is the default one.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (spec), TFLOP/s
PEAK_HBM_GBS = 8000.0       # MI355X HBM3E peak (spec), GB/s
HQ, HKV, D = 40, 8, 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="prefill_1m", choices=["prefill_1m", "cfg3_fwd_bwd", "cfg3_packed", "cfg2_vit", "vit_tower", "rope_qk", "rope_fused", "decode"])
    ap.add_argument("--seq", type=int, default=0, help="override the sequence length (testing)")
    ap.add_argument("--heads", default="", help="override 'q_heads,kv_heads' (testing; e.g. 5,1 = one rank's share under TP=8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def launch_command(n, argv, port):
    """The command line that starts ``n`` ranks of this script on this node (one process per GPU, rendezvous on the
    loopback address), as the reference's launch scripts do with torchrun (scripts/modellink/qwen25/
    finetune_qwen25_14b_intern_300m_ptd_tp8pp1_stage4.sh:239: starting the ranks is the script's job, not its caller's)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """``python bench.py --gpus N`` without a launcher around it: this process has made no GPU call yet and makes none;
    it starts the N ranks as a child job, relays rank 0's JSON line and returns the job's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")       # torchrun would otherwise pin every rank to one host thread
    proc = subprocess.Popen(launch_command(n, argv, free_port()), stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for line in proc.stdout:
        if line.startswith("{") and line.rstrip().endswith("}"):
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:                                      # anything else a rank prints is not part of the contract's one line
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, saw {lines}\n")
        rc = 1
    return rc


def dist_setup(n):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if n > 1 and world != n:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}: the launcher started a different number of ranks")
    # LV_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (RCCL refuses two
    # ranks on one device); ranks then share devices round-robin.  The judged runs use nccl (= RCCL).
    backend = os.environ.get("LV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        # Establish the ring's point-to-point channels before anything is timed (RCCL sets a pair's channel up on its
        # first send / recv, which would otherwise land in the first step when the run has no warm-up step).
        dev = torch.device("cuda", local)
        token, got = torch.full((8,), float(rank), device=dev), torch.empty((8,), device=dev)
        ops_ = [dist.P2POp(dist.isend, token, (rank + 1) % world), dist.P2POp(dist.irecv, got, (rank - 1) % world)]
        for req in dist.batch_isend_irecv(ops_):
            req.wait()
        torch.cuda.synchronize()
        if int(got[0].item()) != (rank - 1) % world:
            raise SystemExit(f"rank {rank}: ring neighbour check failed (got {got[0].item()})")
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x: float, world) -> float:
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def causal_fwd_flops(lengths, hq=None, d=D):
    hq = HQ if hq is None else hq
    return sum(2.0 * d * hq * float(L) * float(L) for L in lengths)


def cpu_baseline(seq, rows=2048, heads=(0, 1)):
    """The reference's eager attention path (oracle restatement of
    dot_product_attention.py:183-286) on ``rows`` query rows x all keys x
    ``heads`` query heads, extrapolated linearly in rows x heads: the eager path
    materialises the whole [sq, sk] score matrix before masking, so the cost per
    row does not depend on the row's position."""
    from oracle import attention_ref as A
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1)
    rows = max(1, min(rows, seq))
    heads = list(heads)
    kv_heads = [h // (HQ // HKV) for h in heads]
    row0 = (seq - rows) // 2
    q = torch.randn(rows, 1, len(heads), D, generator=g) * 0.5
    k = torch.randn(seq, 1, len(heads), D, generator=g) * 0.5     # K/V already expanded per query head,
    v = torch.randn(seq, 1, len(heads), D, generator=g) * 0.5     # as repeat_interleave leaves them
    mask = (torch.arange(seq)[None, :] > (torch.arange(rows)[:, None] + row0))[None, None]

    def run():
        return A.eager_attention(q, k, v, mask, math.sqrt(D))
    run()  # warm-up (thread pool, allocator)
    reps, total = 0, 0.0
    while total < 12.0 and reps < 16:      # about 10-20 s of CPU work, bounded
        t0 = time.perf_counter()
        run()
        total += time.perf_counter() - t0
        reps += 1
    dt = total / reps
    full = dt * (seq / rows) * (HQ / len(heads))
    return {"value": seq / full, "unit": "tokens/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{rows} query rows x {seq} keys x {len(heads)} of {HQ} heads, fp32 eager baddbmm/softmax/bmm, "
                      f"{reps} passes, {total:.1f}s measured, extrapolated linearly in rows x heads"}


def cpu_baseline_cfg1():
    """BASELINE configs[0] in full on the host cores: the eager path over 8,192 packed tokens in the five segments
    [258, 1290, 3000, 5120, 8192] (cumulative ends), all 40 query heads, one head at a time (the [40, 8192, 8192]
    score tensor would not fit at once)."""
    from oracle import attention_ref as A
    T, ends = 8192, [258, 1290, 3000, 5120, 8192]
    g = torch.Generator().manual_seed(0)
    seg = torch.zeros(T, dtype=torch.int64)
    for e in ends[:-1]:
        seg[e:] += 1
    pos = torch.arange(T)
    mask = ((seg[:, None] != seg[None, :]) | (pos[None, :] > pos[:, None]))[None, None]     # True = masked
    k = torch.randn(T, 1, HKV, D, generator=g)
    v = torch.randn(T, 1, HKV, D, generator=g)
    t0 = time.perf_counter()
    for h in range(HQ):
        q = torch.randn(T, 1, 1, D, generator=g)
        kh = h // (HQ // HKV)
        A.eager_attention(q, k[:, :, kh:kh + 1], v[:, :, kh:kh + 1], mask, math.sqrt(D))
    dt = time.perf_counter() - t0
    return {"tokens_per_s": T / dt, "seconds": dt, "what": "configs[0]: 8192 packed tokens, 5 segments, 40/8 heads, one layer, eager fp32, timed in full"}


def profiled_traffic(summary_name, kernel):
    """Memory-side bytes per launch of `kernel`, from the committed rocprofv3 PMC passes of this same
    command (profiles/<summary_name>; collected with tools_diag/prof.sh in separate --pmc runs).
    FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
    coalesced reads at 64 bytes, so it is doubled."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", summary_name)
    try:
        with open(path) as f:
            d = json.load(f)
        fetch = next(v["per_dispatch"] for k, v in d.items() if kernel in k and k.endswith("|FETCH_SIZE"))
        write = next(v["per_dispatch"] for k, v in d.items() if kernel in k and k.endswith("|WRITE_SIZE"))
    except (OSError, StopIteration, KeyError, ValueError):
        return None, None
    return (2.0 * fetch + write) * 1024.0, f"bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB from profiles/{summary_name}"


def main():
    global HQ, HKV
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.heads:
        HQ, HKV = (int(x) for x in args.heads.split(","))
    rank, world, local = dist_setup(args.gpus)
    from long_vita_amd import ops
    dev = torch.device("cuda", local)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)

    def rnd(*shape):
        return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * 0.5).to(torch.bfloat16)

    wl = args.workload
    extra = {}
    kernel_ms = []
    if wl == "prefill_1m":
        S = args.seq or 1048576
        cp = world
        s_loc = S // cp
        q, k, v = rnd(s_loc, HQ, D), rnd(s_loc, HKV, D), rnd(s_loc, HKV, D)
        scale = 1.0 / math.sqrt(D)
        if cp == 1:
            cu = [0, S]
            desc = ops.blocks_on_device(cu, True, False, dev)
            out = torch.empty_like(q)
            lse = torch.empty((HQ, S), dtype=torch.float32, device=dev)

            def step():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.attn_fwd_launch(q, k, v, desc, scale, True, out, lse)
                e1.record()
                return e0, e1
        else:
            from long_vita_amd.ring_context_parallel import RingAttention
            import torch.distributed as dist
            ring = RingAttention(dist.group.WORLD, cp, rank, causal=True)
            ring.profile = True     # three event records per ring step: where a rank's time goes shows in the bench line
            ring_steps = []

            def harvest():
                ring_steps.append(ring.step_stats())

            def step():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ring.forward(q.view(s_loc, 1, HQ * D), k.view(s_loc, 1, HKV * D), v.view(s_loc, 1, HKV * D), HQ, scale)
                e1.record()
                e1.synchronize()      # one host wait per step (a real layer would not have it; it costs microseconds of 1 s)
                harvest()
                return e0, e1
        tokens_per_step = S
        flops_per_step = causal_fwd_flops([S])
        flops_per_gpu_step = flops_per_step / cp
        config = {"workload": "prefill_1m: causal attention forward, single 1M-token segment" if not args.seq else f"prefill S={S}",
                  "seq_len": S, "q_heads": HQ, "kv_heads": HKV, "head_dim": D, "parallelism": f"cp{cp}", "layers_per_step": 1}
        scaling = "strong"
    elif wl in ("cfg3_fwd_bwd", "cfg3_packed"):
        S = args.seq or 131072
        if wl == "cfg3_packed":
            import numpy as np
            rs = np.random.RandomState(1234)
            lens = []
            while sum(lens) < S:
                lens.append(int(math.exp(rs.uniform(math.log(256), math.log(32768)))))
            cu = [0]
            for L in lens:
                cu.append(min(S, cu[-1] + L))
            cu[-1] = S
        else:
            cu = [0, S]
        q, k, v = (rnd(S, HQ, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True))
        do = rnd(S, HQ, D)
        scale = 1.0 / math.sqrt(D)

        def step():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            q.grad = k.grad = v.grad = None
            e0.record()
            out, _ = ops.varlen_attention(q, k, v, cu, scale, True)
            out.backward(do)
            e1.record()
            return e0, e1
        lens = [b - a for a, b in zip(cu[:-1], cu[1:])]
        tokens_per_step = S
        flops_per_step = 3.5 * causal_fwd_flops(lens)
        flops_per_gpu_step = flops_per_step
        config = {"workload": f"{wl}: varlen causal attention fwd+bwd, {len(lens)} segment(s)", "seq_len": S, "q_heads": HQ, "kv_heads": HKV,
                  "head_dim": D, "parallelism": "cp1"}
        scaling = "weak"
    elif wl == "rope_qk":  # rotary embedding of one layer's Q and K at 128K tokens, in place (bandwidth-bound)
        S = args.seq or 131072
        q, k = rnd(S, HQ, D), rnd(S, HKV, D)
        inv = 1.0 / (1000000 ** (torch.arange(0, D, 2, dtype=torch.float32, device=dev) / D))
        ang = torch.outer(torch.arange(S, device=dev, dtype=torch.float32), inv)
        ang = torch.cat((ang, ang), -1)
        cos_t, sin_t = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)

        def step():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.rope_launch(q, cos_t, sin_t, out=q)
            ops.rope_launch(k, cos_t, sin_t, out=k)
            e1.record()
            return e0, e1
        tokens_per_step = S
        hbm_bytes_per_step = 2.0 * S * (HQ + HKV) * D * 2 + 2.0 * 2 * S * D * 2   # x read + written; both tables once per tensor
        flops_per_step = flops_per_gpu_step = 0.0
        config = {"workload": "rope_qk: rotary embedding of Q [S,40,128] and K [S,8,128], in place", "seq_len": S}
        scaling = "weak"
    elif wl == "rope_fused":  # rotary embedding + causal forward at 128K: rotation of q inside the attention kernel vs as its own pass
        S = args.seq or 131072
        q, k, v = rnd(S, HQ, D), rnd(S, HKV, D), rnd(S, HKV, D)
        inv = 1.0 / (1000000 ** (torch.arange(0, D, 2, dtype=torch.float32, device=dev) / D))
        ang = torch.outer(torch.arange(S, device=dev, dtype=torch.float32), inv)
        ang = torch.cat((ang, ang), -1)
        cos_t, sin_t = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
        scale = 1.0 / math.sqrt(D)
        desc = ops.blocks_on_device([0, S], True, False, dev)
        out = torch.empty_like(q)
        lse = torch.empty((HQ, S), dtype=torch.float32, device=dev)
        q_rot, k_rot = torch.empty_like(q), torch.empty_like(k)

        def two_pass():
            ops.rope_launch(q, cos_t, sin_t, out=q_rot)
            ops.rope_launch(k, cos_t, sin_t, out=k_rot)
            ops.attn_fwd_launch(q_rot, k_rot, v, desc, scale, True, out, lse)

        def step():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.rope_launch(k, cos_t, sin_t, out=k_rot)
            ops.attn_fwd_rope_launch(q, k_rot, v, desc, scale, True, cos_t, sin_t, out=out, lse=lse)
            e1.record()
            return e0, e1
        two_pass()
        ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ta.record()
        for _ in range(5):
            two_pass()
        tb.record()
        torch.cuda.synchronize()
        extra["two_pass_ms"] = ta.elapsed_time(tb) / 5
        tokens_per_step = S
        flops_per_step = flops_per_gpu_step = causal_fwd_flops([S])
        config = {"workload": "rope_fused: rotary embedding of K (own pass) + causal forward that rotates Q as it loads it, one 128K segment",
                  "seq_len": S, "q_heads": HQ, "kv_heads": HKV, "head_dim": D, "parallelism": "cp1"}
        scaling = "weak"
    elif wl == "decode":  # one new token against a 1M-token KV cache, one layer (bandwidth-bound: the cache is read once)
        S = args.seq or 1048576
        q, kc, vc = rnd(1, HQ, D), rnd(1, S, HKV, D), rnd(1, S, HKV, D)
        scale = 1.0 / math.sqrt(D)

        def step():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(16):
                ops.decode_attention(q, kc, vc, scale)
            e1.record()
            return e0, e1
        tokens_per_step = 16
        hbm_bytes_per_step = 16 * 2.0 * S * HKV * D * 2
        flops_per_step = flops_per_gpu_step = 0.0
        config = {"workload": "decode: 16 incremental attention calls, 1 query token x KV cache [1, S, 8, 128] (K and V read once each)",
                  "seq_len": S, "q_heads": HQ, "kv_heads": HKV, "head_dim": D}
        scaling = "weak"
    elif wl == "vit_tower":  # the whole frozen InternViT-300M encoder on 64 tiles of 448 x 448 (random weights)
        from long_vita_amd import intern_vit as iv, vision_model as vmod
        B = 64
        cfg = iv.InternViTConfig()
        torch.manual_seed(1)
        vision = vmod.MegatronVisionModel(hidden_size=5120, vit_config=cfg, vision_projector_freeze=True).to(torch.bfloat16).to(dev)
        tower = vision.vit
        img = rnd(B, 3, cfg.img_h, cfg.img_w)

        def step():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            vision(images=img)      # tower -> class-token drop + pixel shuffle (256 tokens per frame) -> projector
            e1.record()
            return e0, e1
        Sv, hd, ffn = tower.seq_length, cfg.hidden_size, cfg.ffn_hidden_size
        tokens_per_step = B * Sv
        per_token = 2.0 * hd * (4 * hd + 2 * ffn) + 4.0 * Sv * hd
        flops_per_step = (cfg.num_layers * B * Sv * per_token + B * tower.num_patches * 2.0 * 3 * cfg.patch_dim ** 2 * hd
                          + B * 256 * 2.0 * (4 * hd * hd + hd * 5120))
        flops_per_gpu_step = flops_per_step
        config = {"workload": "vit_tower: InternViT-300M encoder on 64 frames of 448x448 (24 layers: GEMMs + attention + fused "
                              "elementwise), pixel shuffle to 256 tokens per frame, projector to 5120",
                  "frames": B, "seq_len": Sv, "hidden": hd, "layers": cfg.num_layers}
        scaling = "weak"
    else:  # cfg2_vit: 24 layers x 64 frames x 16 heads x 1025 tokens, d=64, non-causal
        B, H, Sv, Dv, layers = 64, 16, 1025, 64, 24
        qkv = [rnd(Sv, B * H, Dv) for _ in range(3)]
        out = torch.empty_like(qkv[0])
        lse = torch.empty((B * H, Sv), dtype=torch.float32, device=dev)
        scale = 1.0 / math.sqrt(Dv)

        def step():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(layers):
                ops.attn_fwd_dense_launch(qkv[0], qkv[1], qkv[2], scale, False, out, lse)
            e1.record()
            return e0, e1
        tokens_per_step = B * Sv
        flops_per_step = layers * B * H * 4.0 * Dv * Sv * Sv
        flops_per_gpu_step = flops_per_step
        config = {"workload": "cfg2_vit: InternViT-300M attention, 64 frames x 24 layers", "frames": B, "heads": H, "head_dim": Dv, "seq_len": Sv}
        scaling = "weak"

    for _ in range(args.warmup):
        step()
    if wl == "prefill_1m" and world > 1:
        ring_steps.clear()
    barrier(world)
    t0 = time.perf_counter()
    events = [step() for _ in range(args.steps)]
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    mean_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    per_rank = None
    if wl == "prefill_1m" and world > 1:
        # every rank's own view: device time per step, and per ring step the block kernels + merge against the time the
        # compute stream then stood waiting for the next K/V block (zig-zag imbalance and wire time show up there)
        import torch.distributed as dist
        n = max(1, len(ring_steps))
        mine = {"rank": rank, "device": torch.cuda.current_device(), "device_ms_per_step": mean_kernel_s * 1e3,
                "ring_step_compute_ms": [sum(st[i][0] for st in ring_steps) / n for i in range(cp)],
                "ring_step_wait_ms": [sum(st[i][1] for st in ring_steps) / n for i in range(cp)]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    mean_kernel_s = max_over_ranks(mean_kernel_s, world)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        achieved = flops_per_gpu_step / mean_kernel_s / 1e12
        traffic, traffic_note = None, None
        if wl == "prefill_1m" and not args.seq and world == 1:
            traffic, traffic_note = profiled_traffic("r03_prefill_1m_pmc_summary.json", "lv_fwd_a4_kernel<true>")
        res = {
            "metric": ("attention prefill tokens/sec (node) at 1M ctx" if not args.seq else f"attention prefill tokens/sec (node) at {S} ctx")
            if wl == "prefill_1m" else f"attention tokens/sec ({wl})",
            "value": tokens_per_step * args.steps / dt, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": config,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "note": "algorithmic FLOPs per GPU per step / mean device time per step (HIP events on the launch stream)"},
        }
        if wl in ("rope_qk", "decode"):
            gbs = hbm_bytes_per_step / mean_kernel_s / 1e9
            res["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                               "traffic": None, "note": "algorithmic bytes per step / mean device time per step"}
            if wl == "decode" and not args.seq and not args.heads:
                per_call, traffic_note = profiled_traffic("r01_decode_pmc_summary.json", "lv_decode_kernel<128>")
                res["roofline"]["traffic"] = per_call      # per launch of the dominant kernel (16 launches per step)
        if traffic_note:
            res["roofline"]["traffic_note"] = traffic_note
        res.update(extra)
        if per_rank is not None:
            import torch.distributed as dist
            res["n_ranks_seen"] = dist.get_world_size()
            res["per_rank"] = per_rank
        try:    # the clock the chip holds under a dense MFMA loop: the peak above assumes 2.4 GHz
            res["mfma_clock_mhz"] = ops.mfma_clock_mhz(dev)
            if res["roofline"]["bound"] == "mfma":
                # the same peak at the clock this chip holds under nothing but bf16 MFMAs on random data (its power limit): what
                # a kernel made of MFMAs alone would reach here; `frac` stays against the nominal 2.4 GHz peak
                sustained = PEAK_BF16_TFLOPS * res["mfma_clock_mhz"]["median"] / 2400.0
                res["roofline"]["peak_at_sustained_mfma_clock"] = sustained
                res["roofline"]["frac_of_sustained"] = res["roofline"]["achieved"] / sustained
        except Exception as e:      # diagnostic only
            res["mfma_clock_mhz"] = {"error": str(e)}
        res["host_cpus"] = os.cpu_count()
        if not args.no_cpu_baseline and wl == "prefill_1m" and world == 1:     # the host baseline is timed at N = 1 only
            res["cpu_baseline"] = cpu_baseline(config["seq_len"])
            if world == 1 and not args.seq:
                res["cpu_baseline"]["cfg1_full"] = cpu_baseline_cfg1()
        print(json.dumps(res), flush=True)
    barrier(world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
