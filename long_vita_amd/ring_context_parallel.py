"""Zig-zag causal ring attention over a context-parallel group.

Stands in for MindSpeed's ``ringattn_context_parallel`` (not in the reference
tree; called from lcvlm_modellink/core/transformer/dot_product_attention.py:70-71
with the ``cp_para`` dictionary built at :51-68).  Same call signature, same data
layout ([s_local, b, heads*dim], the local sequence being chunks ``r`` and
``2*CP-1-r`` of the 2*CP equal chunks, lcvlm_modellink/training/utils.py:352-361).

MI355X design: one process per GPU; every step posts ONE grouped RCCL
send/recv of the packed [2, s_local, b*kv_heads*dim] K/V buffer on a side HIP
stream (xGMI is point-to-point: a ring step uses one link per direction), and the
attention block kernel for the buffer already present runs meanwhile on the
compute stream.  Per step the work is one of three rectangles of the score matrix
  step 0                 : local causal block (2c x 2c, causal)
  KV from an earlier rank: all local queries x first KV chunk (2c x c, no mask)
  KV from a later rank   : second query chunk x whole KV block (c x 2c, no mask)
each worth the same FLOPs, merged into a running (out fp32, log-sum-exp) pair.
The backward pass sends the K/V block and the fp32 running sum of its dK/dV round the
same ring (each rank computes its contribution while the sum is still arriving, then adds
and forwards) and returns the finished sum to the block's owner with one extra hop.

The block arithmetic goes through a small backend object so that the schedule and
the communication can be exercised on CPU (gloo) in tests with a checker backend;
the default backend is the HIP library and has no CPU fallback.
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


class HipBackend:
    """Block attention / merge on the GPU through liblvattn."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def block_fwd(self, q3, k3, v3, qr, kr, causal, shift, scale, out_blk, lse_blk, rope=None):
        desc = self.ops.rect_blocks_on_device(qr[0], qr[1], kr[0], kr[1], causal, shift, False, q3.device)
        if rope is not None:    # unrotated queries, rotated in the kernel's prologue
            self.ops.attn_fwd_rope_launch(q3, k3, v3, desc, scale, causal, rope[0], rope[1], out=out_blk, lse=lse_blk,
                                          q_rot=rope[2] if len(rope) > 2 else None)
        else:
            self.ops.attn_fwd_launch(q3, k3, v3, desc, scale, causal, out_blk, lse_blk)

    def block_fwd_rects(self, q3, k3, v3, rects, scale, out_blk, lse_blk, rope=None):
        desc = self.ops.rects_blocks_on_device(rects, False, q3.device)
        if rope is not None:
            self.ops.attn_fwd_rope_launch(q3, k3, v3, desc, scale, True, rope[0], rope[1], out=out_blk, lse=lse_blk,
                                          q_rot=rope[2] if len(rope) > 2 else None)
        else:
            self.ops.attn_fwd_launch(q3, k3, v3, desc, scale, True, out_blk, lse_blk)

    def pack_kv(self, k3, v3, kv, rope=None):
        """kv [2, s, Hkv, d] := (k3, v3), the key half rotated on the way when ``rope`` = (cos, sin) tables is given."""
        if rope is not None:
            self.ops.rope_pack_kv(k3, v3, rope[0], rope[1], out=kv)
        else:
            kv[0].copy_(k3)
            kv[1].copy_(v3)

    def block_bwd_rects(self, q3, k3, v3, out3, dout3, lse, delta, rects, scale, dq_acc, dk_acc, dv_acc):
        desc_q = self.ops.rects_blocks_on_device(rects, False, q3.device)
        desc_k = self.ops.rects_blocks_on_device(rects, True, q3.device)
        self.ops.attn_bwd_launch(q3, k3, v3, out3, dout3, lse, desc_q, desc_k, scale, True, dq_acc, dk_acc, dv_acc, delta, True)

    def merge(self, out_acc, lse_acc, out_blk, lse_blk, row0, rows, init=False):
        self.ops.merge_launch(out_acc, lse_acc, out_blk, lse_blk, row0, rows, init)

    def finalize(self, out_acc, out=None):
        return self.ops.cast_f32_to_bf16(out_acc, out)

    def delta(self, out3, dout3, lse, delta=None):
        if delta is None:
            delta = torch.empty_like(lse)
        from ._lib import lib, check
        check(lib().lvattn_bwd_delta(out3.data_ptr(), dout3.data_ptr(), delta.data_ptr(), out3.shape[0], out3.shape[1],
                                     out3.shape[2], out3.stride(0), out3.stride(1), dout3.stride(0), dout3.stride(1),
                                     delta.stride(0), torch.cuda.current_stream().cuda_stream), "lvattn_bwd_delta")
        return delta

    def block_bwd(self, q3, k3, v3, out3, dout3, lse, delta, qr, kr, causal, shift, scale, dq_acc, dk_acc, dv_acc):
        desc_q = self.ops.rect_blocks_on_device(qr[0], qr[1], kr[0], kr[1], causal, shift, False, q3.device)
        desc_k = self.ops.rect_blocks_on_device(qr[0], qr[1], kr[0], kr[1], causal, shift, True, q3.device)
        self.ops.attn_bwd_launch(q3, k3, v3, out3, dout3, lse, desc_q, desc_k, scale, causal, dq_acc, dk_acc, dv_acc, delta, True)

    def cast(self, x_f32, dtype):
        if dtype == torch.bfloat16 and x_f32.dim() == 3:
            return self.ops.cast_f32_to_bf16(x_f32.contiguous())
        return x_f32.to(dtype)


class TorchDistComm:
    """Point-to-point ring traffic through torch.distributed (RCCL on GPUs, gloo on CPU), issued on side streams so that
    it overlaps the block kernels.

    ``overlap_group`` is the second communicator the reference sets up with --use-cp-send-recv-overlap
    (``cp_group_for_send_recv_overlap``, dot_product_attention.py:59-60): with it a rank's send and its receive go
    through DIFFERENT communicators, i.e. two RCCL kernels that progress side by side instead of one grouped call.  A
    message from ring rank a to a + 1 travels on ``group`` when a is even and on ``overlap_group`` when a is odd, so
    both ends of every pair agree (even ranks send on ``group`` and receive on ``overlap_group``, odd ranks the other
    way round); that needs an even ring size, an odd ring keeps to ``group``.

    ``lane`` selects the side stream: lane 0 carries the K / V blocks, lane 1 the fp32 running sums of dK / dV in the
    backward pass, so a 1 GiB sum in flight never delays the K / V block the next step's kernels wait for."""

    def __init__(self, group, device, overlap_group=None, ring_rank: int = 0, ring_size: int = 0):
        self.group = group
        self.overlap_group = overlap_group if (overlap_group is not None and ring_size % 2 == 0) else None
        self.ring_rank = int(ring_rank)
        self.cuda = device.type == "cuda"
        self.streams = [torch.cuda.Stream(device=device) for _ in range(2)] if self.cuda else None

    @property
    def stream(self):
        return self.streams[0] if self.cuda else None

    def _post(self, send_bufs, recv_bufs, dst, src):
        if self.overlap_group is None:
            p2p = [dist.P2POp(dist.isend, s, dst, self.group) for s in send_bufs]
            p2p += [dist.P2POp(dist.irecv, r, src, self.group) for r in recv_bufs]
            return dist.batch_isend_irecv(p2p)
        even = self.ring_rank % 2 == 0
        send_group, recv_group = (self.group, self.overlap_group) if even else (self.overlap_group, self.group)
        sends = [dist.P2POp(dist.isend, s, dst, send_group) for s in send_bufs]
        recvs = [dist.P2POp(dist.irecv, r, src, recv_group) for r in recv_bufs]
        reqs = []
        # each communicator sees its two ends post in the same relative order: on `group` the even rank's send meets
        # the odd rank's receive first, on the overlap group the odd rank's send meets the even rank's receive second
        for batch in ((sends, recvs) if even else (recvs, sends)):
            if batch:
                reqs += dist.batch_isend_irecv(batch)
        return reqs

    def exchange(self, send_bufs, recv_bufs, dst, src, lane: int = 0):
        """Post isend(send_bufs -> dst) and irecv(recv_bufs <- src); returns a handle for wait().  The traffic starts
        once everything already queued on the compute stream has finished (the receive buffer may still be in use
        there)."""
        if self.cuda:
            stream = self.streams[lane]
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                for r in self._post(send_bufs, recv_bufs, dst, src):
                    r.wait()          # stream-level wait on the side stream; the host does not block
                done = torch.cuda.Event()
                done.record(stream)
            return done
        return self._post(send_bufs, recv_bufs, dst, src)

    def wait(self, handle):
        if self.cuda:
            torch.cuda.current_stream().wait_event(handle)
        else:
            for r in handle:
                r.wait()


class RingAttention:
    """Engine for one context-parallel group.  ``rank`` is the rank inside the CP group;
    ``global_ranks[i]`` the global rank of CP rank i (defaults to 0..cp-1)."""

    def __init__(self, group, cp_size: int, rank: int, causal: bool = True, global_ranks=None, backend=None, comm=None,
                 overlap_group=None):
        self.group = group
        self.overlap_group = overlap_group
        self.cp = int(cp_size)
        self.rank = int(rank)
        self.causal = bool(causal)
        self.global_ranks = list(global_ranks) if global_ranks is not None else list(range(self.cp))
        self.backend = backend
        self.comm = comm
        self._ws: dict = {}        # pass ("fwd" / "bwd") -> (shape key, buffers): one set per pass, the latest shape
        self.profile = False       # record per-step device timestamps into self.timeline (forward)
        self.timeline: list = []

    def _buffers(self, key, make):
        """Per-engine workspace: the buffers of a pass are allocated the first time a shape is seen and reused by every
        later call with that shape (a caching allocator under RCCL traffic can synchronise; a steady-state step
        allocates nothing).  One set per pass is kept: a new shape (another sequence length) frees the old set before
        the new one is made, so what stays resident is at most one forward set (fp32 running output + one bf16 block
        output + two K / V buffers: 5.1 GB at 131,072 local tokens, 40 / 8 heads) and one backward set (fp32 dq, three
        fp32 K/V-shaped buffers, two K / V buffers: 7.0 GB).  ``release()`` frees them.  A result that leaves the engine
        is the caller's from then on, so those are allocated per call."""
        hit = self._ws.get(key[0])
        if hit is None or hit[0] != key:
            self._ws.pop(key[0], None)        # drop the old shape's set first: both need not be resident at once
            hit = self._ws[key[0]] = (key, make())
        return hit[1]

    def release(self):
        """Free the workspace (e.g. between a training phase and an evaluation at another sequence length, or before
        the engine goes idle); the next call allocates it again."""
        self._ws.clear()

    def workspace_bytes(self) -> int:
        def size(x):
            if isinstance(x, torch.Tensor):
                return x.numel() * x.element_size()
            if isinstance(x, (list, tuple)):
                return sum(size(y) for y in x)
            return 0
        return sum(size(v) for _, bufs in self._ws.values() for v in bufs.values())

    def step_stats(self):
        """After a profiled forward and a device synchronisation: milliseconds per ring step as a list of
        (block kernels + merge, compute stream stalled waiting for the K/V block of the next step)."""
        return [(a.elapsed_time(b), b.elapsed_time(c)) for a, b, c in self.timeline]

    # -- helpers -----------------------------------------------------------
    def _be(self):
        if self.backend is None:
            self.backend = HipBackend()
        return self.backend

    def _peers(self):
        nxt = self.global_ranks[(self.rank + 1) % self.cp]
        prv = self.global_ranks[(self.rank - 1) % self.cp]
        return nxt, prv

    def _comm(self, device):
        if self.comm is None:
            self.comm = TorchDistComm(self.group, device, self.overlap_group, self.rank, self.cp)
        return self.comm

    def _exchange(self, side, send_bufs, recv_bufs, dst, src, lane):
        if lane and isinstance(side, TorchDistComm):
            return side.exchange(send_bufs, recv_bufs, dst, src, lane)
        return side.exchange(send_bufs, recv_bufs, dst, src)     # test transports have one lane

    def step_rect(self, step: int, s_loc: int):
        """(q_rect, k_rect, causal, shift) of ring step ``step`` on a rank holding ``s_loc`` rows."""
        c = s_loc // 2
        src = (self.rank - step) % self.cp
        if not self.causal:
            return (0, s_loc), (0, s_loc), False, 0
        if step == 0:
            return (0, s_loc), (0, s_loc), True, 0
        if src < self.rank:
            return (0, s_loc), (0, c), False, 0
        return (c, c), (0, s_loc), False, 0

    def segment_rects(self, step: int, s_loc: int, cu_seqlens):
        """Packed segments under the zig-zag split.  ``cu_seqlens`` are the segment boundaries in GLOBAL token
        positions ([0, ..., S]).  For ring step ``step`` returns four lists, one per (query chunk, key chunk) pair of
        this rank's rows against the block of rank ``rank - step``: rectangles (q_row0, q_len, k_row0, k_len, shift)
        in LOCAL row / key indices, key j visible to row i iff j <= i + shift with shift = global position of the
        rectangle's first row - that of its first key (the causal rule on global positions; segments never mix).
        Inside one pair the rectangles share neither query rows nor key rows, and their query ranges abut."""
        c = s_loc // 2
        src = (self.rank - step) % self.cp
        qchunks = ((self.rank * c, 0), ((2 * self.cp - 1 - self.rank) * c, c))      # (global start, local start)
        kchunks = ((src * c, 0), ((2 * self.cp - 1 - src) * c, c))
        pairs = []
        for qg0, ql0 in qchunks:
            for kg0, kl0 in kchunks:
                rects = []
                if kg0 < qg0 + c:   # some key of the chunk is not after every query of the chunk
                    for s0, s1 in zip(cu_seqlens[:-1], cu_seqlens[1:]):
                        a0, a1 = max(s0, qg0), min(s1, qg0 + c)
                        b0, b1 = max(s0, kg0), min(s1, kg0 + c)
                        if a1 > a0 and b1 > b0 and b0 < a1:
                            rects.append((ql0 + a0 - qg0, a1 - a0, kl0 + b0 - kg0, b1 - b0, a0 - b0))
                pairs.append(rects)
        return pairs

    def _fwd_rects(self, be, q3, k3, v3, rects, scale, out_blk, lse_blk, rope=None):
        kw = {} if rope is None else {"rope": rope}
        if hasattr(be, "block_fwd_rects"):
            be.block_fwd_rects(q3, k3, v3, rects, scale, out_blk, lse_blk, **kw)
        else:
            for q0, ql, k0, kl, sh in rects:
                be.block_fwd(q3, k3, v3, (q0, ql), (k0, kl), True, sh, scale, out_blk, lse_blk, **kw)

    def _bwd_rects(self, be, q3, k3, v3, out3, dout3, lse, delta, rects, scale, dq_acc, dk_acc, dv_acc):
        if hasattr(be, "block_bwd_rects"):
            be.block_bwd_rects(q3, k3, v3, out3, dout3, lse, delta, rects, scale, dq_acc, dk_acc, dv_acc)
        else:
            for q0, ql, k0, kl, sh in rects:
                be.block_bwd(q3, k3, v3, out3, dout3, lse, delta, (q0, ql), (k0, kl), True, sh, scale, dq_acc, dk_acc, dv_acc)

    # -- forward -----------------------------------------------------------
    def forward(self, q, k, v, n_heads: int, scale: float | None = None, return_state: bool = False, cu_seqlens=None, rope=None,
                q_rot_out=None):
        """q [s, b, Hq*d], k/v [s, b, Hkv*d] (local zig-zag shard) -> out [s, b, Hq*d].  With ``cu_seqlens`` (global
        segment boundaries, b == 1) attention stays inside the packed segments.  With ``rope`` = (cos, sin), bf16 tables
        [s, d] for THIS rank's rows (the zig-zag slice of the table, rotary_pos_embedding.py:36-47; b == 1), q and k come
        UNROTATED: the keys are rotated while they are packed into the ring message, the queries inside the block
        kernels.  ``q_rot_out`` ([s, Hq, d] bf16, optional) receives the rotated queries from the launches of step 0,
        which between them visit every local row (what a backward pass needs)."""
        be = self._be()
        s_loc, b, hq_d = q.shape
        d = hq_d // n_heads
        hkv = k.shape[2] // d
        if scale is None:
            scale = 1.0 / math.sqrt(d)
        if self.causal and s_loc % 2:
            raise ValueError("zig-zag ring attention needs an even local sequence length")
        q3 = q.reshape(s_loc, b * n_heads, d)
        dev, cp = q.device, self.cp

        def make():
            w = {"kv": [torch.empty((2, s_loc, b * hkv, d), dtype=k.dtype, device=dev) for _ in range(2 if cp > 1 else 1)],
                 "out_acc": torch.empty((s_loc, b * n_heads, d), dtype=torch.float32, device=dev),
                 "out_blk": torch.empty((s_loc, b * n_heads, d), dtype=q.dtype, device=dev),
                 "lse_blk": torch.empty((b * n_heads, s_loc), dtype=torch.float32, device=dev)}
            return w
        w = self._buffers(("fwd", s_loc, b, n_heads, hkv, d, q.dtype, str(dev)), make)
        kv, out_acc, out_blk, lse_blk = w["kv"], w["out_acc"], w["out_blk"], w["lse_blk"]
        # K and V of this rank as ONE message [2, s, Hkv, d]
        if rope is not None:
            if b != 1:
                raise ValueError("the fused rotary embedding on the ring needs micro-batch 1")
            be.pack_kv(k.reshape(s_loc, hkv, d), v.reshape(s_loc, hkv, d), kv[0], rope)
        elif hasattr(be, "pack_kv"):
            be.pack_kv(k.reshape(s_loc, b * hkv, d), v.reshape(s_loc, b * hkv, d), kv[0])
        else:
            kv[0][0].copy_(k.reshape(s_loc, b * hkv, d))
            kv[0][1].copy_(v.reshape(s_loc, b * hkv, d))
        rkw = {} if rope is None else {"rope": rope}
        rope0 = rope if (rope is None or q_rot_out is None) else (rope[0], rope[1], q_rot_out)    # step 0 also writes q_rot
        lse_acc = torch.empty((b * n_heads, s_loc), dtype=torch.float32, device=dev)   # returned (saved for backward)
        packed = cu_seqlens is not None
        if packed:      # rows are first touched by different chunk pairs: start from the neutral element
            out_acc.zero_()
            lse_acc.fill_(float("-inf"))
        side = self._comm(dev)
        nxt, prv = self._peers()
        timeline = []
        for step in range(cp):
            cur = kv[step % 2] if cp > 1 else kv[0]
            handle = None
            if step + 1 < cp:
                handle = side.exchange([cur], [kv[(step + 1) % 2]], nxt, prv)
            if self.profile:
                t0 = torch.cuda.Event(enable_timing=True); t0.record()
            if not packed:
                qr, kr, causal, shift = self.step_rect(step, s_loc)
                be.block_fwd(q3, cur[0], cur[1], qr, kr, causal, shift, scale, out_blk, lse_blk,
                             **(rkw if step else ({} if rope0 is None else {"rope": rope0})))
                # step 0 covers every local row: it initialises the running result instead of merging into it
                be.merge(out_acc, lse_acc, out_blk, lse_blk, qr[0], qr[1], init=(step == 0))
            else:
                for rects in self.segment_rects(step, s_loc, cu_seqlens):
                    if rects:
                        self._fwd_rects(be, q3, cur[0], cur[1], rects, scale, out_blk, lse_blk, rope if step else rope0)
                        lo = min(r[0] for r in rects)
                        hi = max(r[0] + r[1] for r in rects)
                        be.merge(out_acc, lse_acc, out_blk, lse_blk, lo, hi - lo)
            if self.profile:
                t1 = torch.cuda.Event(enable_timing=True); t1.record()
            if handle is not None:
                side.wait(handle)
            if self.profile:
                t2 = torch.cuda.Event(enable_timing=True); t2.record()
                timeline.append((t0, t1, t2))
        if self.profile:
            self.timeline = timeline
        out = be.finalize(out_acc).reshape(s_loc, b, hq_d)
        if return_state:
            return out, lse_acc
        return out

    # -- backward ----------------------------------------------------------
    def backward(self, q, k, v, out, lse, dout, n_heads: int, scale: float | None = None, cu_seqlens=None):
        """Gradients (dq, dk, dv) in the layouts/dtypes of q, k, v."""
        be = self._be()
        s_loc, b, hq_d = q.shape
        d = hq_d // n_heads
        hkv = k.shape[2] // d
        if scale is None:
            scale = 1.0 / math.sqrt(d)
        q3 = q.reshape(s_loc, b * n_heads, d)
        out3 = out.reshape(s_loc, b * n_heads, d)
        dout3 = dout.reshape(s_loc, b * n_heads, d)
        dev, cp = q.device, self.cp

        def make():
            f32 = dict(dtype=torch.float32, device=dev)
            return {"kv": [torch.empty((2, s_loc, b * hkv, d), dtype=k.dtype, device=dev) for _ in range(2 if cp > 1 else 1)],
                    # dK/dV of a K/V block is the sum of one contribution per rank; the running sum follows the block
                    # round the ring.  Each step computes the local contribution into `dloc` while the sum for this
                    # block is still arriving from the previous rank, then adds and forwards: the fp32 accumulator
                    # traffic (twice the K/V bytes) overlaps the block kernels as the K/V traffic does.
                    "dloc": torch.empty((2, s_loc, b * hkv, d), **f32),
                    "dbuf": [torch.empty((2, s_loc, b * hkv, d), **f32) for _ in range(2)] if cp > 1 else [None, None],
                    "dq_acc": torch.empty((s_loc, b * n_heads, d), **f32),
                    "delta": torch.empty((b * n_heads, s_loc), **f32)}
        w = self._buffers(("bwd", s_loc, b, n_heads, hkv, d, q.dtype, str(dev)), make)
        kv, dloc, dbuf, dq_acc = w["kv"], w["dloc"], w["dbuf"], w["dq_acc"]
        delta = be.delta(out3, dout3, lse, w["delta"])
        kv[0][0].copy_(k.reshape(s_loc, b * hkv, d))
        kv[0][1].copy_(v.reshape(s_loc, b * hkv, d))
        dq_acc.zero_()
        side = self._comm(dev)
        nxt, prv = self._peers()
        pending = None   # exchange of the running sum started at the end of the previous step
        for step in range(cp):
            cur = kv[step % 2] if cp > 1 else kv[0]
            handle = None
            if step + 1 < cp:
                handle = side.exchange([cur], [kv[(step + 1) % 2]], nxt, prv)   # K/V travels ahead of the math
            if cu_seqlens is None:
                qr, kr, causal, shift = self.step_rect(step, s_loc)
                touched = dloc[:, kr[0]:kr[0] + kr[1]]     # the step writes these key rows only: clear and add just them
                touched.zero_()
                be.block_bwd(q3, cur[0], cur[1], out3, dout3, lse, delta, qr, kr, causal, shift, scale, dq_acc, dloc[0], dloc[1])
            else:
                kr = (0, s_loc)
                dloc.zero_()
                for rects in self.segment_rects(step, s_loc, cu_seqlens):   # one launch per chunk pair: no shared rows
                    if rects:
                        self._bwd_rects(be, q3, cur[0], cur[1], out3, dout3, lse, delta, rects, scale, dq_acc, dloc[0], dloc[1])
            if handle is not None:
                side.wait(handle)
            if cp > 1:
                send = dbuf[(step + 1) % 2]
                ks = slice(kr[0], kr[0] + kr[1])
                if pending is not None:
                    side.wait(pending)      # `send` now holds the sum so far for this block, from the previous rank
                    send[:, ks].add_(dloc[:, ks])
                else:                       # step 0 touches every key row of the local block
                    send.copy_(dloc)
                # after the last step this hop returns the finished sum to the block's owner
                pending = self._exchange(side, [send], [dbuf[step % 2]], nxt, prv, 1)    # its own side stream
        if cp > 1:
            side.wait(pending)
            final = dbuf[(cp - 1) % 2]
        else:
            final = dloc
        dq = be.cast(dq_acc, q.dtype).reshape(s_loc, b, hq_d)
        dk = be.cast(final[0], k.dtype).reshape(s_loc, b, hkv * d)
        dv = be.cast(final[1], v.dtype).reshape(s_loc, b, hkv * d)
        return dq, dk, dv


class _RingAttnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, engine, n_heads, scale, cu_seqlens=None):
        out, lse = engine.forward(q, k, v, n_heads, scale, return_state=True, cu_seqlens=cu_seqlens)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.engine, ctx.n_heads, ctx.scale, ctx.cu_seqlens = engine, n_heads, scale, cu_seqlens
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = ctx.engine.backward(q, k, v, out, lse, dout.contiguous(), ctx.n_heads, ctx.scale, cu_seqlens=ctx.cu_seqlens)
        return dq, dk, dv, None, None, None, None


def ring_rope_forward(engine, q, k, v, n_heads, scale, cos_tab, sin_tab, cu_seqlens=None, for_backward=True):
    """Ring forward on UNROTATED q / k [s, 1, heads*d] with the rotary embedding folded in (keys rotated while they are
    packed into the ring message, queries inside the block kernels).  Returns (out, lse, q_rot, k_rot); the last two
    (what the backward kernels read) only when ``for_backward``."""
    from . import ops
    s_loc, b, hq_d = q.shape
    if b != 1:
        raise ValueError("the fused rotary embedding on the ring needs micro-batch 1")
    d = hq_d // n_heads
    q_rot = torch.empty((s_loc, n_heads, d), dtype=q.dtype, device=q.device) if for_backward else None
    out, lse = engine.forward(q, k, v, n_heads, scale, return_state=True, cu_seqlens=cu_seqlens, rope=(cos_tab, sin_tab),
                              q_rot_out=q_rot)
    # the ring buffers are reused by the steps: the rotated keys are made once more for the backward (a bandwidth-bound
    # pass over K, a fifth of the pass over Q that the fusion removes)
    k_rot = ops.rope_launch(k.reshape(s_loc, -1, d), cos_tab, sin_tab) if for_backward else None
    return out, lse, q_rot, k_rot


def ring_rope_backward(engine, q_rot, k_rot, v, out, lse, dout, n_heads, scale, cos_tab, sin_tab, cu_seqlens=None):
    """Gradients w.r.t. the UNROTATED q, k (and v): the ring backward on the rotated operands, then the transposed
    rotation of dq and dk in place."""
    from . import ops
    s_loc, hq, d = q_rot.shape
    dq, dk, dv = engine.backward(q_rot.view(s_loc, 1, hq * d), k_rot.view(s_loc, 1, -1), v, out, lse, dout.contiguous(),
                                 n_heads, scale, cu_seqlens=cu_seqlens)
    dq3, dk3 = dq.view(s_loc, hq, d), dk.view(s_loc, -1, d)
    ops.rope_launch(dq3, cos_tab, sin_tab, conj=True, out=dq3)
    ops.rope_launch(dk3, cos_tab, sin_tab, conj=True, out=dk3)
    return dq, dk, dv


class _RingRopeAttnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, engine, n_heads, scale, cos_tab, sin_tab, cu_seqlens=None):
        need_bwd = any(ctx.needs_input_grad[:3])
        out, lse, q_rot, k_rot = ring_rope_forward(engine, q, k, v, n_heads, scale, cos_tab, sin_tab, cu_seqlens, need_bwd)
        if need_bwd:
            ctx.save_for_backward(q_rot, k_rot, v, out, lse, cos_tab, sin_tab)
        ctx.engine, ctx.n_heads, ctx.scale, ctx.cu_seqlens = engine, n_heads, scale, cu_seqlens
        return out

    @staticmethod
    def backward(ctx, dout):
        q_rot, k_rot, v, out, lse, cos_tab, sin_tab = ctx.saved_tensors
        dq, dk, dv = ring_rope_backward(ctx.engine, q_rot, k_rot, v, out, lse, dout, ctx.n_heads, ctx.scale, cos_tab, sin_tab,
                                        ctx.cu_seqlens)
        return dq, dk, dv, None, None, None, None, None, None


_engines: dict = {}


def _group_ranks(group):
    """What identifies a process group across its lifetime: the global ranks in it (an ``id()`` can be handed to a new
    object once the old group is gone)."""
    if group is None or not dist.is_available() or not dist.is_initialized():
        return None
    try:
        return tuple(dist.get_process_group_ranks(group))
    except Exception:       # a group from another backend wrapper: fall back to what the caller listed
        return None


def _engine_for(group, overlap, cp_size, rank, causal, ranks):
    """The engine of one context-parallel group, kept across calls for its workspace and side streams.  Keyed by the
    group's member ranks; an entry whose group objects are not the ones passed now (the group was destroyed and made
    again) is replaced, never reused."""
    key = (_group_ranks(group), cp_size, rank, causal, tuple(ranks) if ranks is not None else None)
    eng = _engines.get(key)
    if eng is not None and (eng.group is not group or eng.overlap_group is not overlap):
        eng.release()
        eng = None
    if eng is None:
        eng = RingAttention(group, cp_size, rank, causal=causal, global_ranks=ranks, overlap_group=overlap)
        _engines[key] = eng
    return eng


def ringattn_context_parallel(q, k, v, n, cp_para, softmax_scale=None, attn_mask=None, dropout_p=0.,
                              actual_seq_qlen=None, actual_seq_kvlen=None, rope=None):
    """Drop-in for mindspeed.core.context_parallel.ring_context_parallel.ringattn_context_parallel
    as called at dot_product_attention.py:70-71.

    ``cp_para`` keys read: causal, cp_group, cp_size, rank, cp_global_ranks, cp_group_for_send_recv_overlap (the second
    communicator of --use-cp-send-recv-overlap: sends and receives then use different communicators); (the double-ring
    window keys cp_inner_ranks / cp_outer_ranks / ... are accepted and must describe a single
    ring: one 8-GPU node has no inter-node window), pse must be None.

    ``rope`` = (cos, sin) (not in the reference's signature: its rotary embedding is applied by the caller, with the
    scripts' --use-fused-rotary-pos-emb by a fused NPU op): bf16 tables [s_local, d] for THIS rank's rows
    (rotary_pos_embedding.py:36-47); q and k then come UNROTATED and the rotation happens inside the ring
    (differentiable; results bit-identical to rotating first when both run the ping-pong forward kernel, within kernel
    rounding of it where the plain ring picks the one-wave kernel for long blocks)."""
    if dropout_p not in (0, 0.0):
        raise NotImplementedError("attention dropout is not supported on the ring path (the shipped scripts use 0)")
    if cp_para.get("pse") is not None:
        raise NotImplementedError("pse / alibi is not supported on the ring path")
    cu_seqlens = None
    if actual_seq_qlen is not None:
        # packed segments: cumulative END offsets over the WHOLE sequence (before the split over the ranks), as the
        # data pipeline sets them (lcvlm_modellink/training/utils.py:671-672)
        if cp_para.get("causal", True) is False:
            raise NotImplementedError("packed segments with a non-causal ring are not supported")
        if actual_seq_kvlen is not None and list(actual_seq_kvlen) != list(actual_seq_qlen):
            raise NotImplementedError("different q / kv segmentations are not supported")
        if q.shape[1] != 1:
            raise NotImplementedError("packed segments on the ring need micro-batch 1 (as every shipped script uses)")
        cu_seqlens = (0,) + tuple(int(x) for x in actual_seq_qlen)
        if cu_seqlens[-1] != q.shape[0] * int(cp_para["cp_size"]):
            raise ValueError(f"actual_seq_len ends at {cu_seqlens[-1]}, the sequence has {q.shape[0] * int(cp_para['cp_size'])} tokens")
    cp_size = int(cp_para["cp_size"])
    rank = int(cp_para["rank"])
    group = cp_para.get("cp_group")
    overlap = cp_para.get("cp_group_for_send_recv_overlap")
    ranks = cp_para.get("cp_global_ranks")
    causal = bool(cp_para.get("causal", True))
    eng = _engine_for(group, overlap, cp_size, rank, causal, ranks)
    d = q.shape[-1] // n
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d)
    if rope is not None:
        return _RingRopeAttnFunction.apply(q, k, v, eng, n, float(scale), rope[0], rope[1], cu_seqlens)
    return _RingAttnFunction.apply(q, k, v, eng, n, float(scale), cu_seqlens)


def merge_partial_attention(outs: torch.Tensor, lses: torch.Tensor) -> torch.Tensor:
    """outs [n, ..., d] fp32 (each normalised over its own keys), lses [n, ...] (natural log of its softmax
    denominator, -inf for an empty key set) -> attention over the union of the key sets, [..., d] fp32."""
    m = lses.max(dim=0).values
    m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
    w = torch.exp(lses - m)                      # exp(-inf) = 0 drops empty parts
    den = w.sum(dim=0)
    num = (w.unsqueeze(-1) * outs).sum(dim=0)
    return num / torch.where(den > 0, den, torch.ones_like(den)).unsqueeze(-1)


def decode_context_parallel(q, k_local, v_local, scale: float, group=None, cp_size: int = 1, partial=None):
    """One new token per sequence against a KV cache that is SHARDED over the context-parallel group (after a ring
    prefill every rank holds the keys / values of its own zig-zag chunks; which positions a rank holds does not
    matter, the new token sees them all).  q [b, hq, d] (the same on every rank), k_local / v_local [b, k_loc, hkv, d]
    -> out [b, hq, d].  Each rank runs the incremental kernel over its shard, then one all_gather of (out, lse)
    -- (d + 1) floats per head -- and a log-sum-exp merge.  The reference has no counterpart: under context
    parallelism its generation loop runs the whole prompt again for every token
    (tasks/inference/text_generation/utils.py:351-385).
    ``partial(q, k, v, scale) -> (out, lse)`` replaces the HIP kernel in CPU tests."""
    if partial is None:
        from . import ops

        def partial(q_, k_, v_, s_):
            return ops.decode_attention(q_, k_, v_, s_, return_lse=True)
    b, hq, d = q.shape
    if k_local.shape[1] > 0:
        out, lse = partial(q, k_local, v_local, scale)
    else:   # a rank may hold nothing yet
        out = torch.zeros((b, hq, d), dtype=q.dtype, device=q.device)
        lse = torch.full((b, hq), float("-inf"), dtype=torch.float32, device=q.device)
    if cp_size == 1:
        return out
    pack = torch.cat([out.float(), lse.float().unsqueeze(-1)], dim=-1).contiguous()
    parts = [torch.empty_like(pack) for _ in range(cp_size)]
    dist.all_gather(parts, pack, group=group)
    parts = torch.stack(parts)
    return merge_partial_attention(parts[..., :d], parts[..., d]).to(q.dtype)
