"""Test-only helpers for the ring attention engine: an oracle-backed block backend
(CPU, float64 arithmetic) and an in-process loopback transport that lets several
virtual ranks run as threads of one process."""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

from oracle import attention_ref as A


class CheckerBackend:
    """Same interface as long_vita_amd.ring_context_parallel.HipBackend, arithmetic by the oracle."""

    @staticmethod
    def _mask(qr, kr, causal, shift):
        if not causal:
            return None
        i = np.arange(qr[1])[:, None]
        j = np.arange(kr[1])[None, :]
        return j > i + shift

    def block_fwd(self, q3, k3, v3, qr, kr, causal, shift, scale, out_blk, lse_blk):
        q = q3[qr[0]:qr[0] + qr[1]]
        k = k3[kr[0]:kr[0] + kr[1]]
        v = v3[kr[0]:kr[0] + kr[1]]
        o, l = A.block_attention(q, k, v, scale, self._mask(qr, kr, causal, shift))
        out_blk[qr[0]:qr[0] + qr[1]] = torch.from_numpy(o).to(out_blk.dtype)
        lse_blk[:, qr[0]:qr[0] + qr[1]] = torch.from_numpy(l).to(lse_blk.dtype)

    def merge(self, out_acc, lse_acc, out_blk, lse_blk, row0, rows, init=False):
        sl = slice(row0, row0 + rows)
        if init:        # the accumulators hold nothing yet (and may hold anything)
            out_acc[sl] = out_blk[sl].to(out_acc.dtype)
            lse_acc[:, sl] = lse_blk[:, sl]
            return
        o, l = A.merge_blocks(out_acc[sl], lse_acc[:, sl], out_blk[sl], lse_blk[:, sl])
        out_acc[sl] = torch.from_numpy(o).to(out_acc.dtype)
        lse_acc[:, sl] = torch.from_numpy(l).to(lse_acc.dtype)

    def finalize(self, out_acc, out=None):
        return out_acc.clone()

    def delta(self, out3, dout3, lse, delta=None):
        return (out3.double() * dout3.double()).sum(-1).t().contiguous().to(lse.dtype)

    def block_bwd(self, q3, k3, v3, out3, dout3, lse, delta, qr, kr, causal, shift, scale, dq_acc, dk_acc, dv_acc):
        qs, ks = slice(qr[0], qr[0] + qr[1]), slice(kr[0], kr[0] + kr[1])
        q, k, v, do = (x.double().numpy() for x in (q3[qs], k3[ks], v3[ks], dout3[qs]))
        L = lse[:, qs].double().numpy()
        dl = delta[:, qs].double().numpy()
        mask = self._mask(qr, kr, causal, shift)
        Hq, Hkv = q.shape[1], k.shape[1]
        rep = Hq // Hkv
        for h in range(Hq):
            g = h // rep
            s = q[:, h] @ k[:, g].T * scale
            p = np.exp(s - L[h][:, None])
            if mask is not None:
                p = np.where(mask, 0.0, p)
            dv_acc[ks, g] += torch.from_numpy(p.T @ do[:, h]).to(dv_acc.dtype)
            ds = p * (do[:, h] @ v[:, g].T - dl[h][:, None]) * scale
            dq_acc[qs, h] += torch.from_numpy(ds @ k[:, g]).to(dq_acc.dtype)
            dk_acc[ks, g] += torch.from_numpy(ds.T @ q[:, h]).to(dk_acc.dtype)

    def cast(self, x, dtype):
        return x.to(dtype)


class LoopbackFabric:
    """Mailboxes shared by the virtual ranks of one process.  Like RCCL / gloo point-to-point traffic,
    the n-th receive a rank POSTS from a peer matches the n-th send that peer posted to it, whatever
    order the receives are waited for in."""

    def __init__(self, n):
        self.n = n
        self.lock = threading.Lock()
        self.boxes = {}

    def box(self, src, dst, seq):
        with self.lock:
            return self.boxes.setdefault((src, dst, seq), queue.Queue())


class LoopbackComm:
    def __init__(self, fabric: LoopbackFabric, me: int):
        self.f, self.me = fabric, me
        self.sent = {}
        self.posted = {}

    def exchange(self, send_bufs, recv_bufs, dst, src):
        for s in send_bufs:
            n = self.sent.get(dst, 0)
            self.sent[dst] = n + 1
            self.f.box(self.me, dst, n).put(s.clone())
        tickets = []
        for r in recv_bufs:
            n = self.posted.get(src, 0)
            self.posted[src] = n + 1
            tickets.append((r, n))
        return (tickets, src)

    def wait(self, handle):
        tickets, src = handle
        for r, n in tickets:
            r.copy_(self.f.box(src, self.me, n).get(timeout=120))


def run_virtual_ranks(n, fn):
    """Run fn(rank) on n threads; re-raise the first exception."""
    res, err = [None] * n, []

    def body(r):
        try:
            res[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=body, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    if err:
        raise err[0]
    return res


def zigzag_local(x: torch.Tensor, cp: int, r: int) -> torch.Tensor:
    c = x.shape[0] // (2 * cp)
    return torch.cat([x[r * c:(r + 1) * c], x[(2 * cp - 1 - r) * c:(2 * cp - r) * c]], dim=0)


def unzigzag(parts, cp: int) -> torch.Tensor:
    c = parts[0].shape[0] // 2
    chunks = [None] * (2 * cp)
    for r, p in enumerate(parts):
        chunks[r] = p[:c]
        chunks[2 * cp - 1 - r] = p[c:]
    return torch.cat(chunks, dim=0)
