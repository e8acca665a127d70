"""Ring attention schedule + transport on CPU: virtual ranks over a loopback fabric,
and real torch.distributed (gloo, world_size 2) processes.  Block arithmetic is done
by the oracle-backed checker backend, so what is under test is the zig-zag schedule,
the buffer rotation, the merge order and the backward's accumulator hand-off."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import attention_ref as A
from tests.ring_helpers import (CheckerBackend, LoopbackComm, LoopbackFabric, run_virtual_ranks, unzigzag, zigzag_local)


def _problem(S, Hq, Hkv, D, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(S, Hq, D, generator=g)
    k = torch.randn(S, Hkv, D, generator=g)
    v = torch.randn(S, Hkv, D, generator=g)
    do = torch.randn(S, Hq, D, generator=g)
    return q, k, v, do


@pytest.mark.parametrize("cp", [1, 2, 4])
def test_ring_virtual_ranks_forward_backward(cp):
    from long_vita_amd.ring_context_parallel import RingAttention
    S, Hq, Hkv, D = 64, 4, 2, 8
    q, k, v, do = _problem(S, Hq, Hkv, D)
    scale = 0.3
    ref_o, ref_lse = A.varlen_attention(q, k, v, [0, S], scale)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, [0, S], scale)
    fabric = LoopbackFabric(cp)

    def body(r):
        eng = RingAttention(None, cp, r, causal=True, backend=CheckerBackend(), comm=LoopbackComm(fabric, r))
        ql, kl, vl, dl = (zigzag_local(x, cp, r).reshape(S // cp, 1, -1) for x in (q, k, v, do))
        out, lse = eng.forward(ql, kl, vl, Hq, scale, return_state=True)
        dq, dk, dv = eng.backward(ql, kl, vl, out, lse, dl, Hq, scale)
        return out[:, 0], lse, dq[:, 0], dk[:, 0], dv[:, 0]
    res = run_virtual_ranks(cp, body)
    out = unzigzag([r[0] for r in res], cp).reshape(S, Hq, D)
    np.testing.assert_allclose(out.numpy(), ref_o, atol=2e-5)
    lse = unzigzag([r[1].t() for r in res], cp).t()
    np.testing.assert_allclose(lse.numpy(), ref_lse, atol=2e-5)
    for idx, ref, H in ((2, rdq, Hq), (3, rdk, Hkv), (4, rdv, Hkv)):
        got = unzigzag([r[idx] for r in res], cp).reshape(S, H, D)
        np.testing.assert_allclose(got.numpy(), ref, atol=1e-4)


@pytest.mark.parametrize("cp", [2, 4])
def test_ring_virtual_ranks_full_attention(cp):
    """cp_attention_mask_type other than 'causal': every rank attends to every block (no zig-zag needed, any
    equal split works); forward and backward against the oracle's dense attention."""
    from long_vita_amd.ring_context_parallel import RingAttention
    S, Hq, Hkv, D = 48, 4, 2, 8
    q, k, v, do = _problem(S, Hq, Hkv, D, seed=5)
    scale = 0.3
    ref_o, ref_lse = A.varlen_attention(q, k, v, [0, S], scale, causal=False)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, [0, S], scale, causal=False)
    fabric = LoopbackFabric(cp)
    n = S // cp

    def body(r):
        eng = RingAttention(None, cp, r, causal=False, backend=CheckerBackend(), comm=LoopbackComm(fabric, r))
        ql, kl, vl, dl = (x[r * n:(r + 1) * n].reshape(n, 1, -1) for x in (q, k, v, do))
        out, lse = eng.forward(ql, kl, vl, Hq, scale, return_state=True)
        dq, dk, dv = eng.backward(ql, kl, vl, out, lse, dl, Hq, scale)
        return out[:, 0], dq[:, 0], dk[:, 0], dv[:, 0]
    res = run_virtual_ranks(cp, body)
    for idx, ref, H in ((0, ref_o, Hq), (1, rdq, Hq), (2, rdk, Hkv), (3, rdv, Hkv)):
        got = torch.cat([r[idx] for r in res]).reshape(S, H, D)
        np.testing.assert_allclose(got.numpy(), ref, atol=1e-4)


@pytest.mark.parametrize("cp,cu", [(2, [0, 5, 6, 30, 64]), (4, [0, 17, 40, 41, 100, 128]), (2, [0, 64]), (4, [0, 16, 32, 48, 64, 80, 96, 112, 128])])
def test_ring_virtual_ranks_packed_segments(cp, cu):
    """Packed segments under the zig-zag split (the reference's reset-attention-mask + CP branch): forward and backward
    over virtual ranks against the oracle's varlen attention with the same global segment boundaries."""
    from long_vita_amd.ring_context_parallel import RingAttention
    S, Hq, Hkv, D = cu[-1], 4, 2, 8
    q, k, v, do = _problem(S, Hq, Hkv, D, seed=7)
    scale = 0.3
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale)
    fabric = LoopbackFabric(cp)

    def body(r):
        eng = RingAttention(None, cp, r, causal=True, backend=CheckerBackend(), comm=LoopbackComm(fabric, r))
        ql, kl, vl, dl = (zigzag_local(x, cp, r).reshape(S // cp, 1, -1) for x in (q, k, v, do))
        out, lse = eng.forward(ql, kl, vl, Hq, scale, return_state=True, cu_seqlens=tuple(cu))
        dq, dk, dv = eng.backward(ql, kl, vl, out, lse, dl, Hq, scale, cu_seqlens=tuple(cu))
        return out[:, 0], lse, dq[:, 0], dk[:, 0], dv[:, 0]
    res = run_virtual_ranks(cp, body)
    out = unzigzag([r[0] for r in res], cp).reshape(S, Hq, D)
    np.testing.assert_allclose(out.numpy(), ref_o, atol=2e-5)
    lse = unzigzag([r[1].t() for r in res], cp).t()
    np.testing.assert_allclose(lse.numpy(), ref_lse, atol=2e-5)
    for idx, ref, H in ((2, rdq, Hq), (3, rdk, Hkv), (4, rdv, Hkv)):
        got = unzigzag([r[idx] for r in res], cp).reshape(S, H, D)
        np.testing.assert_allclose(got.numpy(), ref, atol=1e-4)


def test_segment_rects_cover_each_visible_pair_once():
    """Every (query, key) pair of the same segment with key <= query lies in exactly one rectangle over all ranks and
    steps, and no other pair does."""
    from long_vita_amd.ring_context_parallel import RingAttention
    for cp, cu in ((2, [0, 5, 6, 30, 64]), (4, [0, 17, 40, 41, 100, 128]), (8, [0, 1, 100, 256])):
        S = cu[-1]
        c = S // (2 * cp)
        seg = np.zeros(S, dtype=int)
        for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
            seg[a:b] = i
        want = (seg[:, None] == seg[None, :]) & (np.arange(S)[None, :] <= np.arange(S)[:, None])
        cover = np.zeros((S, S), dtype=int)
        for r in range(cp):
            eng = RingAttention(None, cp, r)
            gq = np.concatenate([np.arange(r * c, (r + 1) * c), np.arange((2 * cp - 1 - r) * c, (2 * cp - r) * c)])
            for step in range(cp):
                src = (r - step) % cp
                gk = np.concatenate([np.arange(src * c, (src + 1) * c), np.arange((2 * cp - 1 - src) * c, (2 * cp - src) * c)])
                for rects in eng.segment_rects(step, 2 * c, cu):
                    rows = [x for q0, ql, *_ in rects for x in range(q0, q0 + ql)]
                    keys = [x for _, _, k0, kl, _ in rects for x in range(k0, k0 + kl)]
                    assert len(rows) == len(set(rows)) and len(keys) == len(set(keys))   # one launch: no shared rows
                    for q0, ql, k0, kl, sh in rects:
                        for i in range(ql):
                            for j in range(min(kl, max(0, i + sh + 1))):
                                cover[gq[q0 + i], gk[k0 + j]] += 1
        assert (cover == want.astype(int)).all(), (cp, cu)


def test_segment_rects_exact_cover_on_random_packings():
    """The same exact-cover property on seeded random packings: context-parallel sizes 2 / 4 / 8, chunk lengths 1 .. 12,
    segment boundaries anywhere (on chunk edges, one-token segments, segments spanning many chunks, a single segment)."""
    from long_vita_amd.ring_context_parallel import RingAttention
    rng = np.random.default_rng(20240611)
    for trial in range(60):
        cp = int(rng.choice([2, 4, 8]))
        c = int(rng.integers(1, 13))
        S = 2 * cp * c
        n_cut = int(rng.integers(0, min(S - 1, 9) + 1))
        cuts = sorted(set(int(x) for x in rng.choice(np.arange(1, S), size=n_cut, replace=False))) if n_cut and S > 1 else []
        if rng.random() < 0.3 and S > 2 * c:          # a boundary exactly on a chunk edge
            cuts = sorted(set(cuts + [int(rng.integers(1, 2 * cp)) * c]))
        cu = [0] + cuts + [S]
        seg = np.zeros(S, dtype=int)
        for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
            seg[a:b] = i
        want = ((seg[:, None] == seg[None, :]) & (np.arange(S)[None, :] <= np.arange(S)[:, None])).astype(int)
        cover = np.zeros((S, S), dtype=int)
        for r in range(cp):
            eng = RingAttention(None, cp, r)
            gq = np.concatenate([np.arange(r * c, (r + 1) * c), np.arange((2 * cp - 1 - r) * c, (2 * cp - r) * c)])
            for step in range(cp):
                src = (r - step) % cp
                gk = np.concatenate([np.arange(src * c, (src + 1) * c), np.arange((2 * cp - 1 - src) * c, (2 * cp - src) * c)])
                for rects in eng.segment_rects(step, 2 * c, cu):
                    rows = [x for q0, ql, *_ in rects for x in range(q0, q0 + ql)]
                    keys = [x for _, _, k0, kl, _ in rects for x in range(k0, k0 + kl)]
                    assert len(rows) == len(set(rows)) and len(keys) == len(set(keys)), (cp, c, cu)
                    for q0, ql, k0, kl, sh in rects:
                        assert ql > 0 and kl > 0 and 0 <= q0 and q0 + ql <= 2 * c and 0 <= k0 and k0 + kl <= 2 * c, (cp, c, cu)
                        for i in range(ql):
                            for j in range(min(kl, max(0, i + sh + 1))):
                                cover[gq[q0 + i], gk[k0 + j]] += 1
        assert (cover == want).all(), (trial, cp, c, cu)


def _gloo_worker(rank, world, port, S, Hq, Hkv, D, scale, out_dir, overlap=False):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from long_vita_amd.ring_context_parallel import ringattn_context_parallel, _engines, _engine_for, RingAttention
    # the second communicator of --use-cp-send-recv-overlap (dot_product_attention.py:59-60): sends and receives split
    over_group = dist.new_group(list(range(world))) if overlap else None
    q, k, v, do = _problem(S, Hq, Hkv, D)
    ql, kl, vl, dl = (zigzag_local(x, world, rank).reshape(S // world, 1, -1).clone() for x in (q, k, v, do))
    ql.requires_grad_(True); kl.requires_grad_(True); vl.requires_grad_(True)
    cp_para = dict(causal=True, cp_group=dist.group.WORLD, cp_size=world, rank=rank, cp_global_ranks=list(range(world)),
                   cp_group_for_send_recv_overlap=over_group, pse=None, pse_type=None)
    # inject the checker backend into the engine the public entry point will create
    eng = _engine_for(dist.group.WORLD, over_group, world, rank, True, list(range(world)))
    eng.backend = CheckerBackend()
    assert list(_engines) == [(tuple(range(world)), world, rank, True, tuple(range(world)))]
    out = ringattn_context_parallel(ql, kl, vl, Hq, cp_para, scale, None, 0.)
    out.backward(dl)
    assert len(_engines) == 1 and _engine_for(dist.group.WORLD, over_group, world, rank, True, list(range(world))) is eng
    assert (eng._comm(ql.device).overlap_group is over_group) and eng.workspace_bytes() > 0
    # a group made again with the same ranks must not inherit the engine of the old one
    other = dist.new_group(list(range(world)))
    assert _engine_for(other, None, world, rank, True, list(range(world))) is not eng and eng.workspace_bytes() == 0
    torch.save({"out": out.detach(), "dq": ql.grad, "dk": kl.grad, "dv": vl.grad}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True], ids=["one_group", "send_recv_overlap_group"])
def test_ring_gloo_world2(tmp_path, overlap):
    S, Hq, Hkv, D, scale, world = 48, 4, 2, 8, 0.25, 2
    port = 29500 + (os.getpid() % 2000) + (2000 if overlap else 0)
    mp.spawn(_gloo_worker, args=(world, port, S, Hq, Hkv, D, scale, str(tmp_path), overlap), nprocs=world, join=True)
    q, k, v, do = _problem(S, Hq, Hkv, D)
    ref_o, _ = A.varlen_attention(q, k, v, [0, S], scale)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, [0, S], scale)
    res = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=True) for r in range(world)]
    for name, ref, H in (("out", ref_o, Hq), ("dq", rdq, Hq), ("dk", rdk, Hkv), ("dv", rdv, Hkv)):
        got = unzigzag([r[name][:, 0] for r in res], world).reshape(S, H, D)
        np.testing.assert_allclose(got.numpy(), ref, atol=1e-4, err_msg=name)


def test_step_rectangles_cover_causal_triangle():
    """Every (query chunk, key chunk) pair with key <= query is computed exactly once."""
    from long_vita_amd.ring_context_parallel import RingAttention
    for cp in (1, 2, 4, 8):
        c = 4
        cover = np.zeros((2 * cp, 2 * cp), dtype=int)
        for r in range(cp):
            eng = RingAttention(None, cp, r)
            own = [r, 2 * cp - 1 - r]
            for step in range(cp):
                src = (r - step) % cp
                theirs = [src, 2 * cp - 1 - src]
                (q0, ql), (k0, kl), causal, _ = eng.step_rect(step, 2 * c)
                for qi in range(q0 // c, (q0 + ql) // c):
                    for ki in range(k0 // c, (k0 + kl) // c):
                        if causal and ki > qi:
                            continue
                        cover[own[qi], theirs[ki]] += 1
        assert (cover == np.tril(np.ones_like(cover))).all(), cp


def _oracle_partial(q, k, v, scale):
    outs, lses = [], []
    for i in range(q.shape[0]):
        o, l = A.block_attention(q[i][None], k[i], v[i], scale, None)
        outs.append(torch.from_numpy(o[0]).float())
        lses.append(torch.from_numpy(l[:, 0]).float())
    return torch.stack(outs), torch.stack(lses)


def _gloo_decode_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from long_vita_amd.ring_context_parallel import decode_context_parallel
    g = torch.Generator().manual_seed(3)
    b, hq, hkv, d, S = 2, 4, 2, 8, 40
    q, k, v = torch.randn(b, hq, d, generator=g), torch.randn(b, S, hkv, d, generator=g), torch.randn(b, S, hkv, d, generator=g)
    # the zig-zag shard of the cache this rank holds after a ring prefill
    idx = torch.cat([torch.arange(S).view(2 * world, -1)[rank], torch.arange(S).view(2 * world, -1)[2 * world - 1 - rank]])
    out = decode_context_parallel(q, k[:, idx], v[:, idx], 0.3, dist.group.WORLD, world, partial=_oracle_partial)
    # second call: rank 1 holds nothing (an empty shard must drop out of the merge)
    kk = k if rank == 0 else k[:, :0]
    vv = v if rank == 0 else v[:, :0]
    out2 = decode_context_parallel(q, kk, vv, 0.3, dist.group.WORLD, world, partial=_oracle_partial)
    torch.save({"out": out, "out2": out2}, os.path.join(out_dir, f"d{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_decode_over_sharded_cache_gloo_world2(tmp_path):
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gloo_decode_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(3)
    b, hq, hkv, d, S = 2, 4, 2, 8, 40
    q, k, v = torch.randn(b, hq, d, generator=g), torch.randn(b, S, hkv, d, generator=g), torch.randn(b, S, hkv, d, generator=g)
    ref, _ = _oracle_partial(q, k, v, 0.3)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"d{r}.pt"), weights_only=True)
        np.testing.assert_allclose(res["out"].numpy(), ref.numpy(), atol=2e-5)
        np.testing.assert_allclose(res["out2"].numpy(), ref.numpy(), atol=2e-5)


def _gloo_vision_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from long_vita_amd.vision_parallel import encode_images_context_parallel, images_needed_by_rank
    S, T, n_img = 256, 8, 7
    g = torch.Generator().manual_seed(9)
    images = torch.randn(n_img, 3, 4, 4, generator=g)
    # a video at the front of the sample: consecutive 10-token frames (T context tokens each), one straddling a chunk edge
    starts = torch.tensor([1, 11, 21, 27, 60, 61 + 64, 200])
    ind_s = starts[:, None] + torch.arange(T)[None]
    ext = torch.stack([torch.zeros_like(ind_s), ind_s])
    calls = []

    def encode(x):
        calls.append(x.shape[0])
        return x.reshape(x.shape[0], 1, -1)[:, :, :6].repeat(1, T, 1) * 2.0 + 1.0   # [n, T, 6], a pure function of the image
    out = encode_images_context_parallel(images, ext, S, world, rank, encode, dist.group.WORLD)
    need = images_needed_by_rank(ext, S, world, rank)
    torch.save({"out": out, "need": need, "calls": torch.tensor(calls), "ref": encode(images[need])}, os.path.join(out_dir, f"v{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_image_encode_dealt_over_cp_ranks_gloo_world2(tmp_path):
    world = 2
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_gloo_vision_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    total = 0
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"v{r}.pt"), weights_only=True)
        assert torch.equal(res["out"], res["ref"])          # what the reference's per-rank encode would hold
        assert int(res["calls"][0]) in (3, 4)               # 7 images dealt 4 + 3, whatever each rank needs
        total += int(res["calls"][0])
    assert total == 7
