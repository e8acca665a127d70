"""Gradients at the sizes the benchmarks run, checked PER ELEMENT against float64 on sampled rows.

The small parity cases compare whole tensors with the float64 oracle; at 131,072 and 1,048,576 tokens that is out of
reach, but one row of dq, dk or dv is cheap at any length (oracle/sampled_rows.py, pinned on the CPU against the full
float64 backward).  The long launches are where the kernels differ from the small ones: staging-ring slots that wrap
thousands of times, a thousand key blocks per head, 64-bit offsets (1M rows x 5120 elements is past 2^32).  Sampled are
the first / a middle / the last key block, block edges, segment edges and rows inside a last partial block.
Bound per row: |got - ref| <= 2e-2 |ref| + 5e-3 max|ref_row| (``row_excess`` <= 1)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention_ref as A
from oracle import sampled_rows as R
from tests.ring_helpers import LoopbackComm, LoopbackFabric, run_virtual_ranks, zigzag_local

REL_TOL = 2e-2
LSE_TOL = 1e-3


def _rnd(dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return lambda *shape: (torch.randn(*shape, generator=g, device=dev) * 0.5).to(torch.bfloat16)


def _two_bounds(got, exact, saved, tensor_max):
    """(tight, stated) excess of one gradient row.  tight: against float64 with delta from the saved bf16 output (what a
    fused backward reads), floor relative to the ROW's own largest element.  stated: against the exact float64 gradient
    under the project's per-element tolerance, whose absolute floor is relative to the TENSOR's largest element
    (5e-3 max|tensor|; tests/test_gpu_attention.py ``elementwise_excess``)."""
    return (R.row_excess(got, saved, tiny=1e-4 * tensor_max), R.row_excess(got, exact, floor=0.0, tiny=5e-3 * tensor_max))


def _check_rows(tag, q, k, v, do, out, lse, dq, dk, dv, seg, groups, q_rows, k_rows, scale, stats_from=None):
    """Sampled rows of one causal segment against float64.  Returns the largest excess seen per tensor and bound."""
    a, b = seg
    rep = q.shape[1] // k.shape[1]
    worst = {"dq": [0.0, 0.0], "dk": [0.0, 0.0], "dv": [0.0, 0.0], "lse": 0.0}
    tmax = {"dq": float(dq.float().abs().max()), "dk": float(dk.float().abs().max()), "dv": float(dv.float().abs().max())}

    def note(name, row, pair):
        if max(pair) > 0.7:
            print(f"  {tag}: {name} row {row}: excess tight {pair[0]:.3f} stated {pair[1]:.3f}", flush=True)
        worst[name] = [max(x, y) for x, y in zip(worst[name], pair)]
    for g in groups:
        heads = list(range(g * rep, g * rep + rep))
        for i in q_rows:
            for h in (heads[0], heads[-1]):
                exact, l, _ = R.dq_row(q, k, v, do, i, h, seg, scale)
                saved, _, _ = R.dq_row(q, k, v, do, i, h, seg, scale, out_saved=out)
                note("dq", (i, h), _two_bounds(dq[i, h], exact, saved, tmax["dq"]))
                worst["lse"] = max(worst["lse"], abs(float(lse[h, i]) - float(l)))
        r0 = a if stats_from is None else stats_from
        st_lse, st_delta = R.row_stats(q, k, v, do, heads, seg, (r0, b), scale)
        sv_delta = R.saved_delta(out, do, heads, (r0, b))
        worst["lse"] = max(worst["lse"], float((lse[heads, r0:b].double() - st_lse).abs().max()))
        for j in k_rows:
            assert j >= r0
            edk, edv = R.dkdv_row(q, k, v, do, j, g, seg, st_lse, st_delta, r0, scale)
            sdk, sdv = R.dkdv_row(q, k, v, do, j, g, seg, st_lse, sv_delta, r0, scale)
            note("dk", (j, g), _two_bounds(dk[j, g], edk, sdk, tmax["dk"]))
            note("dv", (j, g), _two_bounds(dv[j, g], edv, sdv, tmax["dv"]))
        del st_lse, st_delta, sv_delta
    print(f"{tag}: worst row excess [tight, stated] {worst}", flush=True)
    return worst


def _assert_rows(w, what):
    for name in ("dq", "dk", "dv"):
        assert w[name][0] <= 1.0 and w[name][1] <= 1.0, (what, name, w)
    assert w["lse"] <= LSE_TOL, (what, w)


def test_cfg3_single_segment_gradient_rows_against_float64():
    """cfg3-A: one causal segment of 131,072 tokens, 40 / 8 heads, d = 128, forward + backward through the autograd
    entry point.  Two KV head groups in full float64 statistics; dq rows and dk / dv rows at the first, a middle and the
    last key block, on block edges (127 | 128, 255 | 256) and inside the last block."""
    from long_vita_amd import ops
    S, Hq, Hkv, D = 131072, 40, 8, 128
    dev = torch.device("cuda")
    rnd = _rnd(dev, 31)
    q, k, v, do = rnd(S, Hq, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hq, D)
    scale = A.default_scale(D)
    out, lse = ops.varlen_attention(q, k, v, [0, S], scale, True)
    out.backward(do)
    torch.cuda.synchronize()
    rows = [0, 1, 127, 128, 255, 256, 4095, 65535, 65536, 100000, S - 129, S - 128, S - 2, S - 1]
    w = _check_rows("cfg3-A", q.detach(), k.detach(), v.detach(), do, out.detach(), lse, q.grad, k.grad, v.grad, (0, S),
                    (0, Hkv - 1), rows, rows, scale)
    _assert_rows(w, "cfg3-A")


def test_cfg3_packed_gradient_rows_against_float64():
    """cfg3-B: 131,072 tokens in log-uniform segments (the benchmark's packing, seed 1234).  The first, the longest, a
    short and the last segment: rows on both segment edges, around the segment's first 128-key block edge and inside its
    last partial block; the rows just outside a segment must not leak in (they belong to the neighbours' own checks)."""
    from long_vita_amd import ops
    S, Hq, Hkv, D = 131072, 40, 8, 128
    rs = np.random.RandomState(1234)
    lens = []
    while sum(lens) < S:
        lens.append(int(math.exp(rs.uniform(math.log(256), math.log(32768)))))
    cu = [0]
    for L in lens:
        cu.append(min(S, cu[-1] + L))
    cu[-1] = S
    dev = torch.device("cuda")
    rnd = _rnd(dev, 32)
    q, k, v, do = rnd(S, Hq, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hkv, D).requires_grad_(True), rnd(S, Hq, D)
    scale = A.default_scale(D)
    out, lse = ops.varlen_attention(q, k, v, cu, scale, True)
    out.backward(do)
    torch.cuda.synchronize()
    segs = list(zip(cu[:-1], cu[1:]))
    longest = max(range(len(segs)), key=lambda i: segs[i][1] - segs[i][0])
    shortest = min(range(len(segs)), key=lambda i: segs[i][1] - segs[i][0])
    for si in sorted({0, longest, shortest, len(segs) - 1}):
        a, b = segs[si]
        L = b - a
        rows = sorted({a, a + 1, a + min(127, L - 1), a + min(128, L - 1), a + L // 2, b - 1 - (L - 1) % 128, b - 2, b - 1})
        w = _check_rows(f"cfg3-B segment {si} [{a}, {b})", q.detach(), k.detach(), v.detach(), do, out.detach(), lse, q.grad,
                        k.grad, v.grad, (a, b), (3,), rows, rows, scale)
        _assert_rows(w, si)


def _excess_on_device(got, ref, rtol=REL_TOL, floor=0.25):
    """elementwise_excess of tests/test_gpu_attention.py, evaluated on the device (the tensors are gigabytes)."""
    ref = ref.float()
    bound = rtol * ref.abs() + rtol * floor * ref.abs().max()
    return float(((got.float() - ref).abs() / bound).max())


@pytest.mark.timeout(900)
def test_full_size_ring_backward_all_ranks_1m():
    """Backward of the 1,048,576-token, CP = 8 case with ALL eight ranks (threads of this process on one GPU, the
    transport an in-process mailbox that carries the real K / V blocks and the real fp32 running sums of dK / dV round
    the ring): every rank's dq, dk AND dv against its zig-zag rows of ONE whole-context backward, per element; and
    sampled rows of that whole-context backward against float64, so the chain ends at the oracle: dq rows anywhere,
    dk / dv rows of the last 32K keys with float64 row statistics, dk / dv rows of early keys (which sum over up to 1M
    rows x 5 heads) with the float64 algebra on the forward's log-sum-exp (itself checked on rows here and in
    test_gpu_attention.py) and delta from the saved bf16 output, as the kernels read it."""
    from long_vita_amd import ops
    from long_vita_amd.ring_context_parallel import RingAttention
    S, Hq, Hkv, D, cp = 1048576, 40, 8, 128, 8
    rep = Hq // Hkv
    dev = torch.device("cuda")
    rnd = _rnd(dev, 12)
    q, k, v, do = rnd(S, Hq, D), rnd(S, Hkv, D), rnd(S, Hkv, D), rnd(S, Hq, D)
    scale = A.default_scale(D)
    desc_q = ops.blocks_on_device([0, S], True, False, dev)
    desc_k = ops.blocks_on_device([0, S], True, True, dev)
    whole, whole_lse = ops.attn_fwd_launch(q, k, v, desc_q, scale, True)
    dq_all, dk_all, dv_all, _ = ops.attn_bwd_launch(q, k, v, whole, do, whole_lse, desc_q, desc_k, scale, True)
    torch.cuda.synchronize()

    # 1. the whole-context backward against float64 on sampled rows
    g = 5
    w = _check_rows("1M whole context, last 32K keys", q, k, v, do, whole, whole_lse, dq_all, dk_all, dv_all, (0, S), (g,),
                    (0, 255, 256, 65535, 65536, 524287, 524288, 900001, S - 1),
                    (S - 32768, S - 32768 + 127, S - 32768 + 128, S - 10000, S - 129, S - 128, S - 1), scale, stats_from=S - 32768)
    _assert_rows(w, "1M last keys")
    # early keys: every later row of the group's five heads contributes; float64 algebra on the forward's log-sum-exp
    heads = list(range(g * rep, g * rep + rep))
    kernel_lse = whole_lse[heads].double()
    sv_delta = R.saved_delta(whole, do, heads, (0, S))
    tmax_k, tmax_v = float(dk_all.float().abs().max()), float(dv_all.float().abs().max())
    worst = [0.0, 0.0]
    for j in (0, 127, 128, 65536, 524288 - 1, 524288):
        rdk, rdv = R.dkdv_row(q, k, v, do, j, g, (0, S), kernel_lse, sv_delta, 0, scale)
        worst = [max(worst[0], R.row_excess(dk_all[j, g], rdk, tiny=1e-4 * tmax_k)), max(worst[1], R.row_excess(dv_all[j, g], rdv, tiny=1e-4 * tmax_v))]
    del kernel_lse, sv_delta
    print(f"1M whole-context backward, early keys: worst row excess dk {worst[0]:.3f} dv {worst[1]:.3f}", flush=True)
    assert max(worst) <= 1.0, worst

    # 2. all eight ranks of the ring against it
    s_loc = S // cp
    fabric = LoopbackFabric(cp)

    def body(r):
        torch.cuda.set_device(0)
        eng = RingAttention(None, cp, r, causal=True, comm=LoopbackComm(fabric, r))
        ql, kl, vl, ol, dl = (zigzag_local(x, cp, r).reshape(s_loc, 1, -1) for x in (q, k, v, whole, do))
        lse_l = zigzag_local(whole_lse.t(), cp, r).t().contiguous()
        dq, dk, dv = eng.backward(ql, kl, vl, ol, lse_l, dl, Hq, scale)
        res = {}
        for name, got, ref, H in (("dq", dq, dq_all, Hq), ("dk", dk, dk_all, Hkv), ("dv", dv, dv_all, Hkv)):
            want = zigzag_local(ref, cp, r)
            got = got.reshape(s_loc, H, D)
            res[name] = (float((got.float() - want.float()).abs().max() / want.float().abs().max()), _excess_on_device(got, want))
        return res
    results = run_virtual_ranks(cp, body)
    torch.cuda.synchronize()
    for r, res in enumerate(results):
        print(f"rank {r}: (tensor-level rel err, elementwise excess) {res}", flush=True)
        for name, (rel, excess) in res.items():
            assert rel <= REL_TOL and excess <= 1.0, (r, name, rel, excess)
