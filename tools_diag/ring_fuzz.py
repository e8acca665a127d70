"""One-off fuzz of the ring engine over virtual ranks on one GPU (HIP kernels, loopback transport): seeded random
context-parallel sizes, chunk lengths, head ratios and packed-segment boundaries (or none), forward + backward
against the float64 oracle on the whole sequence.  python tools_diag/ring_fuzz.py [first_seed] [n_seeds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import attention_ref as A  # noqa: E402
from long_vita_amd.ring_context_parallel import RingAttention  # noqa: E402
from tests.ring_helpers import LoopbackComm, LoopbackFabric, run_virtual_ranks, unzigzag, zigzag_local  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
dev = torch.device("cuda", 0)


def rel(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


bad, t0 = 0, time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    cp = int(rng.choice([2, 4, 8]))
    c = int(rng.choice([32, 64, 96, 130, 256, 300]))
    S = 2 * cp * c
    Hq, Hkv = [(4, 2), (10, 2), (5, 1), (8, 8)][int(rng.integers(0, 4))]
    D = int(rng.choice([128, 128, 64]))
    packed = rng.random() < 0.7
    cu = None
    if packed:
        n_cut = int(rng.integers(0, 7))
        cuts = sorted(set(int(x) for x in rng.choice(np.arange(1, S), size=n_cut, replace=False))) if n_cut else []
        if rng.random() < 0.4:
            cuts = sorted(set(cuts + [int(rng.integers(1, 2 * cp)) * c]))
        cu = [0] + cuts + [S]
    g = torch.Generator().manual_seed(1000 + seed)
    q = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    ocu = cu if cu else [0, S]
    ref_o, ref_lse = A.varlen_attention(q, k, v, ocu, scale)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, ocu, scale)
    fabric = LoopbackFabric(cp)
    kw = {"cu_seqlens": tuple(cu)} if cu else {}

    def body(r):
        torch.cuda.set_device(0)
        eng = RingAttention(None, cp, r, causal=True, comm=LoopbackComm(fabric, r))
        ql, kl, vl, dl = (zigzag_local(x, cp, r).reshape(S // cp, 1, -1).to(dev) for x in (q, k, v, do))
        out, lse = eng.forward(ql, kl, vl, Hq, scale, return_state=True, **kw)
        dq, dk, dv = eng.backward(ql, kl, vl, out, lse, dl, Hq, scale, **kw)
        torch.cuda.synchronize()
        return out[:, 0].cpu(), lse.cpu(), dq[:, 0].cpu(), dk[:, 0].cpu(), dv[:, 0].cpu()
    res = run_virtual_ranks(cp, body)
    out = unzigzag([r[0] for r in res], cp).reshape(S, Hq, D)
    lse = unzigzag([r[1].t() for r in res], cp).t()
    errs = [rel(out, ref_o), float(np.abs(lse.numpy() - ref_lse).max())]
    for idx, ref, H in ((2, rdq, Hq), (3, rdk, Hkv), (4, rdv, Hkv)):
        errs.append(rel(unzigzag([r[idx] for r in res], cp).reshape(S, H, D), ref))
    ok = errs[0] <= 2e-2 and errs[1] <= 1e-3 and max(errs[2:]) <= 2e-2
    if not ok:
        bad += 1
        print("FAIL", seed, f"cp={cp} c={c} Hq={Hq} Hkv={Hkv} D={D} cu={cu}", errs, flush=True)
    if (seed - first) % 10 == 9:
        print(f"seed {seed}: {bad} failures so far, {time.time() - t0:.0f}s", flush=True)
print(f"done {n} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
