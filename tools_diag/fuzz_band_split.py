"""One-off wider fuzz of this round's table forms against the float64 oracle: banded causal attention (random band
widths, packed ragged segments) and split sweeps (random forced piece lengths for the forward / dQ and the dK/dV launch,
workspaces pre-filled with NaN), forward and backward:  python tools_diag/fuzz_band_split.py [first_seed] [n_seeds]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import attention_ref as A
from long_vita_amd import ops
from tests.test_gpu_fuzz import _case, _rel

first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()
ops._poison_workspaces = True
for seed in range(first, first + n):
    D, Hkv, Hq, causal, cu = _case(seed)
    rs = np.random.RandomState(7000 + seed)
    mode = "band" if (causal and rs.rand() < 0.5) else "split"
    T = cu[-1]
    g = torch.Generator().manual_seed(9000 + seed)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    band = None
    if mode == "band":
        band = int(rs.choice([0, 1, 5, 31, 32, 33, 63, 64, 100, 255, 256, 257, 500]))
        ops.set_split_override(None, None)
        what = f"band={band}"
    else:
        pf = int(rs.choice([0, 256, 320, 512, 1024]))
        pb = int(rs.choice([0, 32, 96, 128, 256, 512]))
        ops.set_split_override(pf, pb)
        what = f"pieces=({pf},{pb})"
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal, band)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, causal, band)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, lse = ops.varlen_attention(qc, kc, vc, cu, scale, causal, band)
    out.backward(do.cuda())
    torch.cuda.synchronize()
    fin = np.isfinite(ref_lse)
    lse_err = float(np.abs(lse.detach().cpu().numpy().astype(np.float64)[fin] - ref_lse[fin]).max()) if fin.any() else 0.0
    errs = (_rel(out, ref_o), lse_err, _rel(qc.grad, rdq), _rel(kc.grad, rdk), _rel(vc.grad, rdv))
    ok = errs[0] <= 2e-2 and errs[1] <= 1e-3 and max(errs[2:]) <= 2e-2 and all(np.isfinite(e) for e in errs)
    if not ok:
        bad += 1
        print("FAIL", seed, what, f"D={D} Hq={Hq} Hkv={Hkv} causal={causal} cu={cu}", errs, flush=True)
    if (seed - first) % 25 == 0:
        print(f"seed {seed} ({what}) ok so far, {bad} failures, {time.time() - t0:.0f}s", flush=True)
ops.set_split_override(None, None)
ops._poison_workspaces = False
print("done", n, "cases,", bad, "failures")
sys.exit(1 if bad else 0)
