"""One-off wider fuzz of the attention kernels against the float64 oracle (more seeds and every kernel variant than
tests/test_gpu_fuzz.py runs): python tools_diag/fuzz_more.py [first_seed] [n_seeds]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import attention_ref as A
from long_vita_amd import ops, _lib
from tests.test_gpu_fuzz import _case, _rel

first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = _lib.lib()
variants = [(b"default", b"default", b"default"), (b"pp", b"pp", b"rs1"), (b"lockstep", b"lockstep", b"single"), (b"pp16", b"default", b"rs"),
            (b"a4", b"default", b"g")]
bad = 0
t0 = time.time()
for seed in range(first, first + n):
    D, Hkv, Hq, causal, cu = _case(seed)
    T = cu[-1]
    g = torch.Generator().manual_seed(5000 + seed)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale, causal)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale, causal)
    fwdk, dqk, dkdvk = variants[seed % len(variants)] if seed % 3 == 0 else variants[0]
    _lib.check(L.lvattn_select_fwd_kernel(fwdk), "fwd")
    _lib.check(L.lvattn_select_bwd_kernels(dqk, dkdvk), "bwd")
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, lse = ops.varlen_attention(qc, kc, vc, cu, scale, causal)
    out.backward(do.cuda())
    torch.cuda.synchronize()
    errs = (_rel(out, ref_o), float(np.abs(lse.detach().cpu().numpy().astype(np.float64) - ref_lse).max()),
            _rel(qc.grad, rdq), _rel(kc.grad, rdk), _rel(vc.grad, rdv))
    ok = errs[0] <= 2e-2 and errs[1] <= 1e-3 and max(errs[2:]) <= 2e-2
    if not ok:
        bad += 1
        print("FAIL", seed, fwdk, dqk, dkdvk, f"D={D} Hq={Hq} Hkv={Hkv} causal={causal} cu={cu}", errs, flush=True)
    if (seed - first) % 25 == 0:
        print(f"seed {seed} ok so far, {bad} failures, {time.time() - t0:.0f}s", flush=True)
_lib.check(L.lvattn_select_fwd_kernel(b"default"), "fwd")
_lib.check(L.lvattn_select_bwd_kernels(b"default", b"default"), "bwd")
print("done", n, "cases,", bad, "failures")
sys.exit(1 if bad else 0)
