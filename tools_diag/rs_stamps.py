"""Diagnostic: phase timings of the role-split dK/dV kernel from a stamped build (long_vita_amd/lv_stamp.so)."""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from long_vita_amd import ops, _lib
L = _lib.lib()
L.lv_debug_rs_stamps.restype = ctypes.c_int
L.lv_debug_rs_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
S, HQ, HKV, D = 131072, 40, 8, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
q, k, v, do = rnd(S, HQ, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HKV, D).requires_grad_(True), rnd(S, HQ, D)
for units in (2, 1):
    _lib.check(L.lvattn_select_bwd_kernels(b"default", b"rs" if units == 2 else b"rs1"), "sel")
    for rep in range(2):
        q.grad = k.grad = v.grad = None
        out, _ = ops.varlen_attention(q, k, v, [0, S], 1 / math.sqrt(D), True)
        out.backward(do)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        L.lv_debug_rs_stamps(buf, 1)
    for role, name in ((0, "V-wave"), (1, "K-wave")):
        vec, b1, mat, b2, iters, waves = [buf[role * 8 + i] for i in range(6)]
        print(f"units/iter {units} {name}: per iteration  vector work {vec / iters:.0f}  wait at its barrier {b1 / iters:.0f}  "
              f"matrix work {mat / iters:.0f}  wait at its barrier {b2 / iters:.0f}  total {(vec + b1 + mat + b2) / iters:.0f} ticks"
              f"  ({iters} iterations, {waves} waves; {16 * units} MFMAs = {512 * units} cycles per matrix phase)")
