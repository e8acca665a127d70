"""Ring attention with the HIP backend on one GPU: CP virtual ranks run as threads
of this process over a loopback transport (the box has a single MI355X; the RCCL
transport itself is exercised by bench.py --gpus N on a multi-GPU node)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention_ref as A
from tests.ring_helpers import LoopbackComm, LoopbackFabric, run_virtual_ranks, unzigzag, zigzag_local

REL_TOL = 2e-2
LSE_TOL = 1e-3


def rel(got, ref):
    g = got.detach().float().cpu().numpy().astype(np.float64)
    return float(np.abs(g - ref).max() / np.abs(ref).max())


@pytest.fixture(params=["default", "a4"])
def forward_kernel(request):
    """Ring blocks this small take the ping-pong forward kernel when the choice is the library's; "a4" forces the
    one-wave kernel that serves the long-context ring steps, so its rectangular and half-chunk blocks are checked here."""
    from long_vita_amd import _lib
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(request.param.encode()), "select")
    yield request.param
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")


@pytest.mark.parametrize("cp,S", [(2, 1024), (4, 2048), (8, 4096)])
def test_ring_hip_backend_matches_full_causal(cp, S, forward_kernel):
    from long_vita_amd.ring_context_parallel import RingAttention
    Hq, Hkv, D = 10, 2, 128
    g = torch.Generator().manual_seed(cp)
    q = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, [0, S], scale)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, [0, S], scale)
    fabric = LoopbackFabric(cp)
    dev = torch.device("cuda", 0)

    def body(r):
        torch.cuda.set_device(0)
        eng = RingAttention(None, cp, r, causal=True, comm=LoopbackComm(fabric, r))
        ql, kl, vl, dl = (zigzag_local(x, cp, r).reshape(S // cp, 1, -1).to(dev) for x in (q, k, v, do))
        out, lse = eng.forward(ql, kl, vl, Hq, scale, return_state=True)
        dq, dk, dv = eng.backward(ql, kl, vl, out, lse, dl, Hq, scale)
        torch.cuda.synchronize()
        return out[:, 0].cpu(), lse.cpu(), dq[:, 0].cpu(), dk[:, 0].cpu(), dv[:, 0].cpu()
    res = run_virtual_ranks(cp, body)
    out = unzigzag([r[0] for r in res], cp).reshape(S, Hq, D)
    assert rel(out, ref_o) <= REL_TOL
    lse = unzigzag([r[1].t() for r in res], cp).t()
    assert np.abs(lse.numpy() - ref_lse).max() <= LSE_TOL
    for idx, ref, H in ((2, rdq, Hq), (3, rdk, Hkv), (4, rdv, Hkv)):
        got = unzigzag([r[idx] for r in res], cp).reshape(S, H, D)
        assert rel(got, ref) <= REL_TOL, idx


def _mp_worker(rank, world, port, S, Hq, Hkv, D, scale, out_dir):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from long_vita_amd.ring_context_parallel import ringattn_context_parallel
    g = torch.Generator().manual_seed(17)
    q = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    dev = torch.device("cuda", 0)
    ql, kl, vl, dl = (zigzag_local(x, world, rank).reshape(S // world, 1, -1).to(dev) for x in (q, k, v, do))
    ql.requires_grad_(True); kl.requires_grad_(True); vl.requires_grad_(True)
    cp_para = dict(causal=True, cp_group=dist.group.WORLD, cp_size=world, rank=rank, cp_global_ranks=list(range(world)),
                   cp_group_for_send_recv_overlap=None, pse=None, pse_type=None)
    out = ringattn_context_parallel(ql, kl, vl, Hq, cp_para, scale, None, 0.)
    out.backward(dl)
    torch.cuda.synchronize()
    torch.save({"out": out.detach().cpu(), "dq": ql.grad.cpu(), "dk": kl.grad.cpu(), "dv": vl.grad.cpu()},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ring_two_processes_one_gpu_torch_distributed(tmp_path):
    """The shipped transport class (torch.distributed point-to-point posted on a side stream, completion handed to
    the compute stream as an event) and the HIP backend in two real processes.  Both ranks sit on the single GPU
    of the test box, so the process group is gloo (RCCL refuses two ranks on one device); on a multi-GPU node
    the same code runs with backend nccl (bench.py --gpus N)."""
    import os
    import torch.multiprocessing as mp
    S, Hq, Hkv, D, world = 2048, 10, 2, 128, 2
    scale = A.default_scale(D)
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_mp_worker, args=(world, port, S, Hq, Hkv, D, scale, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(17)
    q = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    ref_o, _ = A.varlen_attention(q, k, v, [0, S], scale)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, [0, S], scale)
    res = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=True) for r in range(world)]
    for name, ref, H in (("out", ref_o, Hq), ("dq", rdq, Hq), ("dk", rdk, Hkv), ("dv", rdv, Hkv)):
        got = unzigzag([r[name][:, 0] for r in res], world).reshape(S, H, D)
        assert rel(got, ref) <= REL_TOL, name


def test_transport_on_the_rccl_backend_one_rank():
    """The shipped transport class on the REAL RCCL backend (process group "nccl"): one rank posts the grouped
    isend + irecv of a K/V block to and from itself on the side stream while the block kernel runs on the compute
    stream, hands completion over as an event, and the received copy gives bit-identical attention; plus the
    collectives bench.py --gpus N calls (barrier, all_reduce MAX, all_gather_object) and the ring engine on that group.
    Two ranks on one device are refused by RCCL, so more than one rank runs over gloo in the tests above; this one is what
    can be said about the RCCL calls themselves on a one-GPU box.  Runs in a process of its own (tests/rccl_one_rank.py)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(36500 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_one_rank.py")
    res = subprocess.run([sys.executable, "-u", script], env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0 and "rccl self-exchange ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def _entry_worker(rank, world, port, S, Hq, Hkv, D, actual_seq_len, overlap=False):
    """One CP rank of the entry-point test: ambient state as Megatron / MindSpeed would hold it, the batch cut by
    get_batch_on_this_cp_rank, attention through flash_attention_forward, results reassembled by sync_output."""
    import os
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from long_vita_amd import cp_partition, state
    from long_vita_amd.core_attention import flash_attention_forward
    packed = actual_seq_len is not None
    # --use-cp-send-recv-overlap: a second communicator over the same ranks (dot_product_attention.py:59-60)
    over_group = dist.new_group(list(range(world))) if overlap else None
    state.mpu.initialize(dist.group.WORLD, world, rank, overlap_group=over_group)
    state.set_args(state.make_args(context_parallel_size=world, seq_length=S, reset_attention_mask=packed,
                                   reset_position_ids=packed, use_cp_send_recv_overlap=overlap))
    state.set_actual_seq_len(actual_seq_len)
    g = torch.Generator().manual_seed(23)
    q = torch.randn(S, 1, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, 1, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, 1, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(S, 1, Hq * D, generator=g).to(torch.bfloat16)
    # the batch is [b, s, ...] with the sequence on dim 1 (lcvlm_modellink/training/utils.py:351-362)
    batch = {"q": q.transpose(0, 1).cuda(), "k": k.transpose(0, 1).cuda(), "v": v.transpose(0, 1).cuda(),
             "do": do.transpose(0, 1).cuda(), "position_ids": torch.arange(S)[None].cuda()}
    batch = cp_partition.get_batch_on_this_cp_rank(batch)
    ql, kl, vl = (batch[x].transpose(0, 1).contiguous().requires_grad_(True) for x in "qkv")
    assert ql.shape == (S // world, 1, Hq, D)
    me = types.SimpleNamespace(hidden_size_per_attention_head=D, scale_mask_softmax=types.SimpleNamespace(scale=None),
                               softmax_scale=None, attention_dropout=types.SimpleNamespace(p=0.0), alibi=None,
                               pse=None, pse_type=None)
    out = flash_attention_forward(me, ql, kl, vl, None, None, None)
    assert out.shape == (S // world, 1, Hq * D)
    out.backward(batch["do"].transpose(0, 1))
    torch.cuda.synchronize()
    from long_vita_amd.ring_context_parallel import _engines
    (eng,) = _engines.values()
    assert eng.overlap_group is over_group and eng.comm.overlap_group is over_group
    # sync_output gathers [b, s_local, ...] over the CP group and undoes the zig-zag (tasks/.../utils.py:557-581)
    whole = {name: cp_partition.sync_output(t.detach().transpose(0, 1).contiguous().cpu())[0]
             for name, t in (("out", out), ("dq", ql.grad.flatten(2)), ("dk", kl.grad.flatten(2)), ("dv", vl.grad.flatten(2)))}
    if rank == 0:
        cu = [0] + list(actual_seq_len) if packed else [0, S]
        scale = A.default_scale(D)
        ref_o, _ = A.varlen_attention(q[:, 0], k[:, 0], v[:, 0], cu, scale)
        rdq, rdk, rdv = A.varlen_attention_grad(q[:, 0], k[:, 0], v[:, 0], do.reshape(S, Hq, D), cu, scale)
        for name, ref in (("out", ref_o), ("dq", rdq), ("dk", rdk), ("dv", rdv)):
            err = rel(whole[name].reshape(ref.shape), ref)
            assert err <= REL_TOL, (name, err)
    # the hybrid ring x Ulysses algorithm uses another subgroup and partition: it must refuse, not run as one ring
    state.set_args(state.make_args(context_parallel_size=world, seq_length=S, context_parallel_algo="hybrid_cp_algo"))
    try:
        flash_attention_forward(me, ql.detach(), kl.detach(), vl.detach(), None, None, None)
        raise AssertionError("hybrid_cp_algo ran")
    except NotImplementedError:
        pass
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("actual_seq_len,overlap", [(None, False), ([300, 301, 1500, 2048], False), (None, True)],
                         ids=["causal", "packed", "causal_send_recv_overlap_group"])
def test_cp_branches_through_flash_attention_forward_two_processes(actual_seq_len, overlap):
    """The context-parallel branches of the patched entry point (dot_product_attention.py:374-379 generic causal,
    :339-343 reset_attention_mask + actual_seq_len) end to end in two processes: state.mpu -> cp_para ->
    ringattn_context_parallel -> HIP ring, forward and backward, inputs cut by get_batch_on_this_cp_rank and results
    reassembled by sync_output's all_gather, against the oracle on the whole sequence."""
    import os
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000) + (0 if actual_seq_len is None else 1) + (2 if overlap else 0)
    mp.spawn(_entry_worker, args=(2, port, 2048, 10, 2, 128, actual_seq_len, overlap), nprocs=2, join=True)


class _FeedComm:
    """Stands in for the ring's transport for ONE rank: each exchange delivers, on a side stream, the K / V block the
    rank would receive at that step (the zig-zag shard of rank r - step - 1), taken from the whole K / V."""

    def __init__(self, rank, cp, k_all, v_all):
        self.rank, self.cp, self.k_all, self.v_all = rank, cp, k_all, v_all
        self.step = 0
        self.stream = torch.cuda.Stream()

    def exchange(self, send_bufs, recv_bufs, dst, src):
        self.stream.wait_stream(torch.cuda.current_stream())
        src_rank = (self.rank - self.step - 1) % self.cp
        self.step += 1
        with torch.cuda.stream(self.stream):
            recv_bufs[0][0].copy_(zigzag_local(self.k_all, self.cp, src_rank))
            recv_bufs[0][1].copy_(zigzag_local(self.v_all, self.cp, src_rank))
            done = torch.cuda.Event()
            done.record(self.stream)
        return done

    def wait(self, handle):
        torch.cuda.current_stream().wait_event(handle)


def test_full_size_ring_rank_equals_single_launch_1m():
    """The 1,048,576-token, CP = 8 case of the scaling runs: what two of the ranks compute over their eight ring steps
    (block kernels on the three kinds of rectangle + log-sum-exp merges) against the rows of ONE causal launch over
    the whole context."""
    from long_vita_amd import ops
    from long_vita_amd.ring_context_parallel import RingAttention
    S, Hq, Hkv, D, cp = 1048576, 40, 8, 128, 8
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(11)
    rnd = lambda *shape: (torch.randn(*shape, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v = rnd(S, Hq, D), rnd(S, Hkv, D), rnd(S, Hkv, D)
    scale = A.default_scale(D)
    whole, whole_lse = ops.varlen_attention(q, k, v, [0, S], scale, True)
    s_loc = S // cp
    for r in (0, 5):
        eng = RingAttention(None, cp, r, causal=True, comm=_FeedComm(r, cp, k, v))
        ql, kl, vl = (zigzag_local(x, cp, r).reshape(s_loc, 1, -1) for x in (q, k, v))
        out, lse = eng.forward(ql, kl, vl, Hq, scale, return_state=True)
        torch.cuda.synchronize()
        want = zigzag_local(whole, cp, r).float()
        err = (out.reshape(s_loc, Hq, D).float() - want).abs().max().item()
        assert err <= REL_TOL * want.abs().max().item(), (r, err)
        want_lse = zigzag_local(whole_lse.t(), cp, r).t()
        assert (lse - want_lse).abs().max().item() <= LSE_TOL, r
        del eng, out, lse, want


@pytest.mark.parametrize("cp,cu", [(2, [0, 300, 301, 1500, 2048]), (4, [0, 700, 2100, 2101, 3900, 4096]), (8, [0, 4096])])
def test_ring_hip_backend_packed_segments(cp, cu, forward_kernel):
    """Packed segments under the zig-zag split through the HIP kernels (several rectangles per launch, one launch
    per chunk pair), forward and backward, against the oracle's varlen attention on the whole sequence."""
    from long_vita_amd.ring_context_parallel import RingAttention
    S, Hq, Hkv, D = cu[-1], 10, 2, 128
    g = torch.Generator().manual_seed(100 + cp)
    q = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    scale = A.default_scale(D)
    ref_o, ref_lse = A.varlen_attention(q, k, v, cu, scale)
    rdq, rdk, rdv = A.varlen_attention_grad(q, k, v, do, cu, scale)
    fabric = LoopbackFabric(cp)
    dev = torch.device("cuda", 0)

    def body(r):
        torch.cuda.set_device(0)
        eng = RingAttention(None, cp, r, causal=True, comm=LoopbackComm(fabric, r))
        ql, kl, vl, dl = (zigzag_local(x, cp, r).reshape(S // cp, 1, -1).to(dev) for x in (q, k, v, do))
        out, lse = eng.forward(ql, kl, vl, Hq, scale, return_state=True, cu_seqlens=tuple(cu))
        dq, dk, dv = eng.backward(ql, kl, vl, out, lse, dl, Hq, scale, cu_seqlens=tuple(cu))
        torch.cuda.synchronize()
        return out[:, 0].cpu(), lse.cpu(), dq[:, 0].cpu(), dk[:, 0].cpu(), dv[:, 0].cpu()
    res = run_virtual_ranks(cp, body)
    out = unzigzag([r[0] for r in res], cp).reshape(S, Hq, D)
    assert rel(out, ref_o) <= REL_TOL
    lse = unzigzag([r[1].t() for r in res], cp).t()
    assert np.abs(lse.numpy() - ref_lse).max() <= LSE_TOL
    for idx, ref, H in ((2, rdq, Hq), (3, rdk, Hkv), (4, rdv, Hkv)):
        got = unzigzag([r[idx] for r in res], cp).reshape(S, H, D)
        assert rel(got, ref) <= REL_TOL, idx


@pytest.mark.parametrize("cp,cu", [(2, None), (4, None), (4, [0, 700, 2100, 2101, 3900, 4096])])
def test_ring_forward_with_rotary_embedding_folded_in(cp, cu, kernel_with_rotation):
    """Ring forward on UNROTATED q / k with the rotary tables of each rank's rows (keys rotated while they are packed
    into the ring message, queries inside the block kernels) equals, bit for bit, the ring forward on q / k rotated
    beforehand by the standalone kernel."""
    from long_vita_amd import ops
    from long_vita_amd.ring_context_parallel import RingAttention
    from oracle import rope_ref as R
    S = 4096 if cu else 1024 * cp
    Hq, Hkv, D = 10, 2, 128
    g = torch.Generator().manual_seed(7 + cp)
    q = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    ang = torch.from_numpy(R.angle_table(S, D, 1000000)).reshape(S, D)
    cos_all, sin_all = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    scale = A.default_scale(D)
    fabric = LoopbackFabric(cp)
    dev = torch.device("cuda", 0)
    kw = {"cu_seqlens": tuple(cu)} if cu else {}

    def body(r):
        torch.cuda.set_device(0)
        ql, kl, vl = (zigzag_local(x, cp, r).to(dev) for x in (q, k, v))
        cos_l, sin_l = zigzag_local(cos_all, cp, r).contiguous().to(dev), zigzag_local(sin_all, cp, r).contiguous().to(dev)
        s_loc = S // cp
        eng = RingAttention(None, cp, r, causal=True, comm=LoopbackComm(fabric, r))
        fused, lse_f = eng.forward(ql.reshape(s_loc, 1, -1), kl.reshape(s_loc, 1, -1), vl.reshape(s_loc, 1, -1), Hq, scale,
                                   return_state=True, rope=(cos_l, sin_l), **kw)
        fused, lse_f = fused.clone(), lse_f.clone()
        qr, kr = ops.rope_launch(ql, cos_l, sin_l), ops.rope_launch(kl, cos_l, sin_l)
        plain, lse_p = eng.forward(qr.reshape(s_loc, 1, -1), kr.reshape(s_loc, 1, -1), vl.reshape(s_loc, 1, -1), Hq, scale,
                                   return_state=True, **kw)
        torch.cuda.synchronize()
        return torch.equal(fused, plain) and torch.equal(lse_f, lse_p)
    assert all(run_virtual_ranks(cp, body))


def _rope_case(S, seed):
    from oracle import rope_ref as R
    Hq, Hkv, D = 10, 2, 128
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(S, Hkv, D, generator=g).to(torch.bfloat16)
    do = torch.randn(S, Hq, D, generator=g).to(torch.bfloat16)
    ang = torch.from_numpy(R.angle_table(S, D, 1000000)).reshape(S, D)
    return Hq, Hkv, D, q, k, v, do, torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)


@pytest.mark.parametrize("cp,cu", [(2, None), (4, [0, 700, 2100, 2101, 3900, 4096])])
def test_ring_rope_forward_backward_equals_rotating_first(cp, cu, kernel_with_rotation):
    """The ring with the rotary embedding folded in (ring_rope_forward / ring_rope_backward, what the autograd function
    of ringattn_context_parallel(rope=...) runs) against the plain ring on q / k rotated beforehand with the transposed
    rotation applied to its dq / dk afterwards: out, lse and dq / dk / dv bit for bit, over virtual ranks."""
    from long_vita_amd import ops
    from long_vita_amd.ring_context_parallel import RingAttention, ring_rope_backward, ring_rope_forward
    S = 4096 if cu else 1024 * cp
    Hq, Hkv, D, q, k, v, do, cos_all, sin_all = _rope_case(S, 17 + cp)
    scale = A.default_scale(D)
    fabric = LoopbackFabric(cp)
    dev = torch.device("cuda", 0)
    cus = None if cu is None else tuple(cu)

    def body(r):
        torch.cuda.set_device(0)
        s_loc = S // cp
        cos_l, sin_l = zigzag_local(cos_all, cp, r).contiguous().to(dev), zigzag_local(sin_all, cp, r).contiguous().to(dev)
        ql, kl, vl, dl = (zigzag_local(x, cp, r).reshape(s_loc, 1, -1).to(dev) for x in (q, k, v, do))
        eng = RingAttention(None, cp, r, causal=True, comm=LoopbackComm(fabric, r))
        out, lse, q_rot, k_rot = ring_rope_forward(eng, ql, kl, vl, Hq, scale, cos_l, sin_l, cus)
        dq, dk, dv = ring_rope_backward(eng, q_rot, k_rot, vl, out, lse, dl, Hq, scale, cos_l, sin_l, cus)
        fused = [t.clone() for t in (out, lse, dq, dk, dv)]
        qr = ops.rope_launch(ql.view(s_loc, Hq, D), cos_l, sin_l).view(s_loc, 1, -1)
        kr = ops.rope_launch(kl.view(s_loc, Hkv, D), cos_l, sin_l).view(s_loc, 1, -1)
        assert torch.equal(qr.view(s_loc, Hq, D), q_rot) and torch.equal(kr.view(s_loc, Hkv, D), k_rot)
        out2, lse2 = eng.forward(qr, kr, vl, Hq, scale, return_state=True, cu_seqlens=cus)
        dq2, dk2, dv2 = eng.backward(qr, kr, vl, out2, lse2, dl, Hq, scale, cu_seqlens=cus)
        dq2 = ops.rope_launch(dq2.view(s_loc, Hq, D), cos_l, sin_l, conj=True).view(s_loc, 1, -1)
        dk2 = ops.rope_launch(dk2.view(s_loc, Hkv, D), cos_l, sin_l, conj=True).view(s_loc, 1, -1)
        torch.cuda.synchronize()
        return all(torch.equal(a, b) for a, b in zip(fused, (out2, lse2, dq2, dk2, dv2)))
    assert all(run_virtual_ranks(cp, body))


def test_ring_rope_autograd_entry_point_single_rank(kernel_with_rotation):
    """ringattn_context_parallel(..., rope=(cos, sin)) under torch autograd (one rank: the autograd engine runs every
    backward of a device on one thread, so virtual ranks cannot exchange inside it) against apply-rope-first through the
    same entry point, packed segments included: out and the gradients of the unrotated q, k, v bit for bit."""
    from long_vita_amd import ops
    from long_vita_amd.ring_context_parallel import ringattn_context_parallel
    S = 1536
    Hq, Hkv, D, q, k, v, do, cos_all, sin_all = _rope_case(S, 31)
    dev = torch.device("cuda", 0)
    cos_l, sin_l = cos_all.to(dev), sin_all.to(dev)
    scale = A.default_scale(D)
    para = {"causal": True, "cp_group": None, "cp_size": 1, "rank": 0, "cp_global_ranks": [0]}
    for ends in (None, [300, 301, 1000, 1536]):
        res = []
        for fused in (True, False):
            ql, kl, vl = (x.to(dev).requires_grad_(True) for x in (q, k, v))
            if fused:
                out = ringattn_context_parallel(ql.view(S, 1, -1), kl.view(S, 1, -1), vl.view(S, 1, -1), Hq, para, scale,
                                                None, 0., ends, ends, rope=(cos_l, sin_l))
            else:
                qr, kr = ops.rope(ql, cos_l, sin_l), ops.rope(kl, cos_l, sin_l)
                out = ringattn_context_parallel(qr.view(S, 1, -1), kr.view(S, 1, -1), vl.view(S, 1, -1), Hq, para, scale,
                                                None, 0., ends, ends)
            out.backward(do.to(dev).view(S, 1, -1))
            torch.cuda.synchronize()
            res.append((out.detach().clone(), ql.grad.clone(), kl.grad.clone(), vl.grad.clone()))
        assert all(torch.equal(a, b) for a, b in zip(res[0], res[1])), ends
