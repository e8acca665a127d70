"""One rank on the real RCCL backend: the transport class's grouped send / receive (here to and from itself), posted on
its side stream while a block kernel runs on the compute stream, and the collectives bench.py --gpus N calls."""
import os
import sys
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from long_vita_amd.ring_context_parallel import TorchDistComm, RingAttention  # noqa: E402
from long_vita_amd import ops  # noqa: E402

comm = TorchDistComm(dist.group.WORLD, dev)
second = dist.new_group([0])        # a second "nccl" communicator, as --use-cp-send-recv-overlap sets one up
g = torch.Generator(device=dev).manual_seed(3)
S, HQ, HKV, D = 8192, 10, 2, 128
q = torch.randn(S, HQ, D, generator=g, device=dev).to(torch.bfloat16)
kv = [torch.randn(2, S, HKV, D, generator=g, device=dev).to(torch.bfloat16), torch.zeros(2, S, HKV, D, device=dev, dtype=torch.bfloat16)]
desc = ops.blocks_on_device([0, S], True, False, dev)
for rep in range(3):
    handle = comm.exchange([kv[0]], [kv[1]], 0, 0)          # isend + irecv in one group, on the side stream
    out, lse = ops.attn_fwd_launch(q, kv[0][0], kv[0][1], desc, D ** -0.5, True)   # the block kernel, meanwhile
    comm.wait(handle)
    out2, lse2 = ops.attn_fwd_launch(q, kv[1][0], kv[1][1], desc, D ** -0.5, True)  # on the received copy
    torch.cuda.synchronize()
    assert torch.equal(kv[0], kv[1]), "received block differs"
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    kv[1].zero_()
# a grouped send + receive on ONE communicator from a rank to itself matches; split over two communicators the send on
# the first would wait for a receive on the first that nobody posts.  What one rank can exercise of the split form is
# each communicator and each lane on its own: K / V on lane 0 through the first group while an fp32 block goes through
# the second group on lane 1, both under a block kernel.
f32 = [torch.randn(2, S, HKV, D, generator=g, device=dev), torch.zeros(2, S, HKV, D, device=dev)]
lane1 = TorchDistComm(second, dev)
h0 = comm.exchange([kv[0]], [kv[1]], 0, 0, lane=0)
h1 = lane1.exchange([f32[0]], [f32[1]], 0, 0, lane=1)
out, lse = ops.attn_fwd_launch(q, kv[0][0], kv[0][1], desc, D ** -0.5, True)
comm.wait(h0)
lane1.wait(h1)
torch.cuda.synchronize()
assert torch.equal(kv[0], kv[1]) and torch.equal(f32[0], f32[1]), "two communicators, two lanes: received blocks differ"
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
objs = [None]
dist.all_gather_object(objs, {"rank": 0})
assert float(t.item()) == 1.5 and objs[0]["rank"] == 0
eng = RingAttention(dist.group.WORLD, 1, 0, causal=True)
o = eng.forward(q.view(S, 1, HQ * D), kv[0][0].reshape(S, 1, HKV * D), kv[0][1].reshape(S, 1, HKV * D), HQ)
torch.cuda.synchronize()
assert torch.equal(o.view(S, HQ, D), out)
dist.destroy_process_group()
print("rccl self-exchange ok")
