"""Host-side cost of one torch.distributed P2P exchange as multi.Transport issues it (4 ops, fixed tensors),
measured with a single-rank RCCL group sending to itself: the GPU work is two local copies, so what is left
is what the host pays per step.  python tools/p2p_host_cost.py [bytes]"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
kw = dict(dtype=torch.uint8, device="cuda:0")
sl, sr, rl, rr = (torch.zeros(nbytes, **kw) for _ in range(4))
ops = [dist.P2POp(dist.isend, sr, 0), dist.P2POp(dist.irecv, rl, 0), dist.P2POp(dist.isend, sl, 0), dist.P2POp(dist.irecv, rr, 0)]
def once():
    for req in dist.batch_isend_irecv(ops):
        req.wait()
try:
    for _ in range(5): once()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K): once()
    host = (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / K
    print(f"exchange of 2 x {nbytes} B each way: host {host*1e6:.0f} us per step to issue, {total*1e6:.0f} us per step to complete")
except Exception as e:
    print("self P2P not supported here:", repr(e)[:300])
dist.destroy_process_group()
