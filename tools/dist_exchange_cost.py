#!/usr/bin/env python3
"""Cost of the pieces that close bench.py's timed region under torch.distributed (run with torchrun, any world size):
the statistics exchange (device reduction + one all-gather + device->host copy) and the barrier."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from tinympc_amd.distributed import allreduce_stats
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
stats = torch.arange(10, dtype=torch.float64, device=f"cuda:{lr}")
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def bar():
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
one = torch.zeros(1, device=f"cuda:{lr}")
res = {"allreduce_stats (all_gather + D2H)": t(lambda: allreduce_stats(stats, dist)),
       "sync + dist.barrier + sync": t(bar),
       "all_reduce of one element + sync": t(lambda: (dist.all_reduce(one), torch.cuda.synchronize())),
       "stats.to(cpu)": t(lambda: stats.to("cpu"))}
if dist.get_rank() == 0:
    for k, v in res.items(): print(f"{k:40s} {v:9.1f} us")
dist.destroy_process_group()
