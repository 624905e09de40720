#!/usr/bin/env python3
"""what the pieces of bench.py's closing sequence cost under RCCL with a one-rank communicator (host time, microseconds)"""
import os, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
c = torch.zeros(4, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main)
for _ in range(5):
    dist.all_reduce(c); dist.barrier(); torch.cuda.synchronize()
def t(f, n=50):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e6
def ar():
    side.wait_stream(main)
    with torch.cuda.stream(side): dist.all_reduce(c)
    torch.cuda.synchronize()
print("all_reduce on a side stream + device sync: %.1f us" % t(ar))
print("dist.barrier(): %.1f us" % t(lambda: dist.barrier()))
print("dist.barrier(device_ids=[0]): %.1f us" % t(lambda: dist.barrier(device_ids=[0])))
print("barrier + synchronize: %.1f us" % t(lambda: (dist.barrier(), torch.cuda.synchronize())))
f = torch.zeros(1, device=dev)
print("all_reduce(float) + sync (a hand-made barrier): %.1f us" % t(lambda: (dist.all_reduce(f), torch.cuda.synchronize())))
print("torch.cuda.synchronize() alone: %.1f us" % t(lambda: torch.cuda.synchronize()))
dist.destroy_process_group()
