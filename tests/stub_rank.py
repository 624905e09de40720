"""A stand-in for bench.py's GPU worker, for driving bench.launch_ranks on a box without GPUs (tests/test_dist_gloo.py):
the same rendezvous (RANK / WORLD_SIZE / MASTER_* from the launcher, gloo instead of RCCL), bench.py's own shard plan,
counter all-reduce and max-over-ranks timing, and rank 0 prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--envs", type=int, default=0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--fail-rank", type=int, default=-1)
a = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == a.gpus and int(os.environ["LOCAL_RANK"]) == rank
if rank == a.fail_rank:
    sys.exit(3)
dist.init_process_group("gloo", rank=rank, world_size=world)
envs = bench.parse_args(["--gpus", str(a.gpus)] + (["--envs", str(a.envs)] if a.envs else [])).envs  # bench.py's own default
plan = bench.shard_plan(rank, world, envs)
counters = torch.zeros(4, dtype=torch.int64)
log = []


def run_steps(k):
    log.append("steps")
    counters[0] += plan["n_envs"] * k
    counters[1] += 1


def reduce_in_region():
    log.append("allreduce")
    bench.reduce_counters(counters, dist)


def barrier():
    log.append("barrier")
    dist.barrier()


bench.timed_region(run_steps, a.steps, barrier, reduce_in_region if world > 1 else None)  # bench.py's own region
slowest = bench.reduce_max(1.0 + rank, torch.device("cpu"), dist)
if rank == 0:
    print(json.dumps({"n_gpus": world, "steps_total": int(counters[0]), "first_env": plan["first_env"], "global_envs": plan["global_envs"],
                      "envs_per_gpu": envs, "slowest": slowest, "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                      "region": log}), flush=True)
dist.barrier()
dist.destroy_process_group()
