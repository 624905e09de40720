#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Pommerman tick on MI355X (BASELINE.json's metric).

One "step" = one pass of the hot path over the whole batch: one launch of pom_step_kernel that
advances every env of this rank by one tick with state round-tripping HBM (ticks_per_launch = 1).
Workload (config.workload): 65,536 concurrent 11x11 FFA envs per GPU, start boards with the
reference's cell distribution, Move[4] i.i.d. uniform over {IDLE,UP,DOWN,LEFT,RIGHT,BOMB}
(RandomAgent distribution) from the counter-based stream of include/pom_rng.h, finished envs
restart from their snapshot, 800-tick episode cap.  Inputs are resident in HBM before the timed
region starts; nothing crosses PCIe inside it.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), envs sharded
contiguously with no data-path collective; the only collective is the all-reduce of the
step counters after the last tick (weak scaling: per-GPU batch fixed).

The JSON line also carries
  roofline     — algorithmic HBM bytes per launch (2024 B x envs) / mean launch time
                 measured with HIP events on the launch stream, vs the 8 TB/s HBM3E peak;
  cpu_baseline — the oracle (CPU restatement, bit-exact to the reference) timed on this host's
                 cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 2024  # 1004 B State read + 1004 B State write + 16 B Move[4]  
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def shard_plan(rank: int, world: int, envs_per_gpu: int) -> dict:
    """Contiguous env ranges; env_offset keys the synthetic move stream so shards differ."""
    return {"first_env": rank * envs_per_gpu, "n_envs": envs_per_gpu, "global_envs": world * envs_per_gpu}


def _all_reduce(t, op, dist_mod):
    """all_reduce in place; a device tensor goes through the host when the process group is gloo (CPU tests, and the
    POM_BENCH_REHEARSAL mode that runs several ranks on ONE GPU, which RCCL refuses)"""
    if t.is_cuda and dist_mod.get_backend() == "gloo":
        h = t.cpu()
        dist_mod.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist_mod.all_reduce(t, op=op)
    return t


def reduce_counters(counters, dist_mod=None):
    """Sum the per-rank int64 counters over ranks (RCCL all-reduce on GPU, gloo in CPU tests)."""
    if dist_mod is not None and dist_mod.is_initialized() and dist_mod.get_world_size() > 1:
        _all_reduce(counters, dist_mod.ReduceOp.SUM, dist_mod)
    return counters


def reduce_max(value: float, device, dist_mod=None) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist_mod is not None and dist_mod.is_initialized() and dist_mod.get_world_size() > 1:
        _all_reduce(t, dist_mod.ReduceOp.MAX, dist_mod)
    return float(t.item())


def cpu_baseline(start: np.ndarray, seed: int, dist_id: int, max_steps: int, budget_s: float = 10.0) -> dict:
    """Time the oracle on the host cores on a bounded sample of the same workload."""
    import ctypes as C
    from tests.oracle_lib import ORACLE_DIR, Oracle

    lib_path = os.path.join(ORACLE_DIR, "libpom_oracle_native.so")
    flags = "-O3 -march=native"
    try:  # tuned for THIS host; the generic build is the fallback
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "libpom_oracle_native.so"], check=True, capture_output=True, timeout=120)
        lib = C.CDLL(lib_path)
    except Exception:
        Oracle()
        lib = C.CDLL(os.path.join(ORACLE_DIR, "libpom_oracle.so"))
        flags = "-O3"
    lib.pom_oracle_run_random.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pom_oracle_run_random.restype = C.c_int64
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    per_thread = 2048
    cores = max(1, min(cores, start.size // per_thread))
    chunk_ticks = 50
    totals = [0] * cores
    deadline = time.perf_counter() + budget_s

    def work(k: int) -> None:
        init = np.ascontiguousarray(start[k * per_thread:(k + 1) * per_thread])
        cur = init.copy()
        tick = 0
        while time.perf_counter() < deadline:
            totals[k] += lib.pom_oracle_run_random(cur.ctypes.data, init.ctypes.data, per_thread, chunk_ticks, seed,
                                                   k * per_thread, tick, dist_id, max_steps)
            tick += chunk_ticks

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    wall = time.perf_counter() - t0
    steps = int(sum(totals))
    port = {
        "value": steps / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"oracle/pom_oracle.c ({flags}), {cores} threads x {per_thread} envs of the same boards and move stream, "
                  f"{steps} env-steps in {wall:.1f} s",
    }
    # The reference itself, where its prebuilt library travelled with the repo (oracle/_ref/, built in the container that holds
    # /root/reference): the unmodified bboard::Step, -O3, timed in a child process (it has UB on reachable states; a guard skips
    # those ticks, and a crash only costs this leg).
    ref_lib = os.path.join(ORACLE_DIR, "_ref", "libpomref_bench.so")
    if os.path.exists(ref_lib):
        import tempfile
        try:
            with tempfile.TemporaryDirectory() as td:
                f = os.path.join(td, "boards.npy")
                np.save(f, start[:cores * per_thread])
                out = subprocess.run([sys.executable, os.path.join(ORACLE_DIR, "ref_baseline_run.py"), f, str(seed), str(dist_id),
                                      str(max_steps), str(budget_s)], capture_output=True, text=True, timeout=budget_s * 4 + 60)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            return {
                "value": r["value"], "unit": "env-steps/s", "cores": r["cores"], "kind": "reference",
                "sample": f"unmodified reference bboard::Step (oracle/_ref/libpomref_bench.so, g++ -O3), {r['cores']} threads x "
                          f"{r['per_thread']} envs of the same boards and move stream, {r['steps']} env-steps in "
                          f"{r['timed_s_per_thread']:.1f} s of reference time per thread ({r['skipped_ub_ticks']} ticks with reference UB "
                          f"stepped by the restatement, untimed)",
                "port": port,
            }
        except Exception as exc:  # the figure below is still a measured baseline
            port["reference_leg"] = f"failed: {type(exc).__name__}"
    return port


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--kind", default="ffa", choices=["ffa", "stress"])
    ap.add_argument("--dist", default="random", choices=["harmless", "random", "stress"])
    ap.add_argument("--ticks-per-launch", type=int, default=1)
    ap.add_argument("--max-steps", type=int, default=800)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", default="random", choices=["random", "simple"],
                    help="random: Move[4] from the counter stream (--dist); simple: the device SimpleAgent policy (config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the SimpleAgent segment (profiling runs: one kernel shape only)")
    ap.add_argument("--streams", type=int, default=0, help="sub-batches per step (0 = library default)")
    ap.add_argument("--fresh-boards", action="store_true",
                    help="boards drawn on the device (pom_batch_generate) and a new one per episode instead of the snapshot replay "
                         "BASELINE's configs prescribe (SURVEY §8 f3)")
    ap.add_argument("--envs-per-wave", type=int, default=0)
    ap.add_argument("--lanes-per-env", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    entry.build_hip()
    import pomcpp_amd as pa
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, CNT_STEPS

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the stepper has no CPU path")
    # POM_BENCH_REHEARSAL=1: the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices, gloo instead of
    # RCCL, which refuses two ranks on one device).  For rehearsing the launch / barrier / reduction logic, never for numbers.
    rehearsal = os.environ.get("POM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        # create RCCL's communicator and streams now: they take hardware queues, which the stream tuning below must see
        _all_reduce(torch.zeros(1, device=device), dist.ReduceOp.SUM, dist)
        torch.cuda.synchronize()
    dist_id = {"harmless": 0, "random": 1, "stress": 2}[args.dist]
    plan = shard_plan(rank, world, args.envs)

    if args.fresh_boards and args.kind != "ffa":
        raise SystemExit("--fresh-boards draws ffa boards")
    start = None if args.fresh_boards else pa.make_boards(plan["n_envs"], seed=args.seed * 1000003 + rank, kind=args.kind)
    # torch's default stream is handle 0, which the C-ABI reads as "create your own": run everything on an
    # explicit torch stream so the HIP events below bracket exactly the stream the kernels are launched on
    stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    env = BatchEnvironment(plan["n_envs"], device=local_rank, mode=MODE_ENV, auto_reset=True, max_steps=args.max_steps,
                           env_offset=plan["first_env"], stream=stream.cuda_stream, streams=args.streams,
                           envs_per_wave=args.envs_per_wave, lanes_per_env=args.lanes_per_env,
                           fresh_boards=args.fresh_boards, board_seed=args.seed)
    if args.fresh_boards:
        env.generate(args.seed)
        start = env.get_state()  # for the CPU baseline leg and config 3
    else:
        env.make_game(start)
    counters = torch.zeros(4, dtype=torch.int64, device=device)

    def barrier() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tpl = args.ticks_per_launch
    if args.policy == "simple":
        if tpl != 1:
            raise SystemExit("--policy simple runs one tick per launch")

        def one_step():
            env.step_simple(args.seed, 1)
    else:
        def one_step():
            env.step_random(args.seed, dist_id, ticks=tpl, ticks_per_launch=tpl)
    tuned = None
    if args.streams == 0:
        # Untimed: how many sub-batches per step?  More parts overlap more load/store with compute, but ROCm maps all streams
        # of the process onto 4 hardware queues and parts that share a queue serialize (profiles/r01_streams.txt) — how many
        # are free depends on the process (torch, RCCL), so measure instead of guessing.  Results do not depend on the choice.
        tuned = {}
        for k in (3, 2, 1):  # first touch of a sub-stream creates its hardware queue (~10 ms once): keep that out of the timings
            env.set_streams(k)
            for _ in range(5):
                one_step()
            env.sync()
        for k in (2, 3, 1):
            env.set_streams(k)
            best_k = None
            for _ in range(2):  # the better of two short runs: one stall must not decide the shape
                for _ in range(10):
                    one_step()
                env.sync()
                t_a = time.perf_counter()
                for _ in range(40):
                    one_step()
                env.sync()
                dt = (time.perf_counter() - t_a) / 40 * 1e3
                best_k = dt if best_k is None else min(best_k, dt)
            tuned[k] = best_k
        best = min(tuned, key=tuned.get)
        if world > 1:  # every rank must run the same shape: take the vote of the slowest rank's best
            votes = torch.tensor([tuned[1], tuned[2], tuned[3]], dtype=torch.float64, device=device)
            _all_reduce(votes, dist.ReduceOp.MAX, dist)
            best = int(torch.argmin(votes).item()) + 1
        env.set_streams(best)
    for _ in range(args.warmup):
        one_step()
    env.counters_into(counters.data_ptr())
    reduce_counters(counters, dist)  # warm the RCCL communicator outside the timed region
    env.reset_counters()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        one_step()
    env.flush()  # steps run as sub-batches on internal streams: order them before the event on this stream
    ev1.record(stream)
    env.counters_into(counters.data_ptr())
    reduce_counters(counters, dist)  # the one collective: step / episode totals over all ranks
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_max(elapsed, device, dist)
    step_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream: mean time of one step (all its launches)
    step_ms = reduce_max(step_ms, device, dist)
    # a step is issued as `parts` launches of pom_step_kernel over contiguous sub-batches on parallel streams; time the
    # individual launches too (HIP events on their own streams), outside the timed region
    epw, lpe, parts = env.launch_shape()
    env.profile(True)
    for _ in range(max(1, 256 // parts)):
        env.step_random(args.seed, dist_id, ticks=tpl, ticks_per_launch=tpl)  # the tick kernel alone
    launch_ms, n_launch = env.profile_read()
    env.profile(False)

    # BASELINE config 3 beside the headline (rank 0 of a single-GPU run, untimed region): the same boards played from the start
    # by the device SimpleAgent policy, act x4 + Step per env-step as Environment::Step does.  200 untimed ticks first: games
    # last ~190 ticks under this policy, so the batch is then a steady mix of openings, mid-games and restarts.
    config3 = None
    if world == 1 and args.policy == "random" and tpl == 1 and not args.no_config3:
        env.make_game(start)
        env.set_tick(0)
        env.step_simple(args.seed, 200)
        env.sync()
        ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_c = 200
        ev2.record(stream)
        env.step_simple(args.seed, n_c)
        env.flush()
        ev3.record(stream)
        env.sync()
        ms_c = ev2.elapsed_time(ev3) / n_c
        config3 = {"workload": f"{args.envs} envs, 4x SimpleAgent policy on the device + Step, games from the start, "
                               "200 warm-up ticks", "value": plan["n_envs"] / (ms_c * 1e-3),
                   "unit": "env-steps/s", "ms_per_step": ms_c, "steps": n_c}
    # BASELINE's other single-GPU configs, briefly (untimed region, default run only): config 2 (4,096 envs, random moves) and
    # config 5 (65,536 envs, kick / chain-explosion stress boards and move mix) — parity-test cases first, numbers for context
    other = None
    default_run = (args.envs == 65536 and args.kind == "ffa" and args.dist == "random" and not args.fresh_boards)
    if world == 1 and args.policy == "random" and tpl == 1 and not args.no_config3 and default_run:
        other = {}
        for name, n_o, kind_o, dist_o in (("config2_4096_envs_random", 4096, "ffa", 1), ("config5_65536_envs_stress", 65536, "stress", 2)):
            e2 = BatchEnvironment(n_o, device=local_rank, mode=MODE_ENV, auto_reset=True, max_steps=args.max_steps,
                                  stream=stream.cuda_stream)
            e2.make_game(pa.make_boards(n_o, seed=args.seed * 1000003, kind=kind_o))
            e2.step_random(args.seed, dist_o, ticks=60)
            e2.sync()
            ev4, ev5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_o_steps = 200
            ev4.record(stream)
            e2.step_random(args.seed, dist_o, ticks=n_o_steps)
            e2.flush()
            ev5.record(stream)
            e2.sync()
            ms_o = ev4.elapsed_time(ev5) / n_o_steps
            other[name] = {"value": n_o / (ms_o * 1e-3), "unit": "env-steps/s", "ms_per_step": ms_o, "steps": n_o_steps}
            e2.close()
    total_steps = int(counters[CNT_STEPS].item())
    expect = plan["global_envs"] * args.steps * tpl
    if total_steps != expect:
        raise SystemExit(f"step counter {total_steps} != envs x ticks {expect}")

    if rank == 0:
        algo_bytes = ALGO_BYTES_PER_STEP * plan["n_envs"] * tpl
        achieved = algo_bytes / (step_ms * 1e-3) / 1e9
        traffic = None  # PMC-derived HBM bytes per launch come from the committed rocprofv3 passes, for this workload only
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tj) and args.envs == 65536 and tpl == 1 and args.kind == "ffa" and args.dist == "random":
            traffic = json.load(open(tj))["hbm_bytes_per_step"]  # all launches of one step
        line = {
            "metric": "env_steps_per_sec", "value": total_steps / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, not a measurement)" if rehearsal else ""),
            "config": {
                "workload": f"{args.envs} concurrent 11x11 FFA envs per GPU, {args.kind} boards, "
                            + (f"uniform-{args.dist} Move[4] (RandomAgent distribution)" if args.policy == "random"
                               else "4x SimpleAgent policy on the device (act x4 + Step per env-step, as Environment::Step)")
                            + (", auto-reset onto a fresh device-generated board" if args.fresh_boards else ", auto-reset")
                            + f", {args.max_steps}-tick cap",
                "policy": args.policy,
                "envs_per_gpu": args.envs, "global_envs": plan["global_envs"], "ticks_per_launch": tpl,
                "envs_per_wave": epw, "lanes_per_env": lpe, "launches_per_step": parts,
                "launches_per_step_tuning_ms": tuned,
                "parallelism": f"env-shard x{world}", "episodes_finished": int(counters[1].item()),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": f"pom_step_kernel<{epw}, {lpe}, {'true' if args.fresh_boards else 'false'}>", "step_ms": step_ms, "algorithmic_bytes_per_step": algo_bytes,
                # one step = `launches_per_step` concurrent launches; per launch: bytes / mean duration (matches rocprofv3's AverageNs)
                "launches_per_step": parts,
                "launch": {"algorithmic_bytes": algo_bytes // parts, "ms": launch_ms, "timed_launches": n_launch,
                           "achieved": (algo_bytes / parts) / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else None},
            },
        }
        if config3:
            line["config3_simple_agent"] = config3
        if other:
            line["other_configs"] = other
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(start, args.seed, dist_id, args.max_steps)
        print(json.dumps(line), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
