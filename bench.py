#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Pommerman tick on MI355X (BASELINE.json's metric).

One "step" = one pass of the hot path over the whole batch: every env of this rank advances by one tick with its state
round-tripping HBM (ticks_per_launch = 1), issued as `launches_per_step` launches of pom_step_kernel over contiguous
sub-batches.  Workload (config.workload): 65,536 concurrent 11x11 FFA envs per GPU, start boards with the reference's cell
distribution, Move[4] i.i.d. uniform over {IDLE,UP,DOWN,LEFT,RIGHT,BOMB} (RandomAgent distribution) from the counter-based
stream of include/pom_rng.h, finished envs restart from their snapshot, 800-tick episode cap.  Inputs are resident in HBM
before the timed region starts; nothing crosses PCIe inside it.

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (one fresh process per GPU, before anything touches HIP);
under `torch.distributed.run` (WORLD_SIZE set) it is one of the ranks.  torch.distributed, backend nccl = RCCL; envs are
sharded contiguously with no data-path collective; the timed region is exactly K steps between two (barrier +
torch.cuda.synchronize()) pairs — a rank's clock stops when its own device is idle, the closing dist.barrier follows, rank 0
prints the MAX over ranks; the only collective besides the barriers is the all-reduce of the step counters, issued inside the
region behind the last step (weak scaling: per-GPU batch fixed).  The reference fans out the same way, one env per std::thread
(unit_test/bboard/performance_test.cpp:40-50,71-94).

The JSON line carries
  roofline     — `achieved` / `frac`: the bytes the kernel REALLY moves per step (`traffic`: rocprofv3 PMC FETCH_SIZE x 2 +
                 WRITE_SIZE of this workload — the figure committed under profiles/, or measured by this very run with
                 --measure-traffic; `traffic_source` / `traffic_measured_in_run` say which; the packed record's footprint
                 2 x 320 B per env for workloads without a PMC figure) over the SAME clock as `value` (ms_per_step), against
                 the 8 TB/s HBM3E peak: a physical utilisation, never above 1;
                 `contract_achieved` / `contract_frac`: SURVEY §8(d)'s contract bytes (2024 B per env-step: the reference's
                 1004-B State read and written + Move[4]) over the same clock — bytes the kernel does not move (the device
                 record is packed to 320 B), kept for comparison with rounds 1-2, not a utilisation;
                 `launch`: one launch's bytes and mean duration from HIP events attached to the dispatch (what rocprofv3's
                 kernel trace reports as AverageNs); `limited_by`: what actually bounds the kernel (profiles/).
  cpu_baseline — the unmodified reference bboard::Step (oracle/_ref, built where /root/reference lies) or the restatement,
                 timed on this host's cores on a bounded sample of the same workload (rank 0, N=1 only); `config1`: BASELINE's
                 config 1 (one env, HarmlessAgent moves, one thread).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 2024  # 1004 B State read + 1004 B State write + 16 B Move[4]  (SURVEY.md §8d): the contract figure
PACKED_BYTES_PER_STEP = 2 * 320  # what the device record moves per env-step: 80 dwords read + written (pom_packed.h)
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_JSON = os.path.join("profiles", "r05_traffic.json")
# BASELINE's metric: 64k envs per GPU.  Round 4: the same with several GPUs — weak scaling with the per-GPU work really fixed, so that
# value(N) / (N x value(1)) compares like with like (until then N > 1 ran config 4's 32,768 per GPU: half the chip's wavefront slots,
# 4.3 G against 7.4 G per GPU whatever the interconnect does).  Config 4 itself (262,144 envs on 8 GPUs) is `--gpus 8 --envs 32768`.
ENVS_SINGLE_GPU, ENVS_PER_GPU_SHARDED = 65536, 65536


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
    if dist_mod is not None and dist_mod.is_initialized():  # (a communicator of one rank too: POM_BENCH_RCCL_SOLO)
        _all_reduce(counters, dist_mod.ReduceOp.SUM, dist_mod)
    return counters


def reduce_max(value: float, device, dist_mod=None) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist_mod is not None and dist_mod.is_initialized():
        _all_reduce(t, dist_mod.ReduceOp.MAX, dist_mod)
    return float(t.item())


def timed_region(run_steps, steps: int, barrier, in_region_reduce=None, device_sync=None) -> float:
    """The timed region: exactly `steps` steps between two barriers (each = dist.barrier + torch.cuda.synchronize()).  With
    several ranks `in_region_reduce` — the path's one collective, the all-reduce of the step counters — is issued INSIDE it,
    behind the last step.  `device_sync` (torch.cuda.synchronize): the clock stops when THIS rank's device is idle — every stream:
    the steps, the sub-batches, the side stream's all-reduce — and the closing barrier's dist.barrier follows outside it: a rank
    reports its own time and rank 0 prints the MAX over ranks, so the wait for the other ranks (and a second collective's latency)
    is not part of anybody's K steps.  Without it (the CPU rehearsal, tests/stub_rank.py) the clock stops behind the barrier.
    Returns this rank's wall time."""
    trace = os.environ.get("POM_BENCH_TRACE") == "1"
    barrier()
    t0 = time.perf_counter()
    run_steps(steps)
    t_a = time.perf_counter()
    if in_region_reduce is not None:
        in_region_reduce()
    t_b = time.perf_counter()
    if device_sync is not None:
        device_sync()
        t1 = time.perf_counter()
        barrier()
    else:
        barrier()
        t1 = time.perf_counter()
    if trace:  # host time of the region's three pieces (the device runs behind the first two)
        print(f"[trace] steps queued {1e6 * (t_a - t0):.0f} us, reduction queued {1e6 * (t_b - t_a):.0f} us, until the device was idle {1e6 * (t1 - t_b):.0f} us",
              file=sys.stderr, flush=True)
    return t1 - t0


# ---- the launcher behind `--gpus N` ---------------------------------------------------------------------------------
def launch_ranks(n_ranks: int, argv: list, worker_cmd: list | None = None, timeout_s: float = 1500.0) -> int:
    """Start one fresh process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relay rank 0's stdout — the JSON
    line — and return non-zero if any rank failed.  The parent never imports torch or touches HIP: every rank initialises
    its GPU in a process of its own (re-executing a process that holds a GPU is not allowed on this pool).  If one rank
    dies, the others are stopped by PID."""
    worker_cmd = worker_cmd or [sys.executable, os.path.abspath(__file__)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this image
        procs.append(subprocess.Popen(worker_cmd + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    deadline = time.time() + timeout_s
    rc = 0
    pending = set(range(n_ranks))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the others", file=sys.stderr)
        if rc != 0 or time.time() > deadline:
            for r in pending:
                procs[r].kill()  # exactly the PIDs started above
            for r in pending:
                procs[r].wait()
            if rc == 0:
                rc = 124
                print("bench.py: timed out waiting for the ranks", file=sys.stderr)
            break
        time.sleep(0.05)
    reader.join(5)
    for line in out0:
        sys.stdout.write(line)
    sys.stdout.flush()
    return rc


def _ref_child(args: list, timeout: float) -> dict:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_baseline_run.py")] + args, capture_output=True, text=True,
                         timeout=timeout)
    return json.loads(out.stdout.strip().splitlines()[-1])


def cpu_baseline(start: np.ndarray, seed: int, dist_id: int, max_steps: int, budget_s: float = 10.0) -> dict:
    """Time the CPU path on the host cores on a bounded sample of the same workload."""
    import ctypes as C
    import tempfile
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
    # BASELINE config 1 with the restatement: one env, harmless moves, one thread, 10 x 1000 ticks
    one, init1 = np.ascontiguousarray(start[:1]).copy(), np.ascontiguousarray(start[:1])
    t1 = time.perf_counter()
    n1 = sum(lib.pom_oracle_run_random(one.ctypes.data, init1.ctypes.data, 1, 1000, seed, 0, 1000 * k, 0, max_steps) for k in range(10))
    w1 = time.perf_counter() - t1
    config1 = {"value": n1 / w1, "unit": "env-steps/s", "cores": 1, "kind": "port",
               "sample": f"1 env, HarmlessAgent move distribution, 1 thread, 10 x 1000 ticks of oracle/pom_oracle.c ({flags})"}
    # The reference itself, where its prebuilt library travelled with the repo (oracle/_ref/, built in the container that holds
    # /root/reference): the unmodified bboard::Step, -O3, timed in a child process (it has UB on reachable states; a guard skips
    # those ticks, and a crash only costs this leg).
    ref_lib = os.path.join(ORACLE_DIR, "_ref", "libpomref_bench.so")
    if os.path.exists(ref_lib):
        try:
            with tempfile.TemporaryDirectory() as td:
                f = os.path.join(td, "boards.npy")
                np.save(f, start[:cores * per_thread])
                r = _ref_child([f, str(seed), str(dist_id), str(max_steps), str(budget_s)], budget_s * 4 + 60)
                try:
                    c1 = _ref_child([f, str(seed), "config1"], 60)
                    if c1.get("value"):
                        config1 = {"value": c1["value"], "unit": "env-steps/s", "cores": 1, "kind": "reference",
                                   "sample": f"1 env, HarmlessAgent move distribution (basic_agents.cpp:28-38), 1 thread, {c1['reps']} x "
                                             f"{c1['ticks']} ticks of the unmodified reference bboard::Step (performance_test.cpp:55-59's shape), "
                                             f"{c1['steps']} env-steps in {c1['timed_s'] * 1e3:.2f} ms; restatement: {n1 / w1:.3g}"}
                except Exception:
                    pass
            return {
                "value": r["value"], "unit": "env-steps/s", "cores": r["cores"], "kind": "reference",
                "sample": f"unmodified reference bboard::Step (oracle/_ref/libpomref_bench.so, g++ -O3), {r['cores']} threads x "
                          f"{r['per_thread']} envs of the same boards and move stream, {r['steps']} env-steps in "
                          f"{r['timed_s_per_thread']:.1f} s of reference time per thread ({r['skipped_ub_ticks']} ticks with reference UB "
                          f"stepped by the restatement, untimed)",
                "port": port, "config1": config1,
            }
        except Exception as exc:  # the figure below is still a measured baseline
            port["reference_leg"] = f"failed: {type(exc).__name__}"
    port["config1"] = config1
    return port


def measure_traffic(args) -> dict | None:
    """--measure-traffic: HBM bytes per step of THIS build on THIS workload, from the PMC counters: two child runs of this script under rocprofv3
    (`--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: separate passes, counters only, no tracing); a chained launch covers the whole batch,
    one tick (with sub-batches — `--streams N` under another issue mode — a step is N dispatches: pass `--streams 1` for a per-step figure).  gfx950: FETCH_SIZE reports half of a coalesced streaming read (MI355X guide;
    calibrated in round 1 on a zero-tick launch of this kernel: 0.513), both counters are in KB.  None if the profiler is not
    there or a pass fails — the caller then falls back to the committed measurement."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"failed": "rocprofv3 not found"}
    # (the launches as the parent issues them — chained by default: the profiler plays dispatches one at a time while it counts)
    child = [sys.executable, os.path.abspath(__file__), "--traffic-probe", "--streams", str(args.streams), "--steps", "60", "--warmup", "10",
             "--envs", str(args.envs), "--kind", args.kind, "--dist", args.dist, "--seed", str(args.seed), "--max-steps", str(args.max_steps),
             "--burn-in", str(args.burn_in), "--no-cpu-baseline", "--no-config3"]  # (direct launches: 60 steps = 3 graph chunks would also do)
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(td, counter)
                subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env, check=True,
                               capture_output=True, timeout=120)
                vals = []
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if "pom_step_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            vals.append(float(row["Counter_Value"]))
                if len(vals) < 40:
                    return {"failed": f"{counter} pass: only {len(vals)} dispatches of pom_step_kernel in the counter csv"}
                vals = vals[len(vals) // 2:]  # the steady second half: past the burn-in
                out[counter] = sum(vals) / len(vals)
    except Exception as exc:
        return {"failed": f"{type(exc).__name__}: {str(exc)[:200]}"}
    # FETCH_SIZE x 2: /opt/skills/guides/MI355X_MICROARCH.md ("on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
    # streaming read"); calibrated on a zero-tick launch of this kernel in round 1 (0.513, profiles/r01_traffic.json)
    bytes_per_step = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
    return {"hbm_bytes_per_step": int(round(bytes_per_step)), "fetch_size_kb_raw": out["FETCH_SIZE"], "write_size_kb_raw": out["WRITE_SIZE"],
            "fetch_correction": 2.0}


def config_traffic(key: str, envs: int):
    """HBM bytes per step of the workload `key` from the committed PMC passes (profiles/r05_traffic.json: one block per config, FETCH_SIZE x 2 +
    WRITE_SIZE of ITS kernel on ITS workload, scripts/profile_configs.sh + scripts/traffic_json.py), scaled by envs only if the block was taken
    at another batch size.  (bytes, what they are) — the packed record's footprint if the config has no block."""
    try:
        blk = json.load(open(os.path.join(ROOT, TRAFFIC_JSON)))["configs"][key]
        b = blk["hbm_bytes_per_step"] * envs / blk["envs"]
        kind = "pmc (" + key + ")" + ("" if blk["envs"] == envs else f", scaled from {blk['envs']} envs")
        return int(round(b)), kind
    except (OSError, KeyError, ValueError):
        return PACKED_BYTES_PER_STEP * envs, "packed footprint (2 x 320 B per env)"


def with_bytes(entry: dict, key: str, envs: int, ms_per_step: float) -> dict:
    """`entry` plus its own bytes_per_step / achieved_GBps / frac (of the 8 TB/s peak), from ITS config's PMC block"""
    b, kind = config_traffic(key, envs)
    entry.update({"bytes_per_step": b, "bytes_kind": kind, "achieved_GBps": b / (ms_per_step * 1e-3) / 1e9,
                  "frac": b / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    return entry


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=0,
                    help="envs per GPU (default: 65,536 — the headline's batch — at every N; BASELINE config 4's 262,144 envs on 8 GPUs: "
                         "--gpus 8 --envs 32768)")
    ap.add_argument("--kind", default="ffa", choices=["ffa", "stress"])
    ap.add_argument("--dist", default="random", choices=["harmless", "random", "stress"])
    ap.add_argument("--ticks-per-launch", type=int, default=1)
    ap.add_argument("--max-steps", type=int, default=800)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", default="random", choices=["random", "simple", "tape"],
                    help="random: Move[4] from the counter stream (--dist); simple: the device SimpleAgent policy (config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the other-config segments (profiling runs: one kernel shape only)")
    ap.add_argument("--streams", type=int, default=0,
                    help="sub-batches per step (0 = the library's default on one GPU; with several ranks, measured in an untimed pass)")
    ap.add_argument("--burn-in", type=int, default=300,
                    help="untimed ticks played before the warm-up so that the batch is a steady mix of openings, mid-games and restarts "
                         "(all games start at tick 0 together: nothing explodes before tick 10 and the first deaths come in waves)")
    ap.add_argument("--fresh-boards", action="store_true",
                    help="boards drawn on the device (pom_batch_generate) and a new one per episode instead of the snapshot replay "
                         "BASELINE's configs prescribe (SURVEY §8 f3)")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="measure roofline.traffic in this run (two rocprofv3 --pmc child runs of this script, +10-20 s); default: the "
                         "figure committed under profiles/ for this workload")
    ap.add_argument("--no-traffic", action="store_true", help=argparse.SUPPRESS)  # accepted for older scripts: the default now
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)  # the child of measure_traffic(): steps only, no JSON extras
    ap.add_argument("--envs-per-wave", type=int, default=0)
    ap.add_argument("--lanes-per-env", type=int, default=0)
    args = ap.parse_args(argv)
    if args.envs <= 0:
        world = int(os.environ.get("WORLD_SIZE", "0")) or args.gpus
        args.envs = ENVS_SINGLE_GPU if world <= 1 else ENVS_PER_GPU_SHARDED
    return args


def worker(args) -> None:
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    entry.build_hip()
    import pomcpp_amd as pa
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, CNT_STEPS

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the stepper has no CPU path")
    # POM_BENCH_REHEARSAL=1: the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices, gloo instead of
    # RCCL, which refuses two ranks on one device).  For rehearsing the launch / barrier / reduction logic, never for numbers.
    rehearsal = os.environ.get("POM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPUs are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = "none"
    # POM_BENCH_RCCL_SOLO=1: everything the N > 1 path does — the RCCL communicator, barriers, the tuning vote, the in-region
    # all-reduce on a side stream — with a communicator of ONE rank: the only way to run that code under RCCL on a one-GPU box
    multi = world > 1 or os.environ.get("POM_BENCH_RCCL_SOLO") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (a launcher sets it; a lone POM_BENCH_RCCL_SOLO run takes a port nobody holds)
            with socket.socket() as s_port:
                s_port.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_port.getsockname()[1])
        backend = "gloo" if rehearsal else "nccl"
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
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    tpl = args.ticks_per_launch
    if args.policy == "simple":
        if tpl != 1:
            raise SystemExit("--policy simple runs one tick per launch")

        def run_steps(k: int) -> None:  # k steps: k x launches_per_step launches, queued by one call into the library
            env.step_simple(args.seed, k)
    elif args.policy == "tape":
        # explicit Move[4] from a tape in device memory (pom_batch_step_device_many: chained launches, the tile's ticket picks the
        # tape's tick) as the MAIN workload: what scripts/profile_configs.sh profiles as `tape`
        if tpl != 1:
            raise SystemExit("--policy tape runs one tick per launch")
        tape_main = torch.randint(0, 6, (max(args.steps, args.warmup, args.burn_in, 1), plan["n_envs"], 4), dtype=torch.int32, device=device,
                                  generator=torch.Generator(device=device).manual_seed(args.seed))

        def run_steps(k: int) -> None:
            env.step_device_many(tape_main[:k])
    else:
        def run_steps(k: int) -> None:
            env.step_random(args.seed, dist_id, ticks=k * tpl, ticks_per_launch=tpl)
    # Input preparation, untimed: play the fresh games forward until the batch is a steady mix of game phases.  (The step time is
    # flat from tick ~100 on, scripts/phase_drift.py; the first ticks of 65,536 synchronised openings are much cheaper.)
    if args.burn_in > 0:
        run_steps(args.burn_in)
        env.sync()
    tuned, tuning_steps = None, 0
    if args.streams == 0 and multi:
        # Untimed: how many sub-batches per step?  More parts overlap more load/store with compute, but ROCm maps all streams
        # of the process onto 4 hardware queues and parts that share a queue serialize (profiles/r01_streams.txt) — how many
        # are free depends on the process (torch, RCCL's streams), so with a communicator in the process measure instead of
        # guessing.  A single-GPU run uses the library's default (3 from 49,152 envs up).  Results do not depend on the choice.
        tuned = {}
        k_tune = max(10, min(args.steps, 100))  # calls of the length the timed region will issue: a third stream pays from ~50 ticks up
        for k in (3, 2, 1):  # first touch of a sub-stream creates its hardware queue (~10 ms once): keep that out of the timings
            env.set_streams(k)
            run_steps(5)
            env.sync()
            tuning_steps += 5
        for k in (2, 3, 1):
            env.set_streams(k)
            best_k = None
            for _ in range(2):  # the better of two short runs: one stall must not decide the shape
                run_steps(10)
                env.sync()
                t_a = time.perf_counter()
                run_steps(k_tune)
                env.sync()
                dt = (time.perf_counter() - t_a) / k_tune * 1e3
                best_k = dt if best_k is None else min(best_k, dt)
                tuning_steps += 10 + k_tune
            tuned[k] = best_k
        votes = torch.tensor([tuned[1], tuned[2], tuned[3]], dtype=torch.float64, device=device)
        _all_reduce(votes, dist.ReduceOp.MAX, dist)  # every rank must run the same shape: the slowest rank's best
        env.set_streams(int(torch.argmin(votes).item()) + 1)
    run_steps(args.warmup)
    side = torch.cuda.Stream(device=device) if multi else None
    if multi:
        # warm what the timed region will use, the way it will use it: the side stream (its first use creates a hardware queue:
        # 0.66 ms of host time, scripts/experiments/rccl/join_timing.py) and RCCL's all-reduce issued from it
        env.counters_into(counters.data_ptr())
        side.wait_stream(stream)
        with torch.cuda.stream(side):
            reduce_counters(counters, dist)
        stream.wait_stream(side)
        torch.cuda.synchronize()
    env.reset_counters()

    # The timed region: exactly K steps between two (barrier + torch.cuda.synchronize()) pairs.  With several ranks it also holds
    # the path's one collective, where north_star puts it: the end-of-region reduction of the step / episode counters — the
    # per-wavefront counters summed on the device behind the last step (on the launch stream), then ONE 32-byte all-reduce
    # (RCCL over xGMI) on a side stream.  Envs shard with no exchange on the data path, so this is all the ranks ever say to
    # each other.  On one GPU there is no collective; the counters are read after the region.
    reduce_in_region = multi

    def counters_allreduce() -> None:
        env.counters_into(counters.data_ptr())  # joins the sub-batches, one reduction kernel, on the launch stream
        side.wait_stream(stream)
        with torch.cuda.stream(side):
            reduce_counters(counters, dist)

    env.fork()  # the sub-streams are ordered behind the setup above now, not inside the timed region
    # (torch.cuda.synchronize() waits for every stream of the device: the sub-batches' and the side stream's all-reduce too)
    elapsed = timed_region(run_steps, args.steps, barrier, counters_allreduce if reduce_in_region else None, torch.cuda.synchronize)
    elapsed = reduce_max(elapsed, device, dist)
    if not reduce_in_region:  # one GPU: bookkeeping after the region (what the K steps did)
        env.sync()  # (also the check behind chained launches: a tile a wavefront could not play is caught up, and counted, here)
        env.counters_into(counters.data_ptr())
        torch.cuda.synchronize()
    total_steps = int(counters[CNT_STEPS].item())
    if reduce_in_region and total_steps != plan["global_envs"] * args.steps * tpl:
        # the in-region reduction does not block, so it cannot run that check: should a tile have been left behind (never seen), every
        # rank sees the same short total and all of them reduce again after catching up
        env.sync()
        env.counters_into(counters.data_ptr())
        torch.cuda.synchronize()
        reduce_counters(counters, dist)
        total_steps = int(counters[CNT_STEPS].item())
    episodes_finished = int(counters[1].item())
    # the same number of steps again between two HIP events on the launch stream (untimed: the events and the join they need
    # would add their own latency to the region above): GPU-side time per step, all launches of a step
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record(stream)
    run_steps(args.steps)
    env.flush()  # steps run as sub-batches on internal streams: order them before the event on this stream
    ev1.record(stream)
    torch.cuda.synchronize()
    step_ms = ev0.elapsed_time(ev1) / args.steps
    step_ms = reduce_max(step_ms, device, dist)
    # a step is issued as `parts` launches of pom_step_kernel over contiguous sub-batches on parallel streams; time the
    # individual launches too (HIP events attached to each dispatch), outside the timed region
    epw, lpe, parts = env.launch_shape()
    issue, issue_streams = env.issue_info()
    env.profile(True)
    run_steps(max(1, 256 // parts))
    launch_ms, n_launch = env.profile_read()
    env.profile(False)

    # BASELINE config 3 beside the headline (rank 0 of a single-GPU run, untimed region): the same boards played from the start
    # by the device SimpleAgent policy, act x4 + Step per env-step as Environment::Step does.  200 untimed ticks first: games
    # last ~190 ticks under this policy, so the batch is then a steady mix of openings, mid-games and restarts.
    config3 = None
    if not multi and args.policy == "random" and tpl == 1 and not args.no_config3:
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
        config3 = with_bytes({"workload": f"{args.envs} envs, 4x SimpleAgent policy on the device + Step, games from the start, "
                                          "200 warm-up ticks", "value": plan["n_envs"] / (ms_c * 1e-3),
                              "unit": "env-steps/s", "ms_per_step": ms_c, "steps": n_c}, "c3", plan["n_envs"], ms_c)
    # BASELINE's other single-GPU configs, briefly (untimed region, default run only): config 2 (4,096 envs, random moves) and
    # config 5 (65,536 envs, kick / chain-explosion stress boards and move mix) — parity-test cases first, numbers for context —
    # and the explicit-moves path an RL loop uses (pom_batch_step_device: Move[4] read from device memory, every tick joined
    # with the caller's stream, so no pipelining across ticks).
    other = None
    default_run = (args.envs == 65536 and args.kind == "ffa" and args.dist == "random" and not args.fresh_boards)
    if not multi and args.policy == "random" and tpl == 1 and not args.no_config3 and default_run:
        other = {}
        for name, n_o, kind_o, dist_o in (("config2_4096_envs_random", 4096, "ffa", 1), ("config5_65536_envs_stress", 65536, "stress", 2)):
            e2 = BatchEnvironment(n_o, device=local_rank, mode=MODE_ENV, auto_reset=True, max_steps=args.max_steps,
                                  stream=stream.cuda_stream)
            e2.make_game(pa.make_boards(n_o, seed=args.seed * 1000003, kind=kind_o))
            e2.step_random(args.seed, dist_o, ticks=300)
            e2.sync()
            ev4, ev5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_o_steps = 200
            ev4.record(stream)
            e2.step_random(args.seed, dist_o, ticks=n_o_steps)
            e2.flush()
            ev5.record(stream)
            e2.sync()
            ms_o = ev4.elapsed_time(ev5) / n_o_steps
            other[name] = with_bytes({"value": n_o / (ms_o * 1e-3), "unit": "env-steps/s", "ms_per_step": ms_o, "steps": n_o_steps},
                                     "c2" if n_o == 4096 else "c5", n_o, ms_o)
            e2.close()
        # explicit moves from device memory, as an RL loop steps: auto_reset = RESET_AT_END (what is observed is what the next move
        # applies to), 8 pre-drawn Move[4] arrays cycled, one pom_batch_step_device per tick (one launch, joined with the
        # caller's stream every tick)
        from pomcpp_amd.batch import RESET_AT_END
        gen = torch.Generator(device=device).manual_seed(args.seed)
        mv_dev = torch.randint(0, 6, (8, plan["n_envs"], 4), dtype=torch.int32, device=device, generator=gen)
        e3 = BatchEnvironment(plan["n_envs"], device=local_rank, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=args.max_steps,
                              stream=stream.cuda_stream)
        e3.make_game(start)
        torch.cuda.synchronize()
        for t in range(300):
            e3.step_device(mv_dev[t % 8].data_ptr())
        e3.sync()
        ev6, ev7 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_x = 200
        ev6.record(stream)
        for t in range(n_x):
            e3.step_device(mv_dev[t % 8].data_ptr())
        ev7.record(stream)
        e3.sync()
        ms_x = ev6.elapsed_time(ev7) / n_x
        # the same moves as a K-tick TAPE (pom_batch_step_device_many): Step(State*, Move[4]) with the moves fixed K ticks ahead — replays,
        # open-loop rollouts — issued as chained launches like the headline (the kernel reads tick d of the tape by its ticket)
        n_tape = 200
        tape = torch.randint(0, 6, (n_tape, plan["n_envs"], 4), dtype=torch.int32, device=device, generator=gen)
        torch.cuda.synchronize()
        e3.step_device_many(tape)
        e3.sync()
        ev6.record(stream)
        e3.step_device_many(tape)
        e3.flush()
        ev7.record(stream)
        e3.sync()
        ms_tape = ev6.elapsed_time(ev7) / n_tape
        tape_stats = e3.chain_stats()
        # an RL tick: explicit moves in, the uint8 global planes + attributes of the resulting state out — as two launches
        # (pom_batch_step_device, pom_batch_observe) and as one (pom_batch_step_device_observe: the observation is written while the tile
        # is still in LDS)
        planes, a_at, e_at = e3.observe()
        ms_obs = {}
        for fused in (False, True):
            for rep in range(2):  # the second pass is the measurement
                ev6.record(stream)
                for t in range(n_x):
                    if fused:
                        e3.step_device_observe(mv_dev[t % 8], out=planes, attrs=False)
                    else:
                        e3.step_device(mv_dev[t % 8].data_ptr())
                        e3.observe(out=planes, attrs=False)
                ev7.record(stream)
                e3.sync()
            ms_obs[fused] = ev6.elapsed_time(ev7) / n_x
        # ... and with the compact observation (POM_OBS_CODES: uint8 [5][11][11], 605 B per env instead of 1,936)
        codes, _, _ = e3.observe(dtype="codes", attrs=False)
        for rep in range(2):
            ev6.record(stream)
            for t in range(n_x):
                e3.step_device_observe(mv_dev[t % 8], dtype="codes", out=codes, attrs=False)
            ev7.record(stream)
            e3.sync()
        ms_obs["codes"] = ev6.elapsed_time(ev7) / n_x
        e3.close()
        del tape, planes, a_at, e_at, codes
        # CLOSED LOOP (the reference's real call shape, Environment::Step: this tick's Move[4] from the caller every tick,
        # environment.cpp:139-149): the batch cut into R ranges, each carrying  policy(range) -> step(range) -> policy(range) ...  on a
        # stream of its own (pom_batch_step_device_range), so that one range's policy runs while the others step; K ticks captured into
        # one HIP graph and replayed.  The policy is a stand-in kernel (pom_bench_policy): with `codes` it reads every byte of the
        # fused POM_OBS_CODES observation the step of the same range wrote one tick earlier, and draws Move[4] from it.
        from pomcpp_amd.batch import bench_policy
        loop = {}
        n_ranges = int(os.environ.get("POM_BENCH_LOOP_RANGES", "2"))  # (profiles/r05_closed_loop_sweep.txt: a launch costs ~3 us however it is issued)
        k_loop, reps_loop = 25, 8
        per = plan["n_envs"] // n_ranges // 16 * 16
        ranges = [(i * per, per if i < n_ranges - 1 else plan["n_envs"] - i * per) for i in range(n_ranges)]
        for with_obs in (False, True):
            e5 = BatchEnvironment(plan["n_envs"], device=local_rank, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=args.max_steps,
                                  stream=stream.cuda_stream)
            e5.make_game(start)
            e5.step_random(args.seed, dist_id, ticks=300)  # a steady mix of game phases
            codes5 = e5.observe(dtype="codes", attrs=False)[0] if with_obs else None
            moves5 = torch.zeros((plan["n_envs"], 4), dtype=torch.int32, device=device)
            e5.sync()
            torch.cuda.synchronize()
            side = [torch.cuda.Stream(device=device) for _ in ranges]
            main5 = torch.cuda.Stream(device=device)

            def issue_loop(main_s):
                for s5 in side:
                    s5.wait_stream(main_s)
                for t in range(k_loop):
                    for (f5, c5), s5 in zip(ranges, side):
                        bench_policy(codes5, moves5, f5, c5, t, s5)
                        e5.step_device_range(f5, c5, moves5, s5, codes=codes5)
                for s5 in side:
                    main_s.wait_stream(s5)

            g5 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g5, stream=main5):
                issue_loop(main5)
            cnt0 = int(e5.counters()[0])
            g5.replay()
            torch.cuda.synchronize()
            evl0, evl1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(main5):
                evl0.record(main5)
                for _ in range(reps_loop):
                    g5.replay()
                evl1.record(main5)
            torch.cuda.synchronize()
            ms_l = evl0.elapsed_time(evl1) / (reps_loop * k_loop)
            played = int(e5.counters()[0]) - cnt0
            assert played == plan["n_envs"] * k_loop * (reps_loop + 1), (played, plan["n_envs"] * k_loop * (reps_loop + 1))
            loop["codes" if with_obs else "plain"] = ms_l
            e5.close()
            del g5, codes5, moves5
        other["closed_loop_65536_envs"] = {
            "value": plan["n_envs"] / (loop["plain"] * 1e-3), "unit": "env-steps/s", "us_per_tick": loop["plain"] * 1e3,
            "value_with_codes_observation": plan["n_envs"] / (loop["codes"] * 1e-3), "us_per_tick_with_codes_observation": loop["codes"] * 1e3,
            "ranges": n_ranges, "ticks_per_graph": k_loop, "graph_replays_timed": reps_loop,
            "note": "pom_batch_step_device_range: per tick and range one launch of a stand-in device policy (pom_bench_policy: writes this "
                    "tick's Move[4]; *_with_codes_observation: after reading every byte of the range's fused POM_OBS_CODES observation of "
                    "the tick before) and one launch of the step (with the observation fused), each range on its own stream, the whole loop "
                    "a replayed HIP graph; POM_RESET_AT_END; every env-step counted by the device counters"}
        # Throughput mode: ticks_per_launch = T > 1 keeps the record in LDS for T ticks (synthetic move stream only) —
        # NOT the canonical roofline run (a step there is one HBM round trip per tick); reported on its own, per tick
        for name, n_o, t_o in (("throughput_T4_65536_envs", 65536, 4), ("throughput_T16_65536_envs", 65536, 16),
                               ("throughput_T16_4096_envs", 4096, 16)):
            e4 = BatchEnvironment(n_o, device=local_rank, mode=MODE_ENV, auto_reset=True, max_steps=args.max_steps, stream=stream.cuda_stream)
            e4.make_game(pa.make_boards(n_o, seed=args.seed * 1000003, kind="ffa"))
            launches = 40
            e4.step_random(args.seed, 1, ticks=320, ticks_per_launch=t_o)
            e4.sync()
            ev8, ev9 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev8.record(stream)
            e4.step_random(args.seed, 1, ticks=launches * t_o, ticks_per_launch=t_o)
            e4.flush()
            ev9.record(stream)
            e4.sync()
            ms_t = ev8.elapsed_time(ev9) / (launches * t_o)
            other[name] = {"value": n_o / (ms_t * 1e-3), "unit": "env-steps/s", "ms_per_tick": ms_t, "ticks_per_launch": t_o,
                           "ticks": launches * t_o,
                           "note": "ticks_per_launch > 1: the tile stays in LDS between ticks; not the canonical one-round-trip-per-tick step"}
            e4.close()
        # the literal drop-in, one env at a time: pom_step (= bboard::Step(State*, Move*) behind the C-ABI: the State in a pinned
        # page, ONE launch that reads, steps and writes it back, the host polling the kernel's last store).  Latency, not
        # throughput: compare with cpu_baseline.config1.  Called through ctypes with the pointers prepared (≈ 1 us of Python per call).
        from pomcpp_amd.batch import load_library
        lib1 = load_library()
        one = np.ascontiguousarray(start[:1]).copy()
        mv1 = np.zeros(4, dtype=np.int32)
        p_state, p_moves = one.ctypes.data, mv1.ctypes.data
        for _ in range(20):
            assert lib1.pom_step(p_state, p_moves) == 0
        t_l = time.perf_counter()
        n_l = 1000
        for k in range(n_l):
            mv1[0] = k % 5  # HarmlessAgent's range: no bombs
            lib1.pom_step(p_state, p_moves)
        us_l = (time.perf_counter() - t_l) / n_l * 1e6
        threads8 = None
        try:  # eight native threads, a State each (tests/cpp/step_threads.cpp), as performance_test.cpp:71-94 steps one env per std::thread
            import tempfile
            exe = entry.build_step_threads()
            with tempfile.TemporaryDirectory() as td:
                nt, calls = 8, 2000
                start[:nt].tofile(os.path.join(td, "s.bin"))
                np.random.default_rng(1).integers(0, 5, size=(nt, calls, 4), dtype=np.int32).tofile(os.path.join(td, "m.bin"))
                out_t = subprocess.run([exe, str(nt), str(calls), os.path.join(td, "s.bin"), os.path.join(td, "m.bin")], capture_output=True,
                                       text=True, timeout=120)
                rt = json.loads(out_t.stdout.strip().splitlines()[-1])
                threads8 = {"threads": nt, "calls_per_s_1_thread": rt["calls_per_s_1_thread"], "calls_per_s_all_threads": rt["calls_per_s_all_threads"],
                            "speedup": rt["calls_per_s_all_threads"] / rt["calls_per_s_1_thread"]}
        except Exception as exc:  # the figure is context, not the headline
            threads8 = {"failed": f"{type(exc).__name__}: {str(exc)[:120]}"}
        other["single_env_pom_step"] = {"value": 1e6 / us_l, "unit": "env-steps/s", "us_per_call": us_l, "calls": n_l, "native_threads": threads8,
                                        "note": "pom_step(State*, Move[4]) on ONE env: one launch reads the host State, steps it and writes "
                                                "it back, blocking; the batch API is the product, this is the literal bboard::Step replacement"}
        other["explicit_moves_device_chained_65536_envs"] = with_bytes({
            "value": plan["n_envs"] / (ms_tape * 1e-3), "unit": "env-steps/s", "ms_per_step": ms_tape, "steps": n_tape,
            "chain_tiles_recovered": tape_stats["tiles_recovered"]}, "tape", plan["n_envs"], ms_tape)
        other["explicit_moves_device_chained_65536_envs"].update({
            "note": "pom_batch_step_device_many with auto_reset = POM_RESET_AT_END: a 200-tick Move[4] tape in device memory, chained launches "
                    "(one launch over all tiles per tick on rotating streams, the tile's ticket picks the tape's tick)"})
        other["step_plus_observation_65536_envs"] = {
            "one_launch_us": ms_obs[True] * 1e3, "two_launches_us": ms_obs[False] * 1e3, "value": plan["n_envs"] / (ms_obs[True] * 1e-3),
            "unit": "env-steps/s (each with its uint8 [16][11][11] observation written)",
            "one_launch_codes_us": ms_obs["codes"] * 1e3, "value_codes": plan["n_envs"] / (ms_obs["codes"] * 1e-3),
            "note": "pom_batch_step_device_observe against pom_batch_step_device + pom_batch_observe, explicit moves, POM_RESET_AT_END; "
                    "*_codes: the compact observation (POM_OBS_CODES, uint8 [5][11][11]) in the same launch"}
        other["explicit_moves_device_65536_envs"] = with_bytes({
            "value": plan["n_envs"] / (ms_x * 1e-3), "unit": "env-steps/s", "ms_per_step": ms_x, "steps": n_x,
            "note": "pom_batch_step_device with auto_reset = POM_RESET_AT_END: Move[4] from device memory, one launch per tick on the caller's stream"},
            "head1", plan["n_envs"], ms_x)
    if other is not None:
        # what a LONG call reaches (the timed region above is the driver's shape, a few hundred microseconds from an idle device: a fifth
        # of it is the pipeline filling and draining): 500 steps in one call, HIP events on the launch stream
        evA, evB = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        env.make_game(start)  # (config 3 above left SimpleAgent games in the batch)
        env.set_tick(0)
        run_steps(args.burn_in)
        env.sync()
        evA.record(stream)
        run_steps(500)
        env.flush()
        evB.record(stream)
        env.sync()
        ms_500 = evA.elapsed_time(evB) / 500
        other["headline_500_steps"] = with_bytes({
            "value": plan["n_envs"] / (ms_500 * 1e-3), "unit": "env-steps/s", "us_per_step": ms_500 * 1e3, "steps": 500,
            "cache_resident": True,
            "note": "the headline workload in one 500-step call (chained launches, three streams); bytes = the committed PMC figure "
                    "(L2 <-> fabric: FETCH_SIZE x 2 + WRITE_SIZE); state + snapshot (43 MB) sit inside the 256 MiB memory-side cache"},
            "headc", plan["n_envs"], ms_500)
        # beyond the memory-side cache: 1,048,576 envs = 470 MB of state (+ 470 MB of snapshots): every step streams the records from and
        # to HBM proper.  Boards drawn on the device (same distribution), snapshot replay, sub-batches on parallel streams.
        n_big = 1048576
        eb = BatchEnvironment(n_big, device=local_rank, mode=MODE_ENV, auto_reset=True, max_steps=args.max_steps, stream=stream.cuda_stream)
        eb.generate(args.seed)
        eb.step_random(args.seed, dist_id, ticks=300)
        eb.sync()
        ms_big = None
        for _ in range(3):  # the best of three 60-step segments (a GB of fresh allocations: the first segment can still be settling)
            evA.record(stream)
            eb.step_random(args.seed, dist_id, ticks=60)
            eb.flush()
            evB.record(stream)
            eb.sync()
            seg = evA.elapsed_time(evB) / 60
            ms_big = seg if ms_big is None else min(ms_big, seg)
        big_issue = eb.issue_info()
        eb.close()
        other["headline_1048576_envs"] = with_bytes({
            "value": n_big / (ms_big * 1e-3), "unit": "env-steps/s", "us_per_step": ms_big * 1e3, "steps": 60, "best_of_segments": 3, "issue": big_issue[0],
            "cache_resident": False,
            "note": "1,048,576 envs: 344 MB of records (+ as much of snapshots), beyond the 256 MiB memory-side cache — the fraction of the HBM "
                    "peak proper; bytes = this size's own PMC passes"}, "big", n_big, ms_big)
    # like-for-like base of an N > 1 line: this rank's shard stepped while the other ranks' GPUs idle (same envs per GPU, same kernels)
    single_base = None
    if multi:
        barrier()
        if rank == 0:
            evA, evB = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run_steps(args.warmup)
            env.sync()
            evA.record(stream)
            run_steps(args.steps)
            env.flush()
            evB.record(stream)
            env.sync()
            ms_b = evA.elapsed_time(evB) / args.steps
            single_base = {"value": plan["n_envs"] / (ms_b * 1e-3), "unit": "env-steps/s", "ms_per_step": ms_b, "steps": args.steps,
                           "note": f"rank 0's shard ({plan['n_envs']} envs) stepped alone while the other ranks wait: what `--gpus 1 --envs "
                                   f"{plan['n_envs']}` measures (HIP events); N x this is the linear-scaling line"}
        barrier()
    expect = plan["global_envs"] * args.steps * tpl
    if total_steps != expect:
        raise SystemExit(f"step counter {total_steps} != envs x ticks {expect}")

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        algo_bytes = ALGO_BYTES_PER_STEP * plan["n_envs"] * tpl
        footprint = PACKED_BYTES_PER_STEP * plan["n_envs"]  # per launch group: the record is read and written once whatever tpl
        contract = algo_bytes / (ms_per_step * 1e-3) / 1e9  # same clock as `value`
        # PMC-derived HBM bytes per step: the committed rocprofv3 passes over this very workload (profiles/, taken with
        # scripts/profile_configs.sh), or --measure-traffic: the same two passes as child runs of this script.  Workloads without a
        # PMC figure are priced with the packed record's footprint (the kernel reads and writes each 320-B record once).
        traffic, traffic_source, measured_in_run, traffic_failure = None, None, False, None
        tj = os.path.join(ROOT, TRAFFIC_JSON)
        headline_shape = (tpl == 1 and args.kind == "ffa" and args.dist == "random" and args.policy == "random" and not args.fresh_boards)
        if args.measure_traffic and not multi and tpl == 1 and not args.traffic_probe:
            live = measure_traffic(args)
            if live and "failed" not in live:
                traffic, measured_in_run = live["hbm_bytes_per_step"], True
                traffic_source = (f"measured by this run: rocprofv3 --pmc FETCH_SIZE ({live['fetch_size_kb_raw']:.1f} KB raw per step, x 2: the gfx950 "
                                  f"correction) and --pmc WRITE_SIZE ({live['write_size_kb_raw']:.1f} KB) in two separate child runs of this script "
                                  "with one launch per step")
            else:
                traffic_failure = (live or {}).get("failed", "unknown")
                print(f"bench.py: --measure-traffic FAILED ({traffic_failure}); falling back to the committed figure", file=sys.stderr)
        if traffic is None and os.path.exists(tj) and tpl == 1 and not args.fresh_boards:
            # the block of THIS workload (one per config, each from PMC passes over its own kernel and batch): bytes per env do not depend on how
            # many ranks share the job, so a block taken at another batch size of the same kind is scaled by envs — and says so
            key = ("c3" if args.policy == "simple" else "tape" if args.policy == "tape" else "c5" if args.kind == "stress" else
                   None if not headline_shape else "c2" if plan["n_envs"] <= 8192 else "big" if plan["n_envs"] > 262144 else
                   "headc" if issue == "chain" else "head1")
            blk = json.load(open(tj)).get("configs", {}).get(key) if key else None
            if blk:
                traffic = int(round(blk["hbm_bytes_per_step"] * plan["n_envs"] / blk["envs"]))
                traffic_source = (f"{TRAFFIC_JSON} configs.{key} (rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE in separate passes over this workload, "
                                  f"{blk['envs']} envs" + ("" if blk["envs"] == plan["n_envs"] else f", scaled to this rank's {plan['n_envs']}")
                                  + "; committed, not measured in this run)")
        moved = traffic if traffic is not None else footprint
        hbm = moved / (ms_per_step * 1e-3) / 1e9
        c4 = world > 1 and plan["global_envs"] == 262144
        line = {
            "metric": "env_steps_per_sec", "value": total_steps / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, not a measurement)" if rehearsal else ""),
            "config": {
                "workload": (f"BASELINE config 4: {plan['global_envs']} concurrent 11x11 FFA envs sharded over {world} GPUs, " if c4 else "")
                            + f"{args.envs} concurrent 11x11 FFA envs per GPU, {args.kind} boards, "
                            + (f"uniform-{args.dist} Move[4] (RandomAgent distribution)" if args.policy == "random"
                               else "4x SimpleAgent policy on the device (act x4 + Step per env-step, as Environment::Step)")
                            + (", auto-reset onto a fresh device-generated board" if args.fresh_boards else ", auto-reset")
                            + f", {args.max_steps}-tick cap",
                "policy": args.policy,
                "envs_per_gpu": args.envs, "global_envs": plan["global_envs"], "ticks_per_launch": tpl,
                "envs_per_wave": epw, "lanes_per_env": lpe, "launches_per_step": parts,
                # chain: every launch covers ALL tiles and plays one tick, consecutive launches go to different streams, a ticket
                # word per tile orders that tile's ticks (pomcpp_amd/csrc/pom_chain.h); threads / direct / graph: `launches_per_step`
                # sub-batches per tick on parallel streams
                "issue": issue, "issue_streams": issue_streams,
                "burn_in_ticks": args.burn_in, "launches_per_step_tuning_ms": tuned, "untimed_tuning_steps": tuning_steps,
                "parallelism": f"env-shard x{world}", "ranks": world, "collective_backend": backend,
                "rccl_ranks": dist.get_world_size() if multi else 1,
                "collective_in_timed_region": ("one 32-byte all-reduce(SUM) of the step / episode counters behind the last step, on a side stream"
                                               if reduce_in_region else None),
                "per_gpu_batch_note": (f"{ENVS_PER_GPU_SHARDED} envs per GPU at every N (the headline's batch: weak scaling with the per-GPU "
                                       "work fixed); BASELINE config 4 (262,144 envs on 8 GPUs) is `--gpus 8 --envs 32768`"),
                "episodes_finished": episodes_finished,
                # which of BASELINE.json's configs this line is (None: a shape BASELINE does not list, e.g. 65,536 envs on each of N > 1 GPUs)
                "baseline_config": (4 if c4 else 3 if (args.policy == "simple" and args.envs == 65536 and world == 1) else
                                    5 if (args.kind == "stress" and args.envs == 65536 and world == 1) else
                                    2 if (args.envs == 4096 and world == 1 and args.policy == "random" and args.kind == "ffa") else None),
                "single_gpu_base": single_base,
            },
            "roofline": {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                # physical: bytes really moved per step over the clock `value` uses
                "achieved": hbm, "frac": hbm / HBM_PEAK_GBPS,
                # what the byte counters see: traffic between the L2s and the fabric (FETCH_SIZE x 2 + WRITE_SIZE), Infinity-Cache hits
                # included — an upper bound on HBM bytes.  This workload's records (state + snapshot) fit the 256 MiB memory-side cache when
                # cache_resident is true; other_configs.headline_1048576_envs is the same step with the records streaming from HBM proper
                "traffic_level": "L2 <-> fabric (EA) bytes; memory-side-cache hits included: an upper bound on HBM bytes",
                "cache_resident": 2 * footprint < 256 * 2**20,
                "traffic": traffic, "traffic_source": traffic_source, "traffic_measured_in_run": measured_in_run,
                "traffic_fallback": bool(args.measure_traffic and not measured_in_run), "traffic_failure": traffic_failure,
                "hbm_bytes_per_step": moved, "hbm_bytes_kind": "pmc" if traffic is not None else "packed footprint (2 x 320 B per env)",
                "footprint_bytes_per_step": footprint, "traffic_over_footprint": (traffic / footprint) if traffic else None,
                # SURVEY §8(d)'s contract figure: 2024 B per env-step (the reference's 1004-B State read and written + Move[4]) over the
                # same clock.  Bytes the kernel does not move — the device record is packed to 320 B — so not a utilisation (it can
                # pass 1.0); kept because rounds 1-2 reported it as `frac`.
                "contract_achieved": contract, "contract_frac": contract / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": algo_bytes,
                "limited_by": ("vector-ALU issue: ~900 VALU instructions per wavefront-tick x 4 cycles x 4,096 wavefronts over 1,024 SIMDs is "
                               "~6 us of the ~7.5 us a step takes in a long call (671 B moved per env-step: 8 TB/s would be 11.9 G env-steps/s); "
                               "a 20-step region adds ~43 us: launch / synchronise latency and the lag of the slowest of the 4,096 tile chains "
                               "behind the average one (profiles/r05_region_sweep.txt): DESIGN.md §5, SQ counters in profiles/r05_headc_summary.txt"
                               if issue == "chain" else
                               "instruction issue and the slowest wavefront of a launch (one round of 4 wavefronts per SIMD), not HBM: "
                               "DESIGN.md §4, SQ / I-cache counters in profiles/"),
                # <envs per wavefront, lanes per env, fresh boards, fused policy, reset at end, one tick per launch>, as rocprofv3 names it
                # ..., chained launches>
                "kernel": f"pom_step_kernel<{epw}, {lpe}, {'true' if args.fresh_boards else 'false'}, {'true' if args.policy == 'simple' else 'false'}, false, "
                          f"{'true' if (tpl == 1 and lpe == 4) else 'false'}, {'true' if issue == 'chain' and tpl == 1 else 'false'}>",
                "step_ms_hip_events": step_ms, "ms_per_step": ms_per_step,
                # one step = `launches_per_step` launches; per launch: bytes / mean duration (matches rocprofv3's AverageNs).  Chained
                # launches of consecutive ticks overlap on the device: a launch then lasts longer than the step period, and
                # `achieved` above (bytes per step over the period) is the figure that says what the memory system sees
                "launches_per_step": parts,
                # mean launch duration x launches per step / step period: how many launches are in flight on the device on average
                # (1 = launches one after the other; chained launches of consecutive ticks overlap: 2 - 3)
                "launches_in_flight": (launch_ms * parts / ms_per_step) if (launch_ms > 0 and ms_per_step > 0) else None,
                "launch": {"envs": plan["n_envs"] // parts, "hbm_bytes": moved // parts, "algorithmic_bytes": algo_bytes // parts,
                           "packed_bytes": footprint // parts, "ms": launch_ms, "timed_launches": n_launch,
                           "achieved": (moved / parts) / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else None,
                           "contract_achieved": (algo_bytes / parts) / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else None},
            },
        }
        if config3:
            line["config3_simple_agent"] = config3
        if other:
            line["other_configs"] = other
        if not multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(start, args.seed, dist_id, args.max_steps)
        print(json.dumps(line), flush=True)
    env.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None) -> int:
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's shape `python bench.py --gpus N ...`: become the launcher (no torch, no HIP in this process)
        return launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv))
    worker(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
