#!/usr/bin/env python3
"""Round-4 GPU soak (not collected by pytest; run by hand on the GPU box): the paths added this round under long random use,
every call checked against the oracle on slices — move tapes of random length mixed with random-move and SimpleAgent calls on one
chained handle, with and without forced give-ups (POM_CHAIN_WAIT_US), both reset modes, and the fused step + observation.
usage: python tests/soak_gpu_r04.py [--calls 400] [--envs 65536]"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END
from tests.oracle_lib import Oracle
from tests.test_gpu_chain import _oracle_explicit, _same

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=400)
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--seed", type=int, default=5)
a = ap.parse_args()
ora = Oracle()
N, cap, m = a.envs, 200, 256
lo = N // 2 - 100  # the slice the oracle follows
t_all = time.time()
for at_end in (False, True):
    rng = np.random.default_rng(a.seed + int(at_end))
    start = pa.make_boards(N, seed=17 + int(at_end))
    ini = np.ascontiguousarray(start[lo:lo + m])
    ref = ini.copy()
    status = [dict(done=0, winner=-1, draw=0) for _ in range(m)]
    mems = np.zeros((m, 4, 16), dtype=np.int32)
    env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=RESET_AT_END if at_end else True, max_steps=cap)
    env.make_game(start)
    tick, ticks_tape, ticks_rand, ticks_simple = 0, 0, 0, 0
    for call in range(a.calls):
        kind = int(rng.integers(0, 4)) if not at_end else int(rng.integers(0, 2))  # (the oracle's run_* helpers reset at the start of a tick)
        if kind <= 1:  # a tape of random length
            k = int(rng.integers(1, 48))
            tape = torch.randint(0, 6, (k, N, 4), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()  # the tape is written (torch's stream) before the handle's stream reads it
            env.step_device_many(tape)  # (the wrapper keeps the tensor alive until the handle has been synchronised)
            host = tape[:, lo:lo + m].cpu().numpy()
            for t in range(k):
                _oracle_explicit(ora, ref, ini, status, host[t], cap, at_end)
            ticks_tape += k
        elif kind == 2:
            k, seed = int(rng.integers(1, 60)), int(rng.integers(1, 1 << 30))
            # the run_* helpers keep no status across calls: bring the slice to the same footing (a finished env restarts at the next tick)
            env.set_tick(tick)
            env.step_random(seed, 1, ticks=k)
            ora.run_random(ref, ini, k, seed, lo, tick, 1, cap)
            status = [dict(done=int(ref["aliveAgents"][i] <= 1), winner=-1, draw=0) for i in range(m)]
            tick += k
            ticks_rand += k
        else:
            k, seed = int(rng.integers(1, 30)), int(rng.integers(1, 1 << 30))
            env.set_tick(tick)
            env.step_simple(seed, k)
            ora.run_simple(ref, ini, mems, k, seed, lo, tick, cap)
            status = [dict(done=int(ref["aliveAgents"][i] <= 1), winner=-1, draw=0) for i in range(m)]
            tick += k
            ticks_simple += k
        if call % 4 == 3 or call == a.calls - 1:
            got = env.get_state(lo, m)
            assert _same(got, ref), f"reset at end {at_end}: call {call} (kind {kind}) differs from the oracle"
    st = env.chain_stats()
    print(f"{'reset at the END of the tick' if at_end else 'reset at the start of the next tick'}: {N} envs, {a.calls} calls "
          f"({ticks_tape} tape ticks, {ticks_rand} random-move ticks, {ticks_simple} SimpleAgent ticks), slice of {m} envs = oracle; "
          f"chained launches {st['launches']}, settles {st['settles']}, tiles left behind {st['tiles_recovered']}, ticks replayed {st['ticks_replayed']}",
          flush=True)
    env.close()
# the fused step + observation under play
spec = importlib.util.spec_from_file_location("pom_observe_oracle", os.path.join(ROOT, "oracle", "pom_observe_oracle.py"))
ob = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ob)
env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=cap)
env.make_game(pa.make_boards(N, seed=3, kind="stress"))
for t in range(300):
    mv = torch.randint(0, 6, (N, 4), dtype=torch.int32, device="cuda")
    planes, attrs, eattrs = env.step_device_observe(mv)
    if t % 25 == 24:
        s = env.get_state(lo, m)
        want, want_attrs, _ = ob.observe(s)
        assert np.array_equal(planes[lo:lo + m].cpu().numpy(), want) and np.array_equal(attrs[lo:lo + m].cpu().numpy(), want_attrs), t
        codes, _, _ = env.observe(dtype="codes", attrs=False)  # the compact layout of the same state, the whole batch against the 16 planes
        pl = planes.to(torch.int64)
        board = torch.full_like(pl[:, 0], 255)
        for plane, value in ((0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (5, 6), (6, 7), (7, 8), (8, 10), (9, 11), (10, 12), (11, 13)):
            board = torch.where(pl[:, plane] == 1, torch.full_like(board, value), board)
        assert torch.equal(codes[:, 0].to(torch.int64), board) and torch.equal(codes[:, 1:5], planes[:, 12:16]), t
        assert np.array_equal(codes[lo:lo + m].cpu().numpy(), ob.observe_codes(s)), t
env.close()
print(f"fused step + observation: {N} envs x 300 ticks, slices = numpy restatement every 25 ticks; code planes = the 16 planes, all envs")
print(f"soak r04 ok in {time.time() - t_all:.0f} s (POM_CHAIN_WAIT_US={os.environ.get('POM_CHAIN_WAIT_US', 'default')})")
