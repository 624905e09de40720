#!/usr/bin/env python3
"""Parity + speed of the quad kernel (POM_EPW=16 POM_QUAD=1) vs the default, single stream."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
def run(n, kind, dist, env_vars, ticks=200, streams=1):
    for k in ("POM_EPW", "POM_QUAD"):
        os.environ.pop(k, None)
    os.environ.update(env_vars)
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams)
    env.make_game(pa.make_boards(n, seed=3, kind=kind))
    env.step_random(7, dist, ticks=60)
    env.sync()
    t0 = time.perf_counter()
    env.step_random(7, dist, ticks=ticks)
    env.sync()
    dt = time.perf_counter() - t0
    d = hashlib.blake2b(env.get_state().tobytes(), digest_size=8).hexdigest()
    c = env.counters().tolist()
    env.close()
    return dt / ticks * 1e6, d, c
for n, kind, dist in ((65536, "ffa", 1), (65536, "stress", 2), (4096, "ffa", 1), (262144, "ffa", 1)):
    ref = None
    for name, ev in (("EPW32", {"POM_EPW": "32"}), ("EPW16", {"POM_EPW": "16"}), ("QUAD", {"POM_EPW": "16", "POM_QUAD": "1"})):
        for streams in (1, 2):
            us, d, c = run(n, kind, dist, ev, streams=streams)
            ref = ref or (d, c)
            print(f"envs {n:7d} {kind:6s} {name:6s} streams {streams}: {us:8.2f} us/tick {n/us/1e3:7.3f} G/s  {'same' if (d, c) == ref else 'DIFFERENT ' + d + str(c)}")
