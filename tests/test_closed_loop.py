"""Closed-loop stepping (pom_batch_step_device_range): Step(State*, Move[4]) with THIS tick's moves from a device policy every tick
(the reference's call shape, src/bboard/environment.cpp:139-149), the batch cut into ranges that each carry
policy(range) -> step(range) -> policy(range) ... on a stream of their own.  Every tick is played by the oracle with the moves the
device policy chose; the stand-in policy (pom_bench_policy) is recomputed on the host from the observation it read."""
import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END, RESET_AT_START, CNT_STEPS, PomError, bench_policy

pytestmark = pytest.mark.gpu


def _same(got, want):
    want = want.copy()
    want["agents"]["pad"] = 0
    return got.tobytes() == want.tobytes()


def _policy_on_host(codes: np.ndarray, first: int, count: int, tick: int) -> np.ndarray:
    """pom_bench_policy_kernel (pom_batch.hip) restated: a workgroup of 256 lanes takes 64 envs, lane l sums w[k] * (2k + 1) over the dwords
    k = l, l + 256, ... of the group's observation bytes (whole dwords only) and hashes the sum with (env, agent, tick)"""
    flat = codes.reshape(-1).view(np.uint8)
    out = np.zeros((count, 4), dtype=np.int32)
    M = 0xFFFFFFFF
    for b in range((count + 63) // 64):
        e0 = first + 64 * b
        envs = min(64, count - 64 * b)
        lo, hi = e0 * 605, (e0 + envs) * 605
        a0, a1 = (lo + 3) & ~3, hi & ~3
        w = flat[a0:a1].view("<u4").astype(np.uint64)
        acc = np.zeros(256, dtype=np.uint64)
        for k0 in range(0, len(w), 256):
            chunk = w[k0:k0 + 256]
            k = np.arange(k0, k0 + len(chunk), dtype=np.uint64)
            acc[:len(chunk)] = (acc[:len(chunk)] + chunk * ((2 * k + 1) & M)) & M
        for tid in range(4 * envs):
            e = e0 + (tid >> 2)
            x = int(acc[tid]) ^ ((e * 0x9E3779B1) & M) ^ (((tid & 3) * 0x85EBCA6B) & M) ^ ((tick * 0xC2B2AE35) & M)
            x ^= x >> 15
            x = (x * 0x2C1B3C6D) & M
            x ^= x >> 12
            out[e - first, tid & 3] = (x >> 8) % 6
    return out


@pytest.mark.parametrize("reset", [RESET_AT_END, RESET_AT_START])
def test_two_ranges_on_two_streams_every_tick_against_the_oracle(hip_lib, oracle, reset):
    import torch
    from tests.test_observe import _oracle as observe_oracle
    n, cap, ticks = 1008, 50, 90
    ranges = [(0, 512), (512, 496)]
    start = pa.make_boards(n, seed=41)
    ref = start.copy()
    status = [dict(done=0, winner=-1, draw=0) for _ in range(n)]
    ob = observe_oracle()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=reset, max_steps=cap) as env:
        env.make_game(start)
        codes, _, _ = env.observe(dtype="codes", attrs=False)
        moves = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        env.sync()
        torch.cuda.synchronize()
        for t in range(ticks):
            before = codes.cpu().numpy().copy()
            for (first, count), st in zip(ranges, streams):
                bench_policy(codes, moves, first, count, t, st)
                env.step_device_range(first, count, moves, st, codes=codes)
            torch.cuda.synchronize()
            mv = moves.cpu().numpy()
            for first, count in ranges:  # the policy is a function of what it read
                assert np.array_equal(mv[first:first + count], _policy_on_host(before, first, count, t)), (t, first)
            for i in range(n):
                if reset == RESET_AT_START and (status[i]["done"] or ref["timeStep"][i] >= cap):
                    ref[i] = start[i]
                    status[i] = dict(done=0, winner=-1, draw=0)
                oracle.env_step(ref[i:i + 1], mv[i], status[i])
                if reset == RESET_AT_END and (status[i]["done"] or ref["timeStep"][i] >= cap):
                    ref[i] = start[i]
                    status[i] = dict(done=0, winner=-1, draw=0)
            got = env.get_state()
            assert _same(got, ref), f"tick {t}"
            assert np.array_equal(codes.cpu().numpy(), ob.observe_codes(got)), f"tick {t}: the fused observation"
        assert env.counters()[CNT_STEPS] == n * ticks


def test_the_loop_can_be_captured_into_a_graph(hip_lib):
    """four ranges, ten ticks, captured once and replayed: the same final state as the same calls issued one by one"""
    import torch
    n, k, reps = 4096, 10, 3
    ranges = [(i * 1024, 1024) for i in range(4)]
    start = pa.make_boards(n, seed=43)
    out = []
    for graph in (False, True):
        with BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=40) as env:
            env.make_game(start)
            codes, _, _ = env.observe(dtype="codes", attrs=False)
            moves = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
            env.sync()
            torch.cuda.synchronize()
            side = [torch.cuda.Stream() for _ in ranges]

            def issue(main):
                for s in side:
                    s.wait_stream(main)
                for t in range(k):
                    for (first, count), s in zip(ranges, side):
                        bench_policy(codes, moves, first, count, t, s)
                        env.step_device_range(first, count, moves, s, codes=codes)
                for s in side:
                    main.wait_stream(s)

            main = torch.cuda.Stream()
            if graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=main):
                    issue(main)
                for _ in range(reps):
                    g.replay()
            else:
                for _ in range(reps):
                    issue(main)
            torch.cuda.synchronize()
            out.append((env.get_state().tobytes(), codes.cpu().numpy().tobytes(), int(env.counters()[CNT_STEPS])))
    assert out[0] == out[1] and out[0][2] == n * k * reps


def test_ranges_are_whole_tiles(hip_lib):
    import torch
    with BatchEnvironment(100, mode=MODE_ENV, auto_reset=True) as env:
        env.make_game(pa.make_boards(100, seed=1))
        mv = torch.zeros((100, 4), dtype=torch.int32, device="cuda")
        with pytest.raises(PomError):
            env.step_device_range(8, 16, mv)
        with pytest.raises(PomError):
            env.step_device_range(0, 24, mv)
        with pytest.raises(PomError):
            env.step_device_range(96, 16, mv)  # past the batch
        env.step_device_range(96, 4, mv)       # the last, partial tile: up to the batch's end
        env.step_device_range(0, 96, mv)
        env.sync()
        assert env.counters()[CNT_STEPS] == 100


def test_a_range_with_the_16_planes_writes_only_its_own_part(hip_lib, oracle):
    """planes instead of codes; the envs outside the range keep what the array held, their states are not stepped"""
    import torch
    from tests.test_observe import _oracle as observe_oracle
    n, first, count = 208, 64, 96
    start = pa.make_boards(n, seed=47)
    ob = observe_oracle()
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=30) as env:
        env.make_game(start)
        rng = np.random.default_rng(3)
        for _ in range(12):  # some play first, so that bombs and flames are on the boards
            env.step(rng.integers(0, 6, size=(n, 4)).astype(np.int32))
        before = env.get_state()
        planes = torch.full((n, 16, 11, 11), 0xEE, dtype=torch.uint8, device="cuda")
        mv = torch.from_numpy(rng.integers(0, 6, size=(n, 4)).astype(np.int32)).cuda()
        env.sync()
        env.step_device_range(first, count, mv, planes=planes)
        env.sync()
        after = env.get_state()
        inside = np.zeros(n, dtype=bool)
        inside[first:first + count] = True
        assert before[~inside].tobytes() == after[~inside].tobytes()
        assert (after["timeStep"][inside] != before["timeStep"][inside]).all()
        want, _, _ = ob.observe(after)
        got = planes.cpu().numpy()
        assert np.array_equal(got[inside], want[inside])
        assert (got[~inside] == 0xEE).all()
