"""start boards generated on the device (include/pom_boardgen.h).  CPU: the oracle restatement keeps the
reference's InitState distribution (bboard.cpp:59-74,322-333,346-382) and the device body (host build) equals it.  GPU:
pom_batch_generate and fresh boards on auto-reset through the C-ABI, bit-exact against the oracle under random and SimpleAgent
play, every kernel shape."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from pomcpp_amd.state import Item, STATE_DTYPE, is_wood

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I" + os.path.join(ROOT, p) for p in ("include", "pomcpp_amd/csrc", "oracle")]
CORNERS = [0, 10, 120, 110]


def test_oracle_boards_keep_the_reference_distribution(oracle):
    n = 6000
    s = oracle.boardgen(77, np.arange(n), np.arange(n) % 5)
    b = s["board"].reshape(n, 121).astype(np.int64)
    inner = np.ones(121, dtype=bool)
    inner[CORNERS] = False
    cells = b[:, inner]
    m = cells.size
    for frac, want in (((cells == Item.RIGID).mean(), 1 / 7), (is_wood(cells).mean(), 1 / 7), ((cells == 0).mean(), 5 / 7)):
        assert abs(frac - want) < 4 * np.sqrt(want * (1 - want) / m)          # 4 sigma
    # ceil(woods / 2) woods carry a flag; the corners may hide up to 4 woods of either kind
    wood = is_wood(b)
    flagged = wood & ((b & 0xFF) != 0)
    assert np.all(np.abs(flagged.sum(1) - (wood.sum(1) + 1) // 2) <= 4)
    flags = (b[flagged] & 0xFF)
    assert set(np.unique(flags)) == {1, 2, 3, 4}
    counts = np.bincount(flags, minlength=5)[1:]
    assert np.all(np.abs(counts / counts.sum() - 0.25) < 0.01)
    # which woods are flagged does not depend on their position (selection sampling is uniform over subsets)
    first_half = flagged[:, :60].sum() / max(wood[:, :60].sum(), 1)
    second_half = flagged[:, 61:].sum() / max(wood[:, 61:].sum(), 1)
    assert abs(first_half - second_half) < 0.02
    # neighbouring cells are independent: P(wood | left neighbour wood) == P(wood)
    grid = is_wood(b.reshape(n, 11, 11)[:, 1:10, 1:10])
    both = (grid[:, :, 1:] & grid[:, :, :-1]).mean()
    assert abs(both - (1 / 7) ** 2) < 0.002
    # PutAgentsInCorners(0, 1, 2, 3) and a fresh State around the board
    assert np.all(b[:, 0] == Item.AGENT0) and np.all(b[:, 10] == Item.AGENT0 + 1)
    assert np.all(b[:, 120] == Item.AGENT0 + 2) and np.all(b[:, 110] == Item.AGENT0 + 3)
    assert s["agents"]["x"][0].tolist() == [0, 10, 10, 0] and s["agents"]["y"][0].tolist() == [0, 0, 10, 10]
    assert np.all(s["aliveAgents"] == 4) and np.all(s["timeStep"] == 0) and np.all(s["flames_queue"]["timeLeft"] == 4)
    assert np.all(s["agents"]["maxBombCount"] == 1) and np.all(s["agents"]["bombStrength"] == 1)
    assert np.all(s["bombs_count"] == 0) and np.all(s["flames_count"] == 0) and not s["agents"]["dead"].any()


# ---- the pin against the compiled reference: tests/golden/boardgen_stats.npz (tests/golden/gen_boardgen_stats.py) ----------------
REFSTATS = np.load(os.path.join(ROOT, "tests", "golden", "boardgen_stats.npz"))
INNER = np.ones(121, dtype=bool)
INNER[CORNERS] = False


def _chi2_two_sample(a, b, min_expected=25):
    """Pearson statistic and degrees of freedom for 'two histograms, one distribution' (small bins merged from both ends)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    keep = (a + b) > 0
    a, b = a[keep], b[keep]
    while len(a) > 2 and min(a[0] + b[0], a[-1] + b[-1]) < 2 * min_expected:  # merge the tails inwards
        if a[0] + b[0] <= a[-1] + b[-1]:
            a, b = np.concatenate([[a[0] + a[1]], a[2:]]), np.concatenate([[b[0] + b[1]], b[2:]])
        else:
            a, b = np.concatenate([a[:-2], [a[-2] + a[-1]]]), np.concatenate([b[:-2], [b[-2] + b[-1]]])
    na, nb = a.sum(), b.sum()
    exp_a, exp_b = (a + b) * na / (na + nb), (a + b) * nb / (na + nb)
    return float(((a - exp_a) ** 2 / exp_a + (b - exp_b) ** 2 / exp_b).sum()), len(a) - 1


def check_against_reference_distribution(boards):
    """boards: int64 [n, 121] start boards of a generator (agents in the corners).  Compared with what the compiled reference's
    InitBoardItems produced over 20,000 seeds — on the 117 cells the agents do not cover.  The reference's own figures carry
    ~0.1 % of dirt: its stray read past the wood list (bboard.cpp:367-372) turns a passage into 1..4 on some boards, and 14 % of
    its runs fault inside the flag pass (their cell kinds are kept, their flags are not)."""
    from scipy.stats import chi2
    n = len(boards)
    cells = boards[:, INNER]
    wood = (cells >> 8) == 2
    nb = int(REFSTATS["n_boards"])
    inner2d = INNER.reshape(11, 11)
    for mine, ref_map in (((cells == Item.RIGID).mean(), REFSTATS["rigid_per_cell"]), (wood.mean(), REFSTATS["wood_per_cell"])):
        ref = ref_map[inner2d].sum() / (nb * 117)
        assert abs(mine - ref) < 0.0025, (mine, ref)  # binomial sd 0.0002-0.0003 on either side + the reference's stray writes
        assert abs(ref - 1 / 7) < 0.0025              # ... which is also how far the reference itself is from its nominal 1/7
    # woods per board: the whole histogram
    stat, dof = _chi2_two_sample(np.bincount(wood.sum(1), minlength=118), REFSTATS["woods_inner_hist"])
    assert chi2.sf(stat, dof) > 1e-4, (stat, dof)
    # neighbouring cells: the 3 x 3 table of (kind, kind of the right-hand neighbour), columns 1..9
    k = np.where(boards == Item.RIGID, 1, np.where((boards >> 8) == 2, 2, 0)).reshape(n, 11, 11)
    pairs = np.array([[((k[:, :, 1:9] == a) & (k[:, :, 2:10] == b)).sum() for b in range(3)] for a in range(3)])
    stat, dof = _chi2_two_sample(pairs.reshape(-1), REFSTATS["adjacent_pairs_inner"].reshape(-1))
    assert chi2.sf(stat, dof) > 1e-4, (stat, dof)
    # flags: ceil(woods / 2) per board — what the reference gives on every run its stray read does not spoil
    assert int(REFSTATS["flags_given_ok"]) == int(REFSTATS["flags_wanted_ok"])
    assert REFSTATS["shortfall_hist"][2:].sum() == 0  # and at most one short otherwise
    flag = np.where((boards >> 8) == 2, boards & 0xFF, 0)
    hidden = 4  # a corner may hide a wood, flagged or not
    assert np.all(np.abs((flag > 0).sum(1) - (((boards >> 8) == 2).sum(1) + 1) // 2) <= hidden)
    # flag values: uniform over 1..4 in both
    stat, dof = _chi2_two_sample(np.bincount(flag[flag > 0], minlength=5)[1:], REFSTATS["flag_value_totals"])
    assert chi2.sf(stat, dof) > 1e-4, (stat, dof)
    # which woods are flagged does not depend on where they are: per board row, flagged / woods as in the reference
    f2, w2 = (flag > 0).reshape(n, 11, 11), ((boards >> 8) == 2).reshape(n, 11, 11)
    ref_rows = REFSTATS["flagged_per_cell_ok"][:, 1:10].sum(1) / REFSTATS["wood_per_cell_ok"][:, 1:10].sum(1)
    my_rows = f2[:, :, 1:10].sum((0, 2)) / w2[:, :, 1:10].sum((0, 2))
    assert np.all(np.abs(my_rows - ref_rows) < 0.02), (my_rows, ref_rows)


def test_oracle_boards_match_the_compiled_references_distribution(oracle):
    n = 20000
    s = oracle.boardgen(20261004, np.arange(n), np.arange(n) % 3)
    check_against_reference_distribution(s["board"].reshape(n, 121).astype(np.int64))


def test_oracle_boards_are_a_function_of_seed_env_episode(oracle):
    a = oracle.boardgen(5, [3, 3, 4, 3], [0, 1, 0, 0])
    assert a[0].tobytes() == a[3].tobytes()
    assert a[0].tobytes() != a[1].tobytes() and a[0].tobytes() != a[2].tobytes()
    assert oracle.boardgen(6, [3], [0])[0].tobytes() != a[0].tobytes()
    assert oracle.boardgen(5 + (1 << 32), [3], [0])[0].tobytes() != a[0].tobytes()   # the high half of the seed counts


def test_exact_number_of_flags_before_the_corners(oracle):
    """count through the spec's own draws: exactly ceil(woods/2) cells are chosen, whatever the board"""
    for env in range(300):
        s = oracle.boardgen(1234, [env], [env % 7])[0]
        b = s["board"].reshape(121).astype(np.int64).copy()
        # recover what the corners overwrote from the spec (python restatement of the three draw kinds)
        def fmix(h):
            h &= 0xFFFFFFFF; h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF; h ^= h >> 16
            return h
        seed = 1234
        k = fmix(((env % 7) * 0x7FEB352D + (seed >> 32) + 0x5BD1E995) & 0xFFFFFFFF)
        key = fmix((seed & 0xFFFFFFFF) ^ ((env * 0x9E3779B1) & 0xFFFFFFFF) ^ k)
        draw = lambda i: fmix((key + (i + 1) * 0x9E3779B9) & 0xFFFFFFFF)
        kinds = [(draw(c) * 7) >> 32 for c in range(121)]
        woods = [c for c in range(121) if kinds[c] == 2]
        need, chosen = (len(woods) + 1) // 2, {}
        for j, c in enumerate(woods):
            if ((draw(128 + c) * (len(woods) - j)) >> 32) < need:
                chosen[c] = 1 + (draw(256 + c) >> 30)
                need -= 1
        assert need == 0 and len(chosen) == (len(woods) + 1) // 2
        for c in range(121):
            if c in CORNERS:
                continue
            want = Item.RIGID if kinds[c] == 1 else (Item.WOOD + chosen.get(c, 0)) if kinds[c] == 2 else 0
            assert b[c] == want, (env, c)


def test_device_generator_body_matches_oracle(oracle):
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    so = os.path.join(ROOT, "build", "libpom_boardgen_emul.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-fPIC", "-shared", *INC,
                    "tests/emul/pom_boardgen_emul.cpp", "-o", so], check=True, cwd=ROOT)
    lib = C.CDLL(so)
    lib.pom_emul_boardgen.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pom_emul_boardgen.restype = None
    rng = np.random.default_rng(8)
    envs = rng.integers(0, 1 << 31, size=4000)
    eps = rng.integers(0, 5000, size=4000)
    for seed in (0, 9, (7 << 40) + 3):
        want = oracle.boardgen(seed, envs, eps)
        got = np.zeros(envs.size, dtype=STATE_DTYPE)
        for i in range(envs.size):
            lib.pom_emul_boardgen(seed, int(envs[i]), int(eps[i]), got.ctypes.data + i * 1004)
        assert got.tobytes() == want.tobytes()


# ---- GPU, through the C-ABI --------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_boards_match_the_compiled_references_distribution(hip_lib):
    """pom_batch_generate against the histograms recorded from the compiled reference's InitBoardItems (20,000 seeds)"""
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n = 20000
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, env_offset=31337)
    env.generate(20261004)
    boards = env.get_state()["board"].reshape(n, 121).astype(np.int64)
    env.close()
    check_against_reference_distribution(boards)


@pytest.mark.gpu
@pytest.mark.parametrize("n,offset", [(5000, 0), (100, 123456), (17, 7)])
def test_generate_matches_oracle(hip_lib, oracle, n, offset):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, env_offset=offset)
    env.generate(4242)
    want = oracle.boardgen(4242, offset + np.arange(n), np.zeros(n))
    assert env.get_state().tobytes() == want.tobytes()
    assert not env.episodes().any()
    st = env.status()
    assert not st["done"].any() and np.all(st["alive"] == 4)
    env.close()


SHAPES = [dict(), dict(envs_per_wave=16, lanes_per_env=1), dict(envs_per_wave=32), dict(envs_per_wave=64), dict(streams=3)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES, ids=lambda d: "-".join(f"{k}{v}" for k, v in d.items()) or "default")
@pytest.mark.parametrize("tpl", [1, 4])
def test_fresh_boards_under_random_play(hip_lib, oracle, shape, tpl):
    """short games (40-tick cap) so that every env restarts several times; fresh board per restart, drawn inside the tick"""
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, offset, seed, bseed, cap = 1500, 1000, 21, 99, 40
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=cap, env_offset=offset, fresh_boards=True, board_seed=bseed, **shape)
    env.generate(bseed)
    ref = oracle.boardgen(bseed, offset + np.arange(n), np.zeros(n))
    eps = np.zeros(n, dtype=np.int32)
    tick = 0
    for chunk in (4, 48, 100, 60):
        env.step_random(seed, 1, ticks=chunk, ticks_per_launch=tpl)
        oracle.run_random_fresh(ref, eps, chunk, seed, bseed, offset, tick, 1, cap)
        tick += chunk
        assert env.get_state().tobytes() == ref.tobytes(), (shape, tpl, tick)
    # one more tick so that the envs that just finished have restarted: then episodes agree exactly
    assert eps.min() >= 4
    got_eps = env.episodes()
    done = env.status()["done"].astype(bool)
    assert np.array_equal(got_eps, eps)                 # the oracle counts a restart when it happens, as the device does
    assert env.counters()[2] == eps.sum()               # POM_CNT_RESETS
    assert done.sum() == ((ref["aliveAgents"] <= 1) | (ref["timeStep"] >= cap)).sum()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [1, 2])
def test_fresh_boards_under_simple_agents(hip_lib, oracle, streams):
    """the policy judges a restarting env on the board the tick is about to draw for it"""
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, offset, seed, bseed, cap = 1200, 50, 5, 1717, 60
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=cap, env_offset=offset, fresh_boards=True, board_seed=bseed,
                           streams=streams)
    env.generate(bseed)
    ref = oracle.boardgen(bseed, offset + np.arange(n), np.zeros(n))
    eps = np.zeros(n, dtype=np.int32)
    mems = np.zeros((n, 4, 16), dtype=np.int32)
    tick = 0
    for chunk in (10, 70, 130):
        env.step_simple(seed, chunk)
        oracle.run_simple_fresh(ref, eps, mems, chunk, seed, bseed, offset, tick, cap)
        tick += chunk
        assert env.get_state().tobytes() == ref.tobytes(), tick
        assert np.array_equal(env.policy_memory(), mems), tick
    assert eps.min() >= 2 and np.array_equal(env.episodes(), eps)
    env.close()


@pytest.mark.gpu
def test_uploaded_first_game_then_fresh_boards(hip_lib, oracle):
    import pomcpp_amd as pa
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, seed, bseed, cap = 700, 3, 8, 30
    start = pa.make_boards(n, seed=12)
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=cap, fresh_boards=True, board_seed=bseed)
    env.make_game(start)
    ref, eps = start.copy(), np.zeros(n, dtype=np.int32)
    env.step_random(seed, 1, ticks=95)
    oracle.run_random_fresh(ref, eps, 95, seed, bseed, 0, 0, 1, cap)
    assert env.get_state().tobytes() == ref.tobytes() and np.array_equal(env.episodes(), eps) and eps.min() >= 3
    env.make_game(start[:100], first=50)                # an upload starts episode 0 of those envs again
    assert not env.episodes(50, 100).any() and env.episodes(0, 50).all()
    env.close()


@pytest.mark.gpu
def test_snapshot_replay_is_untouched_without_the_option(hip_lib, oracle):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, cap = 300, 25
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=cap)
    env.generate(31)
    first = env.get_state()
    ref = first.copy()
    env.step_random(2, 1, ticks=60)
    oracle.run_random(ref, first, 60, 2, 0, 0, 1, cap)
    assert env.get_state().tobytes() == ref.tobytes() and not env.episodes().any()
    env.close()
