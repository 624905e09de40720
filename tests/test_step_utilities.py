"""The reference's `[step utilities]` cases (unit_test/bboard/step_utility_test.cpp:38-173) against the oracle's exported helpers
and — second half of this file — against the DEVICE tick body's own preparation functions (PomStepper::prep_positions /
prep_dependencies of pomcpp_amd/csrc/pom_step_body.h, host build of tests/emul; on the GPU they are pinned through Step)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pomcpp_amd.state as S
from pomcpp_amd.state import Move


def _dest(oracle, s, m):
    out = np.zeros(8, dtype=np.int32)
    mv = np.asarray(m, dtype=np.int32)
    oracle.lib.pom_oracle_dest_pos(s.ctypes.data, mv.ctypes.data, out.ctypes.data)
    return out


def _resolve(oracle, s, des):
    dep = np.full(4, -1, dtype=np.int32)
    chain = np.full(4, -1, dtype=np.int32)
    n = oracle.lib.pom_oracle_resolve_dependencies(s.ctypes.data, des.ctypes.data, dep.ctypes.data, chain.ctypes.data)
    return n, dep, chain


def _line(s):
    for i in range(4):
        S.put_agent(s[0], i, 0, i)


def test_destination_position_filling(oracle):  # :38-61
    s = S.new_states(1)
    _line(s)
    d = _dest(oracle, s, [Move.DOWN, Move.LEFT, Move.RIGHT, Move.UP])
    assert d.reshape(4, 2).tolist() == [[0, 1], [0, 0], [3, 0], [3, -1]]


def test_fix_switch_position(oracle):  # :63-84
    s = S.new_states(1)
    _line(s)
    d = _dest(oracle, s, [Move.RIGHT, Move.RIGHT, Move.LEFT, Move.LEFT])
    oracle.lib.pom_oracle_fix_switch_move(s.ctypes.data, d.ctypes.data)
    assert d.reshape(4, 2).tolist() == [[1, 0], [1, 0], [2, 0], [2, 0]]


def _place(s, pos):
    for i, (x, y) in enumerate(pos):
        S.put_agent(s[0], x, y, i)


def test_resolve_0_to_1(oracle):  # :97-109
    s = S.new_states(1)
    _place(s, [(0, 0), (1, 0), (8, 4), (9, 8)])
    _, _, chain = _resolve(oracle, s, _dest(oracle, s, [Move.RIGHT] * 3 + [Move.IDLE]))
    assert 1 in chain.tolist()


def test_resolve_two_chains(oracle):  # :110-122
    s = S.new_states(1)
    _place(s, [(0, 0), (1, 0), (8, 8), (9, 8)])
    _, _, chain = _resolve(oracle, s, _dest(oracle, s, [Move.RIGHT] * 3 + [Move.IDLE]))
    assert 1 in chain.tolist() and 3 in chain.tolist()


def test_resolve_complete_chain(oracle):  # :123-135
    s = S.new_states(1)
    _line(s)
    _, _, chain = _resolve(oracle, s, _dest(oracle, s, [Move.RIGHT] * 4))
    assert 3 in chain.tolist()


def test_resolve_ouroboros(oracle):  # :136-153
    s = S.new_states(1)
    _place(s, [(0, 0), (1, 0), (1, 1), (0, 1)])
    n, _, chain = _resolve(oracle, s, _dest(oracle, s, [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP]))
    assert n == 0 and chain[0] == -1


def test_dead_agents_are_roots(oracle):  # :154-172
    s = S.new_states(1)
    _place(s, [(0, 0), (1, 0), (1, 1), (0, 1)])
    S.kill(s[0], 1)
    _, _, chain = _resolve(oracle, s, _dest(oracle, s, [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP]))
    assert 0 in chain.tolist() and 1 in chain.tolist()


# ---- the same vectors through the device tick body's preparation functions (tests/emul/pom_emul.cpp: pom_emul_prep) -------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def prep():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    so = os.path.join(ROOT, "build", "libpom_emul_prep.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "pomcpp_amd/csrc"), "tests/emul/pom_emul.cpp", "-o", so], check=True, cwd=ROOT)
    lib = C.CDLL(so)
    lib.pom_emul_prep.argtypes = [C.c_void_p] * 6
    lib.pom_emul_prep.restype = C.c_int

    def run(s, m):
        dest, dep, roots = np.zeros(8, dtype=np.int32), np.zeros(4, dtype=np.int32), np.zeros(4, dtype=np.int32)
        contact = C.c_int32(0)
        mv = np.asarray(m, dtype=np.int32)
        n = lib.pom_emul_prep(s.ctypes.data, mv.ctypes.data, dest.ctypes.data, dep.ctypes.data, roots.ctypes.data, C.byref(contact))
        assert n >= 0
        return n, dest.reshape(4, 2).tolist(), dep.tolist(), roots.tolist(), contact.value
    return run


def _oracle_prep(oracle, s, m):
    """FillDestPos -> FixSwitchMove -> ResolveDependencies as step.cpp:17-33 chains them"""
    d = _dest(oracle, s, m)
    oracle.lib.pom_oracle_fix_switch_move(s.ctypes.data, d.ctypes.data)
    n, dep, chain = _resolve(oracle, s, d)
    return n, d.reshape(4, 2).tolist(), dep.tolist(), chain.tolist()


def test_device_prep_destination_position_filling(prep):  # :38-61
    s = S.new_states(1)
    _line(s)
    _, dest, _, _, _ = prep(s, [Move.DOWN, Move.LEFT, Move.RIGHT, Move.UP])
    assert dest == [[0, 1], [0, 0], [3, 0], [3, -1]]


def test_device_prep_fix_switch_position(prep):  # :63-84
    s = S.new_states(1)
    _line(s)
    _, dest, _, _, contact = prep(s, [Move.RIGHT, Move.RIGHT, Move.LEFT, Move.LEFT])
    assert contact & 1 and dest == [[1, 0], [1, 0], [2, 0], [2, 0]]


@pytest.mark.parametrize("pos,moves,kill,expect_roots,n_roots", [
    ([(0, 0), (1, 0), (8, 4), (9, 8)], [Move.RIGHT] * 3 + [Move.IDLE], (), {1}, None),                      # :97-109  0 -> 1
    ([(0, 0), (1, 0), (8, 8), (9, 8)], [Move.RIGHT] * 3 + [Move.IDLE], (), {1, 3}, None),                   # :110-122 two chains
    ([(0, 0), (1, 0), (2, 0), (3, 0)], [Move.RIGHT] * 4, (), {3}, None),                                    # :123-135 complete chain
    ([(0, 0), (1, 0), (1, 1), (0, 1)], [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP], (), set(), 0),          # :136-153 ouroboros
    ([(0, 0), (1, 0), (1, 1), (0, 1)], [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP], (1,), {0, 1}, None),    # :154-172 dead agents are roots
])
def test_device_prep_resolve_dependencies(prep, oracle, pos, moves, kill, expect_roots, n_roots):
    s = S.new_states(1)
    _place(s, pos)
    if kill:
        S.kill(s[0], *kill)
    n, dest, dep, roots, _ = prep(s, moves)
    assert expect_roots <= {r for r in roots if r >= 0}
    if n_roots is not None:
        assert n == n_roots and roots[0] == -1
    # and all of it equals the restatement of the reference's helpers, which tests/golden pins against the compiled reference
    on, odest, odep, ochain = _oracle_prep(oracle, s, moves)
    assert (n, dest, dep, roots[:n]) == (on, odest, odep, ochain[:on])


def test_device_prep_equals_the_restated_helpers_on_random_positions(prep, oracle):
    """agents crowded into a 3 x 3 corner so that switches, chains, cycles and dead agents in the way all occur"""
    rng = np.random.default_rng(5)
    seen_contact = 0
    for _ in range(3000):
        s = S.new_states(1)
        _place(s, [tuple(int(v) for v in rng.integers(0, 3, size=2)) for _ in range(4)])
        dead = [i for i in range(4) if rng.random() < 0.2]
        if dead:
            S.kill(s[0], *dead)
        m = [int(v) for v in rng.integers(0, 6, size=4)]
        n, dest, dep, roots, contact = prep(s, m)
        on, odest, odep, ochain = _oracle_prep(oracle, s, m)
        seen_contact += contact & 1
        if contact & 1:
            assert (n, dest, dep, roots[:n]) == (on, odest, odep, ochain[:on]), (s["agents"][0], m)
        else:  # nobody's destination is anybody's cell: nothing to fix, everybody a root in index order
            raw = _dest(oracle, s, m).reshape(4, 2).tolist()
            assert dest == raw == odest and n == on == 4 and roots == [0, 1, 2, 3] == ochain and dep == odep == [-1] * 4
    assert seen_contact > 1000


@pytest.mark.gpu
def test_gpu_step_on_the_step_utility_vectors_and_crowded_positions(hip_lib, oracle):
    """The same vectors through the GPU: the reference's `[step utilities]` positions and moves (step_utility_test.cpp:38-173) and
    6,000 crowded random positions (3 x 3 corner: switches, chains, cycles, dead agents in the way, two agents on one cell), each
    stepped three ticks by the device tick (POM_MODE_RAW = bare bboard::Step) and by the oracle: every state and every UB flag."""
    from pomcpp_amd.batch import BatchEnvironment, MODE_RAW
    cases = [
        ([(0, 0), (1, 0), (2, 0), (3, 0)], [Move.DOWN, Move.LEFT, Move.RIGHT, Move.UP], ()),
        ([(0, 0), (1, 0), (2, 0), (3, 0)], [Move.RIGHT, Move.RIGHT, Move.LEFT, Move.LEFT], ()),
        ([(0, 0), (1, 0), (8, 4), (9, 8)], [Move.RIGHT] * 3 + [Move.IDLE], ()),
        ([(0, 0), (1, 0), (8, 8), (9, 8)], [Move.RIGHT] * 3 + [Move.IDLE], ()),
        ([(0, 0), (1, 0), (2, 0), (3, 0)], [Move.RIGHT] * 4, ()),
        ([(0, 0), (1, 0), (1, 1), (0, 1)], [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP], ()),
        ([(0, 0), (1, 0), (1, 1), (0, 1)], [Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP], (1,)),
    ]
    rng = np.random.default_rng(6)
    for _ in range(6000):
        pos = [tuple(int(v) for v in rng.integers(0, 3, size=2)) for _ in range(4)]
        cases.append((pos, [int(v) for v in rng.integers(0, 6, size=4)], tuple(i for i in range(4) if rng.random() < 0.2)))
    n = len(cases)
    states = S.new_states(n)
    moves = np.zeros((3, n, 4), dtype=np.int32)
    for k, (pos, mv, dead) in enumerate(cases):
        for i, (x, y) in enumerate(pos):
            S.put_agent(states[k], x, y, i)
        if dead:
            S.kill(states[k], *dead)
        moves[0, k] = mv
    moves[1:] = rng.integers(0, 6, size=(2, n, 4))
    ref = states.copy()
    seen = np.zeros(n, dtype=np.uint32)  # the record's flags are sticky: everything raised since the upload
    with BatchEnvironment(n, mode=MODE_RAW) as env:
        env.make_game(states)
        for t in range(3):
            env.step(moves[t])
            seen |= oracle.step_batch(ref, moves[t])
            got = env.get_state()
            want = ref.copy()
            want["agents"]["pad"] = 0
            assert got.tobytes() == want.tobytes(), t
            assert np.array_equal(env.status()["ubflags"], seen), t
    assert (seen & 1).sum() > 20  # lost agents (SURVEY Q-UB1) do occur in crowded corners
