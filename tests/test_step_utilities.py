"""The reference's `[step utilities]` cases (unit_test/bboard/step_utility_test.cpp:38-173) against the
oracle's exported helpers (the device path fuses them into the tick; there they are pinned through Step)."""
import numpy as np

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
