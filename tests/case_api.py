"""Backends for the case scripts of tests/step_cases.py (test infrastructure).

HostAPI  — boards are set up with the product's host helpers (pomcpp_amd.state), SpawnFlame (a
           State method the reference's tests call directly) with the oracle, and stepped by the
           stepper under test: the oracle on CPU, or the HIP path through the C-ABI (pom_step).
RefAPI   — everything through the compiled, unmodified reference (oracle/_ref/libpomref.so);
           only tests/golden/gen_golden.py uses it, in the build container.
Both record the trace [(state before, Move[4], state after)] of every Step.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import pomcpp_amd.state as S
from pomcpp_amd.state import STATE_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Base:
    def __init__(self):
        self.trace = []
        self.checks = 0

    def require(self, cond):
        self.checks += 1
        assert bool(cond)

    def require_agent(self, s, agent, x, y):  # REQUIRE_AGENT, board_logic.cpp:11-17
        self.checks += 3
        assert int(s["agents"][0, agent]["x"]) == x and int(s["agents"][0, agent]["y"]) == y, \
            (agent, int(s["agents"][0, agent]["x"]), int(s["agents"][0, agent]["y"]), x, y)
        assert int(s["board"][0, y, x]) == S.Item.AGENT0 + agent

    def several_steps(self, times, s, m):  # SeveralSteps, board_logic.cpp:22-28
        for _ in range(times):
            self.step(s, m)

    def _record(self, before, m, s):
        after = s.copy()
        after["agents"]["pad"] = 0
        before["agents"]["pad"] = 0
        self.trace.append((before.tobytes(), tuple(int(v) for v in m), after.tobytes()))


class HostAPI(_Base):
    def __init__(self, stepper, oracle):
        super().__init__()
        self._stepper = stepper
        self._oracle = oracle

    def make(self):
        return S.new_states(1)

    def corners(self, s, a0, a1, a2, a3):
        S.put_agents_in_corners(s[0], a0, a1, a2, a3)

    def put_agent(self, s, x, y, agent):
        S.put_agent(s[0], x, y, agent)

    def put_item(self, s, x, y, item):
        S.put_item(s[0], x, y, item)

    def kill(self, s, *agents):
        S.kill(s[0], *agents)

    def plant_bomb(self, s, x, y, agent, set_item=False, life_time=S.BOMB_LIFETIME):
        S.plant_bomb(s[0], x, y, agent, set_item, life_time)

    def set_bomb_direction(self, s, offset, direction):
        S.set_bomb_direction(s[0], offset, direction)

    def spawn_flame(self, s, x, y, strength):
        self._oracle.spawn_flame(s, x, y, strength)

    def step(self, s, m):
        before = s.copy()
        self._stepper(s, np.asarray(m, dtype=np.int32))
        self._record(before, m, s)


class RefAPI(_Base):
    def __init__(self):
        super().__init__()
        lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpomref.so"))
        VP, I = C.c_void_p, C.c_int
        lib.ref_init_state.argtypes = [VP]
        lib.ref_step.argtypes = [VP, VP]
        lib.ref_put_agent.argtypes = [VP, I, I, I]
        lib.ref_put_agents_in_corners.argtypes = [VP, I, I, I, I]
        lib.ref_kill.argtypes = [VP, I]
        lib.ref_put_item.argtypes = [VP, I, I, I]
        lib.ref_plant_bomb.argtypes = [VP, I, I, I, I, I]
        lib.ref_spawn_flame.argtypes = [VP, I, I, I]
        lib.ref_set_bomb_direction.argtypes = [VP, I, I]
        assert lib.ref_state_size() == STATE_DTYPE.itemsize
        self.lib = lib

    def make(self):
        s = np.zeros(1, dtype=STATE_DTYPE)
        self.lib.ref_init_state(s.ctypes.data)
        return s

    def corners(self, s, a0, a1, a2, a3):
        self.lib.ref_put_agents_in_corners(s.ctypes.data, a0, a1, a2, a3)

    def put_agent(self, s, x, y, agent):
        self.lib.ref_put_agent(s.ctypes.data, x, y, agent)

    def put_item(self, s, x, y, item):
        self.lib.ref_put_item(s.ctypes.data, x, y, item)

    def kill(self, s, *agents):
        for a in agents:
            self.lib.ref_kill(s.ctypes.data, a)

    def plant_bomb(self, s, x, y, agent, set_item=False, life_time=S.BOMB_LIFETIME):
        self.lib.ref_plant_bomb(s.ctypes.data, x, y, agent, life_time, int(set_item))

    def set_bomb_direction(self, s, offset, direction):
        self.lib.ref_set_bomb_direction(s.ctypes.data, offset, direction)

    def spawn_flame(self, s, x, y, strength):
        self.lib.ref_spawn_flame(s.ctypes.data, x, y, strength)

    def step(self, s, m):
        before = s.copy()
        mv = np.asarray(m, dtype=np.int32)
        self.lib.ref_step(s.ctypes.data, mv.ctypes.data)
        self._record(before, m, s)
