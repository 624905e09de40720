"""ctypes access to the CPU checker (oracle/libpom_oracle.so) — test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from pomcpp_amd.state import STATE_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class _EnvStatus(C.Structure):
    _fields_ = [("done", C.c_int32), ("winner", C.c_int32), ("draw", C.c_int32)]


class Oracle:
    _lib = None

    def __init__(self):
        if Oracle._lib is None:
            path = os.path.join(ORACLE_DIR, "libpom_oracle.so")
            subprocess.run(["make", "-s", "-C", ORACLE_DIR, "libpom_oracle.so"], check=True)
            lib = C.CDLL(path)
            VP, I = C.c_void_p, C.c_int
            lib.pom_oracle_step.argtypes = [VP, VP]
            lib.pom_oracle_step.restype = C.c_uint32
            lib.pom_oracle_env_step.argtypes = [VP, VP, C.POINTER(_EnvStatus)]
            lib.pom_oracle_env_step.restype = C.c_uint32
            lib.pom_oracle_init_state.argtypes = [VP]
            lib.pom_oracle_put_agent.argtypes = [VP, I, I, I]
            lib.pom_oracle_put_agents_in_corners.argtypes = [VP, I, I, I, I]
            lib.pom_oracle_kill.argtypes = [VP, I]
            lib.pom_oracle_plant_bomb.argtypes = [VP, I, I, I, I, I]
            lib.pom_oracle_spawn_flame.argtypes = [VP, I, I, I]
            lib.pom_oracle_run_random.argtypes = [VP, VP, I, I, C.c_uint64, I, I, I, I]
            lib.pom_oracle_run_random.restype = C.c_int64
            lib.pom_oracle_dest_pos.argtypes = [VP, VP, VP]
            lib.pom_oracle_fix_switch_move.argtypes = [VP, VP]
            lib.pom_oracle_resolve_dependencies.argtypes = [VP, VP, VP, VP]
            lib.pom_oracle_resolve_dependencies.restype = I
            lib.pom_oracle_simple_act.argtypes = [VP, I, VP, I]
            lib.pom_oracle_simple_act.restype = C.c_int32
            lib.pom_oracle_simple_policy.argtypes = [VP, VP, I, C.c_uint64, I, I, VP, VP]
            lib.pom_oracle_run_simple.argtypes = [VP, VP, VP, I, I, C.c_uint64, I, I, I]
            lib.pom_oracle_run_simple.restype = C.c_int64
            lib.pom_oracle_boardgen.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, VP]
            lib.pom_oracle_boardgen.restype = None
            lib.pom_oracle_run_random_fresh.argtypes = [VP, VP, I, I, C.c_uint64, C.c_uint64, I, I, I, I]
            lib.pom_oracle_run_random_fresh.restype = C.c_int64
            lib.pom_oracle_run_simple_fresh.argtypes = [VP, VP, VP, I, I, C.c_uint64, C.c_uint64, I, I, I]
            lib.pom_oracle_run_simple_fresh.restype = C.c_int64
            Oracle._lib = lib
        self.lib = Oracle._lib

    @staticmethod
    def _one(state: np.ndarray) -> np.ndarray:
        assert state.dtype == STATE_DTYPE and state.size == 1 and state.flags["C_CONTIGUOUS"]
        return state

    def step(self, state: np.ndarray, moves) -> int:
        """bboard::Step on one state in place; returns the POM_UB_* flags."""
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        return int(self.lib.pom_oracle_step(self._one(state).ctypes.data, mv.ctypes.data))

    def step_batch(self, states: np.ndarray, moves: np.ndarray) -> np.ndarray:
        """bboard::Step on each state of a contiguous array; returns per-env flags."""
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        flags = np.zeros(states.size, dtype=np.uint32)
        base = states.ctypes.data
        for i in range(states.size):
            flags[i] = self.lib.pom_oracle_step(base + i * 1004, mv[i].ctypes.data)
        return flags

    def env_step(self, state: np.ndarray, moves, status: dict) -> int:
        st = _EnvStatus(status["done"], status["winner"], status["draw"])
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        ub = self.lib.pom_oracle_env_step(self._one(state).ctypes.data, mv.ctypes.data, C.byref(st))
        status.update(done=st.done, winner=st.winner, draw=st.draw)
        return int(ub)

    def spawn_flame(self, state, x, y, strength):
        self.lib.pom_oracle_spawn_flame(self._one(state).ctypes.data, x, y, strength)

    def run_random(self, states: np.ndarray, initial: np.ndarray, ticks: int, seed: int, first_env: int, tick0: int,
                   dist: int, max_steps: int) -> int:
        assert states.flags["C_CONTIGUOUS"] and initial.flags["C_CONTIGUOUS"]
        return int(self.lib.pom_oracle_run_random(states.ctypes.data, initial.ctypes.data, states.size, ticks, seed,
                                                  first_env, tick0, dist, max_steps))

    # ---- SimpleAgent policy (oracle/pom_policy_oracle.c); agent memory as int32[n, 4, 16] ----
    def simple_policy(self, states: np.ndarray, mems: np.ndarray, seed: int, first_env: int, tick: int, done=None) -> np.ndarray:
        assert mems.dtype == np.int32 and mems.shape == (states.size, 4, 16) and mems.flags["C_CONTIGUOUS"]
        moves = np.zeros((states.size, 4), dtype=np.int32)
        d = None if done is None else np.ascontiguousarray(done, dtype=np.int32)
        self.lib.pom_oracle_simple_policy(states.ctypes.data, mems.ctypes.data, states.size, seed, first_env, tick,
                                          None if d is None else d.ctypes.data, moves.ctypes.data)
        return moves

    def run_simple(self, states: np.ndarray, initial: np.ndarray, mems: np.ndarray, ticks: int, seed: int, first_env: int, tick0: int,
                   max_steps: int) -> int:
        assert mems.dtype == np.int32 and mems.shape == (states.size, 4, 16)
        return int(self.lib.pom_oracle_run_simple(states.ctypes.data, initial.ctypes.data, mems.ctypes.data, states.size, ticks, seed,
                                                  first_env, tick0, max_steps))

    # ---- start boards (oracle/pom_boardgen_oracle.c) ----
    def boardgen(self, seed: int, envs, episodes) -> np.ndarray:
        """the start State of (seed, env, episode) for each pair of the two equally long sequences"""
        envs, episodes = np.asarray(envs, dtype=np.int64), np.asarray(episodes, dtype=np.int64)
        out = np.zeros(envs.size, dtype=STATE_DTYPE)
        for i in range(envs.size):
            self.lib.pom_oracle_boardgen(seed, int(envs[i]), int(episodes[i]), out.ctypes.data + i * 1004)
        return out

    def run_random_fresh(self, states: np.ndarray, episodes: np.ndarray, ticks: int, seed: int, board_seed: int, first_env: int,
                         tick0: int, dist: int, max_steps: int) -> int:
        assert states.flags["C_CONTIGUOUS"] and episodes.dtype == np.int32 and episodes.shape == (states.size,)
        return int(self.lib.pom_oracle_run_random_fresh(states.ctypes.data, episodes.ctypes.data, states.size, ticks, seed,
                                                        board_seed, first_env, tick0, dist, max_steps))

    def run_simple_fresh(self, states: np.ndarray, episodes: np.ndarray, mems: np.ndarray, ticks: int, seed: int, board_seed: int,
                         first_env: int, tick0: int, max_steps: int) -> int:
        assert mems.dtype == np.int32 and mems.shape == (states.size, 4, 16) and episodes.dtype == np.int32
        return int(self.lib.pom_oracle_run_simple_fresh(states.ctypes.data, episodes.ctypes.data, mems.ctypes.data, states.size,
                                                        ticks, seed, board_seed, first_env, tick0, max_steps))
