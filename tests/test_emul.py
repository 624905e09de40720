"""Kernel logic without a GPU: the device tick body (pomcpp_amd/csrc/pom_step_body.h) compiled for the host — one lane per env
over a plain-array store (tests/emul/pom_emul.cpp), and in the SHIPPED shape, four lanes per env, as four threads that meet at
every cross-lane operation (tests/emul/pom_emul_quad.cpp) — and fuzzed against the oracle, plus pack/unpack round trips.  These
are test builds only — the product library contains no host stepper."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.state import STATE_DTYPE, Item

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
INC = ["-I" + os.path.join(ROOT, p) for p in ("include", "pomcpp_amd/csrc", "oracle")]


@pytest.fixture(scope="module")
def emul_bins():
    os.makedirs(BUILD, exist_ok=True)
    run = lambda *a: subprocess.run(list(a), check=True, cwd=ROOT)
    run("g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-fPIC", *INC, "-c", "tests/emul/pom_emul.cpp", "-o", "build/pom_emul.o")
    run("gcc", "-O2", "-std=c11", "-fPIC", *INC, "-c", "oracle/pom_oracle.c", "-o", "build/pom_oracle.o")
    run("g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-fPIC", "-pthread", *INC, "-c", "tests/emul/pom_emul_quad.cpp", "-o", "build/pom_emul_quad.o")
    run("g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-pthread", *INC, "tests/emul/emul_fuzz.cpp", "build/pom_emul.o", "build/pom_emul_quad.o",
        "build/pom_oracle.o", "-o", "build/emul_fuzz")
    run("g++", "-shared", "-pthread", "-o", "build/libpom_emul.so", "build/pom_emul.o", "build/pom_emul_quad.o")
    lib = C.CDLL(os.path.join(BUILD, "libpom_emul.so"))
    for fn in (lib.pom_emul_step, lib.pom_emul_quad_step):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        fn.restype = C.c_uint32
    return lib


@pytest.mark.parametrize("scenario", [0, 1, 2, 3])
def test_device_tick_body_matches_oracle_under_random_play(emul_bins, scenario):
    out = subprocess.run([os.path.join(BUILD, "emul_fuzz"), str(scenario), "150000", "5"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "0 mismatches" in out.stdout


@pytest.mark.parametrize("scenario", [0, 1, 2, 3])
def test_quad_tick_body_matches_oracle_under_random_play(emul_bins, scenario):
    """the shipped shape, four lanes per env: states equal the oracle's and the lanes never break what the device assumes of them
    (same cross-lane operation reached by all four, no conflicting writes, replicated registers identical)"""
    out = subprocess.run([os.path.join(BUILD, "emul_fuzz"), str(scenario), "60000", "11", "quad"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "0 mismatches" in out.stdout and "four lanes per env" in out.stdout


def test_pack_unpack_round_trip_and_rejections(emul_bins):
    idle = np.zeros(4, dtype=np.int32)
    boards = pa.make_boards(64, seed=9, kind="stress")
    for i in range(len(boards)):
        s = boards[i:i + 1].copy()
        s["board"][0, 5, 5] = Item.BOMB  # still a legal value
        before = s.copy()
        st = C.c_uint32(1)  # done in ENV mode: the tick is skipped, only pack -> unpack happens
        assert emul_bins.pom_emul_step(s.ctypes.data, idle.ctypes.data, 1, 0, C.byref(st)) == 0
        assert s.tobytes() == before.tobytes()
    bad = pa.new_states(1)
    for poke in (lambda s: s["board"].__setitem__((0, 3, 3), 0x5000),          # between the item ranges
                 lambda s: s["board"].__setitem__((0, 3, 3), -1),
                 lambda s: s["board"].__setitem__((0, 3, 3), Item.AGENT0 + 4),  # a fifth agent
                 lambda s: s["agents"]["x"].__setitem__((0, 1), 11),
                 lambda s: s["bombs_count"].__setitem__(0, 21),
                 lambda s: s["bombs_index"].__setitem__(0, 20),
                 lambda s: s["agents"]["bombStrength"].__setitem__((0, 0), 256)):
        s = bad.copy()
        poke(s)
        assert emul_bins.pom_emul_step(s.ctypes.data, idle.ctypes.data, 0, 0, None) == 0xFFFFFFFF


def test_tile_layout_of_the_device_buffers(emul_bins):
    """pom_packed.h: tiles of 16 envs, struct-of-arrays inside a tile; every env's dwords at the documented places, disjoint"""
    n = 100  # not a multiple of 16
    s = pa.make_boards(n, seed=21, kind="stress")
    s["agents"]["pad"] = 0
    out = np.zeros(n, dtype=STATE_DTYPE)
    emul_bins.pom_emul_tile_roundtrip.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    assert emul_bins.pom_emul_tile_roundtrip(s.ctypes.data, n, out.ctypes.data) == 0
    assert out.tobytes() == s.tobytes()


def test_env_epilogue_timeout_and_winner(emul_bins):
    s = pa.new_states(1)
    pa.put_agents_in_corners(s[0])
    pa.kill(s[0], 1, 2)
    st = C.c_uint32(0)
    mv = np.zeros(4, dtype=np.int32)
    emul_bins.pom_emul_step(s.ctypes.data, mv.ctypes.data, 1, 3, C.byref(st))
    assert st.value == 0 and s["timeStep"][0] == 1
    pa.kill(s[0], 0)
    emul_bins.pom_emul_step(s.ctypes.data, mv.ctypes.data, 1, 3, C.byref(st))
    assert st.value & 1 and ((st.value >> 2) & 7) - 1 == 3  # agent 3 won
    s2 = pa.new_states(1)
    pa.put_agents_in_corners(s2[0])
    st = C.c_uint32(0)
    for _ in range(3):
        emul_bins.pom_emul_step(s2.ctypes.data, mv.ctypes.data, 1, 3, C.byref(st))
    assert st.value & 1 and st.value & 32 and s2["timeStep"][0] == 3  # done by the tick cap


def test_device_tick_body_reproduces_every_reference_step_case(emul_bins, oracle):
    """the restated [step function] suite + quirk vectors (tests/step_cases.py) through the host build of the device body,
    against the states the compiled reference produced (tests/golden/step_cases.npz): covers paths random play does not
    reach (out-of-order timers, full queues, dependency cycles)"""
    from tests.case_api import HostAPI
    from tests.step_cases import ALL_CASES
    golden = np.load(os.path.join(ROOT, "tests", "golden", "step_cases.npz"))

    for step_fn in (emul_bins.pom_emul_step, emul_bins.pom_emul_quad_step):  # one lane per env; four (the shipped shape)
        def stepper(state, moves):
            mv = np.ascontiguousarray(moves, dtype=np.int32)
            buf = np.ascontiguousarray(state).reshape(1)
            flags = step_fn(buf.ctypes.data, mv.ctypes.data, 0, 0, None)
            assert flags != 0xFFFFFFFF and not (flags & 0x40000000)  # representable; the quad model's checks hold
            state[...] = buf.reshape(state.shape)

        for name, case in ALL_CASES.items():
            api = HostAPI(stepper, oracle)
            case(api)
            after = golden[f"{name}__after"]
            assert len(api.trace) == len(after)
            for k, (_b, _m, a) in enumerate(api.trace):
                assert a == after[k].tobytes(), f"{name}: state after step {k} differs from the reference's"


def test_chain_visit_distance_is_signed_and_survives_the_field_width(emul_bins):
    """Chained launches (pomcpp_amd/csrc/pom_chain.h): a wavefront plays tick tick0 + (its visit - the first visit of the call its
    launch belongs to).  Launches of two calls are in flight together, so the difference can be negative; visits live in 28-bit
    fields that are zeroed before they reach 2^27."""
    import ctypes as C
    f = emul_bins.pom_emul_chain_visit_distance
    f.restype, f.argtypes = C.c_uint32, [C.c_uint32, C.c_uint32]
    as_signed = lambda v: v - (1 << 32) if v >= (1 << 31) else v
    for visit, first in ((0, 0), (5, 0), (47, 47), (45, 47), (0, 3), ((1 << 27) - 1, 0), (0, (1 << 27) - 1), (100, (1 << 27) - 5),
                         ((1 << 27) - 5, 100), (123456, 123400)):
        assert as_signed(f(visit, first)) == visit - first, (visit, first)
    # a tick is tick0 + distance in 32-bit arithmetic: an earlier call's ticket gives an earlier tick
    assert (1000 + f(45, 47)) & 0xFFFFFFFF == 998
