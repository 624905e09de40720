"""Code written against pomcpp's C++ API compiled against include/pom_bboard.hpp (the drop-in surface) —
compile+link everywhere, run on the GPU.

Three layers:
  * tests/cpp/dropin_test.cpp — reference-style test code and home-made Agent subclasses;
  * the reference's UNMODIFIED src/agents/simple_agent.cpp, src/agents/basic_agents.cpp, src/bboard/strategy.cpp and
    src/main.cpp compiled where they lie against the header (tests/cpp/shim/bboard.hpp / step_utility.hpp only include
    it; agents.hpp / strategy.hpp / colors.hpp are reached through symlinks in the git-ignored build dir) — only where
    /root/reference exists, i.e. in the build container;
  * tests/cpp/env_game.cpp — games through bboard::Environment (the reference's game-loop surface, bboard.hpp:541-644)
    on the GPU, replayed through the oracle's restatement of Environment::Step (environment.cpp:123-169), with scripted
    agents and — where the build container left the binary under oracle/_ref/ — with the reference's own SimpleAgent.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from pomcpp_amd.state import STATE_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
EXE = os.path.join(BUILD, "dropin_test")
REF = "/root/reference"
REF_OUT = os.path.join(ROOT, "oracle", "_ref")  # binaries built from reference sources live only here (git-ignored)
LINK = ["-L" + os.path.join(ROOT, "pomcpp_amd"), "-lpom_batch", "-Wl,-rpath," + os.path.join(ROOT, "pomcpp_amd"),
        "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-pthread"]
INC = ["-I" + os.path.join(ROOT, "tests", "cpp", "shim"), "-I" + os.path.join(ROOT, "include")]

have_reference = os.path.isdir(os.path.join(REF, "src", "bboard"))


@pytest.fixture(scope="module")
def dropin_exe(hip_lib):
    os.makedirs(BUILD, exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "dropin_test.cpp"), "-o", EXE] + LINK, check=True)
    return EXE


@pytest.fixture(scope="module")
def env_game_exe(hip_lib):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "env_game")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror"] + INC +
                   [os.path.join(ROOT, "tests", "cpp", "env_game.cpp"), "-o", exe] + LINK, check=True)
    return exe


def test_reference_style_code_compiles_against_the_drop_in_header(dropin_exe):
    assert os.path.exists(dropin_exe)


def test_reference_general_vectors_on_the_drop_in_fixed_queue(hip_lib):
    """unit_test/bboard/general_test.cpp:8-61 ("[general]": FixedQueue fill / PopElem / RemoveAt with the ring starting at 0, 5, 2)
    against include/pom_bboard.hpp's FixedQueue, the one drop-in agents compile against; pure host code, runs without a GPU"""
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "general_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "general_test.cpp"), "-o", exe] + LINK, check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "general ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(not have_reference, reason="the reference tree only exists in the build container")
def test_unmodified_reference_agents_and_main_compile_and_link_against_the_drop_in_header(hip_lib):
    """oracle/Makefile target `dropin` (also run by __graft_entry__.build()): src/agents/simple_agent.cpp, basic_agents.cpp,
    src/bboard/strategy.cpp and src/main.cpp compiled where they lie against include/pom_bboard.hpp; main.cpp — SimpleAgents +
    bboard::Environment::MakeGame / GetState / StartGame — linked into oracle/_ref/dropin_main, and the trace program with the
    reference's own SimpleAgent as the four players into oracle/_ref/env_game_ref (runs on the GPU box, test below)"""
    for f in ("dropin_main", "env_game_ref", "dropin_main.o", "dropin_simple_agent.o", "dropin_basic_agents.o", "dropin_strategy.o"):
        path = os.path.join(REF_OUT, f)
        if os.path.exists(path):
            os.remove(path)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dropin"], check=True)
    assert os.path.exists(os.path.join(REF_OUT, "dropin_main")) and os.path.exists(os.path.join(REF_OUT, "env_game_ref"))


def test_host_init_board_items_is_the_boardgen_specification(env_game_exe, oracle, tmp_path):
    """InitState of the header = the board (seed, env 0, episode 0) of include/pom_boardgen.h, as the device generator draws it"""
    for seed in (0x1337, 5, 123456789):
        out = tmp_path / f"b{seed}.bin"
        subprocess.run([env_game_exe, "--board", str(out), str(seed)], check=True, timeout=60)
        got = np.fromfile(out, dtype=STATE_DTYPE)
        want = oracle.boardgen(seed, [0], [0])
        got["agents"]["pad"] = 0
        assert got.tobytes() == want.tobytes()


def _replay(trace_path, oracle):
    """every step of every recorded game must be what the oracle's Environment::Step makes of the same state and moves"""
    buf = open(trace_path, "rb").read()
    pos, games, steps, finished = 0, 0, 0, 0
    while pos < len(buf):
        magic, _g = struct.unpack_from("<ii", buf, pos)
        assert magic == 0x504F4D45
        pos += 8
        ref = np.frombuffer(buf, dtype=STATE_DTYPE, count=1, offset=pos).copy()
        pos += 1004
        assert ref["timeStep"][0] == 0
        status = dict(done=0, winner=-1, draw=0)
        n = 0
        while True:
            rec = struct.unpack_from("<8i", buf, pos)
            pos += 32
            if rec[0] == 2:
                assert rec[1] == n and (rec[5], rec[6], rec[7]) == (status["done"], status["winner"], status["draw"])
                break
            assert rec[0] == 1
            got = np.frombuffer(buf, dtype=STATE_DTYPE, count=1, offset=pos).copy()
            pos += 1004
            dead_before = ref["agents"]["dead"][0].copy()
            moves = list(rec[1:5])
            assert all(m == 0 for m, d in zip(moves, dead_before) if d), "a dead agent was asked for a move"
            oracle.env_step(ref, moves, status)  # ticks on which the reference itself has UB take the documented fallback on both sides
            want = ref.copy()
            want["agents"]["pad"] = 0
            got["agents"]["pad"] = 0
            assert got.tobytes() == want.tobytes(), f"game {games} step {n}"
            assert (rec[5], rec[6], rec[7]) == (status["done"], status["winner"], status["draw"])
            n += 1
        games += 1
        steps += n
        finished += status["done"]
    return games, steps, finished


@pytest.mark.gpu
def test_reference_style_code_runs_on_the_gpu(dropin_exe):
    out = subprocess.run([dropin_exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "dropin ok" in out.stdout, f"rc={out.returncode}\n{out.stdout}\n{out.stderr}"


@pytest.mark.gpu
def test_games_through_bboard_environment_equal_the_oracles_environment_step(env_game_exe, oracle, tmp_path):
    trace = tmp_path / "trace.bin"
    out = subprocess.run([env_game_exe, str(trace), "12", "200"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "env games ok" in out.stdout, f"rc={out.returncode}\n{out.stdout}\n{out.stderr}"
    games, steps, finished = _replay(trace, oracle)
    assert games == 12 and steps > 300 and finished >= 1


@pytest.mark.gpu
def test_reference_simple_agents_play_through_bboard_environment(hip_lib, oracle, tmp_path):
    """the reference's own agents::SimpleAgent (unmodified sources, linked in the build container) as the four players"""
    exe = os.path.join(REF_OUT, "env_game_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/env_game_ref is built only where /root/reference exists")
    trace = tmp_path / "trace_ref.bin"
    out = subprocess.run([exe, str(trace), "6", "300"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "env games ok" in out.stdout, f"rc={out.returncode}\n{out.stdout}\n{out.stderr}"
    games, steps, _ = _replay(trace, oracle)
    assert games == 6 and steps > 200
