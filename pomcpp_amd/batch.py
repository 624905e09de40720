"""ctypes binding of libpom_batch.so (include/pom_batch.h) and `BatchEnvironment`.

`BatchEnvironment` mirrors `bboard::Environment` (/root/reference/include/bboard.hpp:541-644,
src/bboard/environment.cpp:48-213) for n concurrent games on one MI355X:

    reference (one game)                 here (n games)
    env.MakeGame(agents)                 env.make_game(states)            # states: STATE_DTYPE[n]
    env.Step()  -> act x4, bboard::Step  env.step(moves)                  # moves: int32[n,4], dead agents included
    env.IsDone() / IsDraw() / GetWinner  env.is_done() / is_draw() / get_winner()   # arrays of n
    env.GetState()                       env.get_state(first, count)      # STATE_DTYPE[count]

The library is the only stepper.  If it is missing or no HIP device is present, loading or
creating a batch raises — nothing falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref
from typing import Optional

import numpy as np

from .state import STATE_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
MODE_RAW, MODE_ENV = 0, 1
RESET_OFF, RESET_AT_START, RESET_AT_END = 0, 1, 2  # PomBatchOptions.auto_reset (True = RESET_AT_START)
DIST_HARMLESS, DIST_RANDOM, DIST_STRESS = 0, 1, 2
CNT_STEPS, CNT_EPISODES, CNT_RESETS, CNT_UB_TICKS = 0, 1, 2, 3
ISSUE_AUTO, ISSUE_DIRECT, ISSUE_THREADS, ISSUE_GRAPH, ISSUE_CHAIN = 0, 1, 2, 3, 4  # PomBatchOptions.issue_mode
UB_LOST_AGENT, UB_NULL_BOMB, UB_QUEUE_OVERFLOW, UB_REVERT_LOOP, UB_BAD_INDEX = 1, 2, 4, 8, 16


class PomError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"pom_batch error {code}: {text}")
        self.code = code


class _Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p),
        ("mode", C.c_int32), ("auto_reset", C.c_int32), ("max_steps", C.c_int32),
        ("env_offset", C.c_int64), ("envs_per_wave", C.c_int32), ("streams", C.c_int32),
        ("lanes_per_env", C.c_int32), ("fresh_boards", C.c_int32), ("board_seed", C.c_uint64),
        ("issue_mode", C.c_int32), ("reserved_", C.c_int32),
    ]


def library_path() -> str:
    return os.environ.get("POM_LIB") or os.path.join(_HERE, "libpom_batch.so")  # POM_LIB: experimental builds only


_lib = None


def load_library() -> C.CDLL:
    """Load the HIP extension built in-tree by `__graft_entry__.build()`; fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch-ROCm wheels carry their own HIP runtime.  A process that is going to use both (observe(), bench.py) must load
    # torch's first so that this library binds to the same runtime; the other order leaves torch without a device.
    if "torch" not in sys.modules and not os.environ.get("POM_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the gfx950 extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "pomcpp_amd has no CPU stepper to fall back to.")
    lib = C.CDLL(path)
    P, I32, I64, U64, VP = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p
    lib.pom_last_error.restype = C.c_char_p
    lib.pom_device_count.restype = C.c_int
    lib.pom_batch_create.argtypes = [C.POINTER(P), I64, C.POINTER(_Options)]
    lib.pom_batch_destroy.argtypes = [P]
    lib.pom_batch_size.argtypes = [P]
    lib.pom_batch_size.restype = I64
    lib.pom_batch_observe.argtypes = [P, VP, I32, I32, VP, VP]
    if not os.environ.get("POM_LIB") or hasattr(lib, "pom_batch_step_device_observe"):
        lib.pom_batch_step_device_observe.argtypes = [P, VP, VP, I32, I32, VP, VP]
    if hasattr(lib, "pom_batch_step_device_range"):
        lib.pom_batch_step_device_range.argtypes = [P, I64, I64, VP, VP, VP, I32, I32, VP, VP]
        lib.pom_bench_policy.argtypes = [VP, VP, I64, I64, C.c_uint32, VP]
    lib.pom_batch_stream.argtypes = [P, C.POINTER(C.c_void_p)]
    lib.pom_batch_moves_device.argtypes = [P, C.POINTER(C.POINTER(C.c_int32))]
    lib.pom_batch_generate.argtypes = [P, U64]
    lib.pom_batch_episodes.argtypes = [P, I64, I64, VP]
    lib.pom_batch_upload.argtypes = [P, VP, I64, I64]
    lib.pom_batch_download.argtypes = [P, VP, I64, I64]
    lib.pom_batch_snapshot.argtypes = [P]
    lib.pom_batch_step.argtypes = [P, VP]
    lib.pom_batch_step_device.argtypes = [P, VP]
    if not os.environ.get("POM_LIB") or hasattr(lib, "pom_batch_step_device_many"):
        lib.pom_batch_step_device_many.argtypes = [P, VP, I32]
        lib.pom_batch_chain_stats.argtypes = [P, VP]
    lib.pom_batch_step_random.argtypes = [P, U64, I32, I32, I32]
    lib.pom_batch_set_tick.argtypes = [P, I64]
    lib.pom_batch_policy_simple.argtypes = [P, U64, VP]
    lib.pom_batch_step_policy.argtypes = [P]
    lib.pom_batch_step_simple.argtypes = [P, U64, I32]
    lib.pom_batch_policy_memory.argtypes = [P, I64, I64, VP]
    lib.pom_batch_status.argtypes = [P, I64, I64, VP, VP, VP, VP, VP, VP]
    if not os.environ.get("POM_LIB") or hasattr(lib, "pom_batch_last_results"):  # (POM_LIB: older experimental builds lack these)
        lib.pom_batch_last_results.argtypes = [P, I64, I64, VP, VP, VP, VP, VP]
        lib.pom_batch_download_terminal.argtypes = [P, VP, I64, I64]
    lib.pom_batch_counters.argtypes = [P, VP]
    lib.pom_batch_counters_device.argtypes = [P, VP]
    lib.pom_batch_reset_counters.argtypes = [P]
    lib.pom_batch_sync.argtypes = [P]
    lib.pom_batch_flush.argtypes = [P]
    if not os.environ.get("POM_LIB") or hasattr(lib, "pom_batch_fork"):
        lib.pom_batch_fork.argtypes = [P]
    lib.pom_batch_set_streams.argtypes = [P, I32]
    lib.pom_batch_profile.argtypes = [P, C.c_int]
    lib.pom_batch_profile_read.argtypes = [P, C.POINTER(C.c_double), C.POINTER(I64)]
    lib.pom_batch_launch_shape.argtypes = [P, C.POINTER(I32), C.POINTER(I32), C.POINTER(I32)]
    if not os.environ.get("POM_LIB") or hasattr(lib, "pom_batch_issue_info"):
        lib.pom_batch_issue_info.argtypes = [P, C.POINTER(I32), C.POINTER(I32)]
    lib.pom_batch_device_view.argtypes = [P, C.POINTER(VP), C.POINTER(I64), C.POINTER(I32)]
    if not os.environ.get("POM_LIB") or hasattr(lib, "pom_chain_litmus"):
        lib.pom_chain_litmus.argtypes = [I32, I64, I32, I32, VP]
    lib.pom_step.argtypes = [VP, VP]
    if not os.environ.get("POM_LIB") or hasattr(lib, "pom_env_step"):
        lib.pom_env_step.argtypes = [VP, VP, I32, VP, VP, VP, VP]
    _lib = lib
    return lib


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise PomError(rc, lib.pom_last_error().decode(errors="replace"))


def chain_litmus(tiles: int, launches: int, streams: int, device: int = 0) -> dict:
    """pom_chain_litmus: the hand-off of chained launches tested by itself (every visit checks the whole record the visit before
    left)"""
    lib = load_library()
    out = np.zeros(6, dtype=np.int64)
    _check(lib, lib.pom_chain_litmus(device, tiles, launches, streams, out.ctypes.data))
    return dict(zip(("bad_records", "bad_dwords", "visits", "visits_expected", "tiles_wrong", "flags"), (int(v) for v in out)))


def step_one(state: np.ndarray, moves) -> None:
    """`bboard::Step(State*, Move*)` on one host State, executed on the GPU (pom_step)."""
    lib = load_library()
    assert state.dtype == STATE_DTYPE and state.size == 1
    mv = np.ascontiguousarray(moves, dtype=np.int32)
    assert mv.shape == (4,)
    buf = np.ascontiguousarray(state).reshape(1)
    _check(lib, lib.pom_step(buf.ctypes.data, mv.ctypes.data))
    state[...] = buf.reshape(state.shape)


def env_step_one(state: np.ndarray, moves, max_steps: int = 0) -> dict:
    """`Environment::Step`'s tick + bookkeeping on one host State (pom_env_step): timeStep++, done / winner / draw of this
    tick.  The caller does not step a finished game (environment.cpp:125)."""
    lib = load_library()
    assert state.dtype == STATE_DTYPE and state.size == 1
    mv = np.ascontiguousarray(moves, dtype=np.int32)
    assert mv.shape == (4,)
    buf = np.ascontiguousarray(state).reshape(1)
    out = np.zeros(4, dtype=np.int32)
    _check(lib, lib.pom_env_step(buf.ctypes.data, mv.ctypes.data, int(max_steps), out[0:].ctypes.data, out[1:].ctypes.data,
                                 out[2:].ctypes.data, out[3:].ctypes.data))
    state[...] = buf.reshape(state.shape)
    return {"done": int(out[0]), "winner": int(out[1]), "draw": int(out[2]), "ubflags": int(out[3]) & 0xFFFFFFFF}


class BatchEnvironment:
    def __init__(self, n_envs: int, device: int = 0, mode: int = MODE_ENV, auto_reset: bool = False,
                 max_steps: int = 0, env_offset: int = 0, stream: Optional[int] = None, envs_per_wave: int = 0,
                 streams: int = 0, lanes_per_env: int = 0, fresh_boards: bool = False, board_seed: int = 0,
                 issue_mode: int = ISSUE_AUTO):
        self._lib = load_library()
        self._h = C.c_void_p()
        self._views = []  # weak references to tensors that alias the handle's device memory (moves_tensor)
        self._tapes = []  # move tapes handed to step_device_many, kept alive until the handle has been synchronised
        self.n = int(n_envs)
        self.device = int(device)
        o = _Options(C.sizeof(_Options), device, stream, mode, int(auto_reset), max_steps, env_offset, envs_per_wave, streams,
                     lanes_per_env, int(fresh_boards), board_seed, issue_mode, 0)
        _check(self._lib, self._lib.pom_batch_create(C.byref(self._h), self.n, C.byref(o)))

    def close(self) -> None:
        if any(r() is not None for r in getattr(self, "_views", ())):
            raise RuntimeError("close(): a tensor returned by moves_tensor() still views this handle's device memory; "
                               "delete it first")
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pom_batch_destroy(self._h)  # waits for everything queued
            self._h = C.c_void_p()
            self._tapes = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- Environment::MakeGame / GetState ---------------------------------------------------
    def make_game(self, states: np.ndarray, first: int = 0) -> None:
        """Upload start states; they also become the reset snapshot (MakeGame, environment.cpp:53-66)."""
        st = np.ascontiguousarray(states, dtype=STATE_DTYPE)
        _check(self._lib, self._lib.pom_batch_upload(self._h, st.ctypes.data, first, st.size))

    upload = make_game

    def generate(self, board_seed: int) -> None:
        """Start boards drawn on the device (pom_batch_generate, include/pom_boardgen.h): InitState's distribution, no host."""
        _check(self._lib, self._lib.pom_batch_generate(self._h, board_seed))

    def episodes(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """games started so far per env (0 = still the first)"""
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=np.uint32)
        _check(self._lib, self._lib.pom_batch_episodes(self._h, first, count, out.ctypes.data))
        return out

    def get_state(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=STATE_DTYPE)
        _check(self._lib, self._lib.pom_batch_download(self._h, out.ctypes.data, first, count))
        self._tapes.clear()  # the call has synchronised the handle: nothing reads a tape any more
        return out

    download = get_state

    def snapshot(self) -> None:
        _check(self._lib, self._lib.pom_batch_snapshot(self._h))

    # ---- Environment::Step --------------------------------------------------------------------
    def step(self, moves: np.ndarray) -> None:
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        if mv.shape != (self.n, 4):
            raise ValueError(f"moves must be int32[{self.n}, 4] (dead agents included), got {mv.shape}")
        _check(self._lib, self._lib.pom_batch_step(self._h, mv.ctypes.data))

    def step_device(self, moves) -> None:
        """moves: a device tensor int32[n,4] on this handle's device (anything with data_ptr / shape / dtype, e.g. a torch
        tensor: shape, element type, contiguity and device are checked), or the raw device address of such an array."""
        if hasattr(moves, "data_ptr"):
            shape = tuple(getattr(moves, "shape", ()))
            if shape != (self.n, 4):
                raise ValueError(f"moves must be int32[{self.n}, 4] (dead agents included), got shape {shape}")
            if "int32" not in str(getattr(moves, "dtype", "")):
                raise ValueError(f"moves must be int32, got {getattr(moves, 'dtype', None)}")
            if hasattr(moves, "is_contiguous") and not moves.is_contiguous():
                raise ValueError("moves must be contiguous")
            dev = getattr(moves, "device", None)
            if dev is not None and (getattr(dev, "type", "cuda") != "cuda" or getattr(dev, "index", self.device) not in (None, self.device)):
                raise ValueError(f"moves live on {dev}, the batch on device {self.device}")
            self._after_torch(moves)
            moves = moves.data_ptr()
        _check(self._lib, self._lib.pom_batch_step_device(self._h, int(moves)))

    def _after_torch(self, tensor) -> None:
        """order the handle's stream behind torch's current stream: a tensor the caller has just produced there is complete
        before the step that reads it begins (no-op if both are the same stream)"""
        if "torch" not in sys.modules or not hasattr(tensor, "is_cuda"):
            return
        import torch
        dev = torch.device("cuda", self.device)
        mine, theirs = torch.cuda.ExternalStream(self.stream_handle(), device=dev), torch.cuda.current_stream(dev)
        if mine.cuda_stream != theirs.cuda_stream:
            mine.wait_stream(theirs)

    def step_device_many(self, moves, ticks: Optional[int] = None) -> None:
        """K ticks with explicit moves from a tape in device memory: a tensor int32[K, n, 4] on this handle's device (or the raw
        device address of one, with `ticks` = K).  Chained launches where the handle chains (pom_batch_step_device_many)."""
        if hasattr(moves, "data_ptr"):
            shape = tuple(getattr(moves, "shape", ()))
            if len(shape) != 3 or shape[1:] != (self.n, 4) or (ticks is not None and ticks != shape[0]):
                raise ValueError(f"moves must be int32[ticks, {self.n}, 4] (dead agents included), got shape {shape}")
            if "int32" not in str(getattr(moves, "dtype", "")):
                raise ValueError(f"moves must be int32, got {getattr(moves, 'dtype', None)}")
            if hasattr(moves, "is_contiguous") and not moves.is_contiguous():
                raise ValueError("moves must be contiguous")
            dev = getattr(moves, "device", None)
            if dev is not None and (getattr(dev, "type", "cuda") != "cuda" or getattr(dev, "index", self.device) not in (None, self.device)):
                raise ValueError(f"moves live on {dev}, the batch on device {self.device}")
            ticks = shape[0]
            # the tape is read asynchronously and must stay unchanged until the handle has been synchronised: keep the tensor (and
            # with it its memory) alive until then, whatever the caller does with its own reference
            if len(self._tapes) >= 4:  # a loop that never synchronises must not keep every tape alive: wait for the old ones
                self.sync()
            self._tapes.append(moves)
            self._after_torch(moves)
            moves = moves.data_ptr()
        if ticks is None:
            raise ValueError("a raw device address needs `ticks`")
        _check(self._lib, self._lib.pom_batch_step_device_many(self._h, int(moves), int(ticks)))

    def step_device_range(self, first: int, count: int, moves, stream=None, codes=None, planes=None) -> None:
        """Closed-loop stepping (pom_batch_step_device_range): one tick for the envs [first, first + count) — whole tiles of 16 — as ONE
        launch on `stream` (a raw hipStream_t / torch stream; None: the handle's stream), Move[4] from `moves` (int32[n, 4] device tensor
        or address, indexed by env).  `codes` (uint8[n, 5, 11, 11]) or `planes` (uint8[n, 16, 11, 11]): the observation of those envs after
        the tick, written by the same launch.  Nothing is forked or joined: the stream orders the call; sync() the handle once before a loop
        of these (and before capturing them into a graph)."""
        st = getattr(stream, "cuda_stream", stream)
        mv = moves.data_ptr() if hasattr(moves, "data_ptr") else int(moves)
        out, code = (codes, 3) if codes is not None else (planes, 0)
        _check(self._lib, self._lib.pom_batch_step_device_range(self._h, int(first), int(count), mv, st, out.data_ptr() if out is not None else None,
                                                                code, 0, None, None))

    def chain_stats(self) -> dict:
        """chained launches since creation: launches issued, checks run, tiles found left behind, ticks replayed for them"""
        out = np.zeros(4, dtype=np.int64)
        _check(self._lib, self._lib.pom_batch_chain_stats(self._h, out.ctypes.data))
        return dict(zip(("launches", "settles", "tiles_recovered", "ticks_replayed"), (int(v) for v in out)))

    def step_random(self, seed: int, dist: int = DIST_RANDOM, ticks: int = 1, ticks_per_launch: int = 1) -> None:
        _check(self._lib, self._lib.pom_batch_step_random(self._h, seed, dist, ticks, ticks_per_launch))

    # ---- SimpleAgent policy on the device (agents::SimpleAgent) -------------------------------------
    def policy_simple(self, seed: int, want_moves: bool = False):
        """act() of all four agents of every env into the internal move buffer; optionally returns int32[n,4]."""
        out = np.zeros((self.n, 4), dtype=np.int32) if want_moves else None
        _check(self._lib, self._lib.pom_batch_policy_simple(self._h, seed, out.ctypes.data if want_moves else None))
        return out

    def step_policy(self) -> None:
        _check(self._lib, self._lib.pom_batch_step_policy(self._h))

    def step_simple(self, seed: int, ticks: int = 1) -> None:
        _check(self._lib, self._lib.pom_batch_step_simple(self._h, seed, ticks))

    def policy_memory(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.n - first if count is None else count
        out = np.zeros((count, 4, 16), dtype=np.int32)
        _check(self._lib, self._lib.pom_batch_policy_memory(self._h, first, count, out.ctypes.data))
        return out

    def set_tick(self, tick: int) -> None:
        _check(self._lib, self._lib.pom_batch_set_tick(self._h, tick))

    # ---- IsDone / IsDraw / GetWinner ------------------------------------------------------------
    def status(self, first: int = 0, count: Optional[int] = None) -> dict:
        count = self.n - first if count is None else count
        names = ["done", "winner", "draw", "alive", "time_step", "ubflags"]
        arrs = [np.zeros(count, dtype=np.int32) for _ in names]
        _check(self._lib, self._lib.pom_batch_status(self._h, first, count, *[a.ctypes.data for a in arrs]))
        self._tapes.clear()
        out = dict(zip(names, arrs))
        out["ubflags"] = out["ubflags"].view(np.uint32)
        return out

    def last_results(self, first: int = 0, count: Optional[int] = None) -> dict:
        """auto_reset=RESET_AT_END: `finished` = the env's latest tick ended an episode (it now stands on its next start state);
        winner / draw / length / alive describe the env's most recently finished episode."""
        count = self.n - first if count is None else count
        names = ["finished", "winner", "draw", "length", "alive"]
        arrs = [np.zeros(count, dtype=np.int32) for _ in names]
        _check(self._lib, self._lib.pom_batch_last_results(self._h, first, count, *[a.ctypes.data for a in arrs]))
        return dict(zip(names, arrs))

    def get_terminal_state(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """auto_reset=RESET_AT_END: the final State of each env's most recently finished episode (zeros: none yet)"""
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=STATE_DTYPE)
        _check(self._lib, self._lib.pom_batch_download_terminal(self._h, out.ctypes.data, first, count))
        return out

    def is_done(self) -> np.ndarray:
        return self.status()["done"].astype(bool)

    def is_draw(self) -> np.ndarray:
        return self.status()["draw"].astype(bool)

    def get_winner(self) -> np.ndarray:
        return self.status()["winner"]

    # ---- observation export ---------------------------------------------------------
    def step_device_observe(self, moves, per_agent: bool = False, dtype: str = "uint8", attrs: bool = True, out=None):
        """step_device(moves) and observe(...) as ONE launch (pom_batch_step_device_observe): returns what observe() would."""
        return self.observe(per_agent=per_agent, dtype=dtype, attrs=attrs, out=out, _step_moves=moves)

    def observe(self, per_agent: bool = False, dtype: str = "uint8", attrs: bool = True, out=None, _step_moves=None):
        """Planes of every env as torch tensors on the handle's device, written by one kernel on the handle's stream
        (pom_batch_observe; plane list in include/pom_batch.h).  Returns (planes, agent_attrs, env_attrs): planes
        [n,16,11,11] or [n,4,16,11,11]; agent_attrs int32 [n,4,8]; env_attrs int32 [n,4] (None, None if attrs=False).
        dtype "codes": the compact form, uint8 [n,5,11,11] = board codes 0..13, bomb strength / life / direction, flame life
        (POM_OBS_CODES; no per-agent view).  `out` reuses a planes tensor from an earlier call.  torch is only the owner of the
        device memory here."""
        import torch
        kinds = {"uint8": (0, torch.uint8), "float16": (1, torch.float16), "float32": (2, torch.float32), "codes": (3, torch.uint8)}
        if dtype not in kinds:
            raise ValueError(f"dtype must be one of {sorted(kinds)}")
        if dtype == "codes" and per_agent:
            raise ValueError("the codes layout has no per-agent view")
        code, tdt = kinds[dtype]
        dev = torch.device("cuda", self.device)
        shape = (self.n, 5, 11, 11) if dtype == "codes" else (self.n, 4, 16, 11, 11) if per_agent else (self.n, 16, 11, 11)
        if out is None:
            out = torch.empty(shape, dtype=tdt, device=dev)
        elif tuple(out.shape) != shape or out.dtype != tdt or not out.is_contiguous() or out.device != dev:
            raise ValueError("out does not match the requested view")
        a_attrs = torch.empty((self.n, 4, 8), dtype=torch.int32, device=dev) if attrs else None
        e_attrs = torch.empty((self.n, 4), dtype=torch.int32, device=dev) if attrs else None
        # the kernel runs on the handle's stream, the tensors live on torch's current stream: order the two with events
        mine, theirs = torch.cuda.ExternalStream(self.stream_handle(), device=dev), torch.cuda.current_stream(dev)
        if mine.cuda_stream != theirs.cuda_stream:
            mine.wait_stream(theirs)
        if _step_moves is not None:
            if tuple(_step_moves.shape) != (self.n, 4) or "int32" not in str(_step_moves.dtype) or not _step_moves.is_contiguous():
                raise ValueError(f"moves must be a contiguous int32[{self.n}, 4] device tensor")
            _check(self._lib, self._lib.pom_batch_step_device_observe(self._h, _step_moves.data_ptr(), out.data_ptr(), code, int(per_agent),
                                                                      a_attrs.data_ptr() if attrs else None,
                                                                      e_attrs.data_ptr() if attrs else None))
        else:
            _check(self._lib, self._lib.pom_batch_observe(self._h, out.data_ptr(), code, int(per_agent),
                                                          a_attrs.data_ptr() if attrs else None,
                                                          e_attrs.data_ptr() if attrs else None))
        if mine.cuda_stream != theirs.cuda_stream:
            theirs.wait_stream(mine)
        return out, a_attrs, e_attrs

    def moves_tensor(self):
        """The handle's device move buffer as a torch int32 tensor [n, 4] (zero-copy): what policy_simple() fills and
        step_policy() consumes; overwrite entries on the handle's stream to mix in another policy."""
        import torch
        ptr = C.POINTER(C.c_int32)()
        _check(self._lib, self._lib.pom_batch_moves_device(self._h, C.byref(ptr)))
        addr = C.cast(ptr, C.c_void_p).value

        class _Raw:  # the CUDA array interface is how torch adopts foreign device memory
            __cuda_array_interface__ = {"shape": (self.n, 4), "typestr": "<i4", "data": (addr, False), "version": 2}

        t = torch.as_tensor(_Raw(), device=torch.device("cuda", self.device))
        # the memory belongs to the handle: the view keeps its BatchEnvironment alive (no __del__ while a view exists), and an
        # explicit close() with views outstanding is refused
        t._pom_owner = self
        self._views.append(weakref.ref(t))
        return t

    def stream_handle(self) -> int:
        """the hipStream_t (as an integer) this handle's work is ordered on"""
        s = C.c_void_p()
        _check(self._lib, self._lib.pom_batch_stream(self._h, C.byref(s)))
        return s.value or 0

    # ---- counters / plumbing ------------------------------------------------------------------------
    def counters(self) -> np.ndarray:
        out = np.zeros(4, dtype=np.int64)
        _check(self._lib, self._lib.pom_batch_counters(self._h, out.ctypes.data))
        self._tapes.clear()
        return out

    def counters_into(self, dev_ptr: int) -> None:
        _check(self._lib, self._lib.pom_batch_counters_device(self._h, dev_ptr))

    def reset_counters(self) -> None:
        _check(self._lib, self._lib.pom_batch_reset_counters(self._h))

    def sync(self) -> None:
        _check(self._lib, self._lib.pom_batch_sync(self._h))
        self._tapes.clear()

    def flush(self) -> None:
        """Make the handle's stream wait for all steps issued so far (host does not block)."""
        _check(self._lib, self._lib.pom_batch_flush(self._h))

    def fork(self) -> None:
        """Order the internal sub-streams behind the handle's stream now (the next step then needs no cross-stream event first)."""
        if hasattr(self._lib, "pom_batch_fork"):  # (older experimental builds loaded through POM_LIB lack it)
            _check(self._lib, self._lib.pom_batch_fork(self._h))

    def set_streams(self, streams: int) -> None:
        _check(self._lib, self._lib.pom_batch_set_streams(self._h, streams))

    def profile(self, enable: bool) -> None:
        _check(self._lib, self._lib.pom_batch_profile(self._h, int(enable)))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int64()
        _check(self._lib, self._lib.pom_batch_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def launch_shape(self):
        """(envs per wavefront, lanes per env, launches per step)"""
        epw, lpe, parts = C.c_int32(), C.c_int32(), C.c_int32()
        _check(self._lib, self._lib.pom_batch_launch_shape(self._h, C.byref(epw), C.byref(lpe), C.byref(parts)))
        return epw.value, lpe.value, parts.value

    def issue_info(self):
        """(name of the issue mode in force for several-tick calls, streams their launches go to)"""
        mode, streams = C.c_int32(), C.c_int32()
        _check(self._lib, self._lib.pom_batch_issue_info(self._h, C.byref(mode), C.byref(streams)))
        return {ISSUE_DIRECT: "direct", ISSUE_THREADS: "threads", ISSUE_GRAPH: "graph", ISSUE_CHAIN: "chain"}.get(mode.value, "?"), streams.value

    def device_view(self):
        base, n_pad, rec = C.c_void_p(), C.c_int64(), C.c_int32()
        _check(self._lib, self._lib.pom_batch_device_view(self._h, C.byref(base), C.byref(n_pad), C.byref(rec)))
        return base.value, n_pad.value, rec.value


def bench_policy(codes, moves, first: int, count: int, tick: int, stream=None) -> None:
    """pom_bench_policy: the stand-in policy of the closed-loop measurements and tests — Move[4] of the envs [first, first + count) into
    `moves` (int32[n, 4] device tensor), from the POM_OBS_CODES observation `codes` (or None: from (env, agent, tick) alone), one launch on
    `stream`."""
    lib = load_library()
    st = getattr(stream, "cuda_stream", stream)
    _check(lib, lib.pom_bench_policy(codes.data_ptr() if codes is not None else None, moves.data_ptr(), int(first), int(count), int(tick) & 0xFFFFFFFF, st))

