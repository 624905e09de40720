#!/usr/bin/env python3
"""tests/golden/boardgen_stats.npz: the DISTRIBUTION of the reference's start boards, sampled from the compiled, unmodified
`InitBoardItems(state, seed)` (/root/reference/src/bboard/bboard.cpp:346-382, oracle/_ref/libpomref.so) over many seeds — the
pin for row f3 (the device board generator keeps the reference's distribution, not its libstdc++ / mt19937_64 stream).

Build container only; seeds: N fixed values spread over [0, 2^31).  The reference reads one queue slot PAST the woods it collected (`idxSample(0, q.count)`, bboard.cpp:367:
an uninitialised stack word used as a cell index), so every seed runs in a forked child; a child that dies is restarted behind
the seed that killed it.  A call that faults is caught inside the shim (the cell kinds are complete by then and are kept). A board on which that read landed on the board — a flag on a cell that is not wood, or fewer flags
than ceil(woods / 2) because the stray cell was counted — is counted as `stray` and kept out of the flag statistics; the
cell-kind statistics (drawn before the flag pass) use every board.

Recorded (data, not code): per-cell-kind totals, the histogram of woods per board, the histogram of (flags given - ceil(woods/2)),
the flag-value totals, per-cell wood / rigid frequency maps, the pair statistics of horizontally adjacent cells, and the counts of
crashed / stray seeds.
"""
import ctypes as C
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pomcpp_amd.state import STATE_DTYPE  # noqa: E402

N_SEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
OUT = os.path.join(ROOT, "tests", "golden", "boardgen_stats.npz")
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpomref.so"))
lib.ref_init_board_items.argtypes = [C.c_void_p, C.c_int]
lib.ref_init_board_items.restype = C.c_int


# the seeds: fixed, spread over the 31-bit range (consecutive small seeds of mt19937_64 give visibly correlated first draws)
SEEDS = np.random.default_rng(20261004).integers(0, 2**31 - 1, size=N_SEEDS, dtype=np.int64)


def child(first: int, last: int, wfd: int) -> None:
    s = np.zeros(1, dtype=STATE_DTYPE)
    with os.fdopen(wfd, "wb") as w:
        for k in range(first, last):
            faulted = lib.ref_init_board_items(s.ctypes.data, int(SEEDS[k]))
            w.write(struct.pack("<i", k if not faulted else -1 - k) + s["board"][0].astype("<i4").tobytes())
            w.flush()
            if faulted:
                os._exit(1)  # the fault may have been preceded by stray writes: a fresh process for the next seed
    os._exit(0)


boards, crashed, faulted = {}, [], set()
nxt = 0
while nxt < N_SEEDS:
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        child(nxt, N_SEEDS, w)
    os.close(w)
    last = nxt - 1
    with os.fdopen(r, "rb") as f:
        while True:
            rec = f.read(4 + 484)
            if len(rec) < 488:
                break
            seed = struct.unpack("<i", rec[:4])[0]
            if seed < 0:  # faulted inside the flag pass: the cell kinds are good
                seed = -1 - seed
                faulted.add(seed)
            boards[seed] = np.frombuffer(rec[4:], dtype="<i4").reshape(11, 11).copy()
            last = seed
    _, status = os.waitpid(pid, 0)
    if last in faulted:
        nxt = last + 1
    elif last + 1 < N_SEEDS:  # the child died on seed number last + 1 without reporting
        crashed.append(last + 1)
        nxt = last + 2
    else:
        nxt = N_SEEDS

seeds = np.array(sorted(boards), dtype=np.int64)
B = np.stack([boards[int(k)] for k in seeds])  # [n, 11, 11]
kind = np.where(B == 1, 1, np.where((B >> 8) == 2, 2, np.where(B == 0, 0, 3)))  # 0 passage, 1 rigid, 2 wood, 3 anything else
wood = kind == 2
flags = np.where(wood, B & 0xFF, 0)
n_wood = wood.sum(axis=(1, 2))
n_flag = (flags > 0).sum(axis=(1, 2))
want = (n_wood + 1) // 2
was_faulted = np.array([int(k) in faulted for k in seeds])
stray = (kind == 3).any(axis=(1, 2)) | (n_flag != want) | was_faulted
ok = ~stray
flag_hist = np.array([(flags[ok] == v).sum() for v in range(1, 5)], dtype=np.int64)
shortfall = np.bincount((want - n_flag)[~was_faulted].clip(0, 8), minlength=9)
pairs = np.zeros((3, 3), dtype=np.int64)  # horizontally adjacent cells (all boards; kind 3 folded into passage: the cell was drawn as one)
k2 = np.where(kind == 3, 0, kind)
for a in range(3):
    for b in range(3):
        pairs[a, b] = ((k2[:, :, :-1] == a) & (k2[:, :, 1:] == b)).sum()
inner = np.ones((11, 11), dtype=bool)
inner[0, 0] = inner[0, 10] = inner[10, 0] = inner[10, 10] = False  # where PutAgentsInCorners writes the agents afterwards
pairs_inner = np.zeros((3, 3), dtype=np.int64)  # the same over columns 1..9 only: no corner cell in any pair
for a in range(3):
    for b in range(3):
        pairs_inner[a, b] = ((k2[:, :, 1:9] == a) & (k2[:, :, 2:10] == b)).sum()
np.savez_compressed(
    OUT, n_seeds=np.int64(N_SEEDS), n_boards=np.int64(len(seeds)), crashed_seeds=np.array(crashed, dtype=np.int64),
    n_stray=np.int64(stray.sum()), n_faulted=np.int64(len(faulted)),
    kind_totals=np.array([(k2 == v).sum() for v in range(3)], dtype=np.int64),
    wood_per_cell=wood.sum(axis=0).astype(np.int64), rigid_per_cell=(kind == 1).sum(axis=0).astype(np.int64),
    woods_per_board_hist=np.bincount(n_wood, minlength=122).astype(np.int64),
    flag_value_totals=flag_hist, flags_given_ok=np.int64(n_flag[ok].sum()), flags_wanted_ok=np.int64(want[ok].sum()),
    n_ok=np.int64(ok.sum()), shortfall_hist=shortfall.astype(np.int64),
    flagged_per_cell_ok=(flags[ok] > 0).sum(axis=0).astype(np.int64), wood_per_cell_ok=wood[ok].sum(axis=0).astype(np.int64),
    adjacent_pairs=pairs, adjacent_pairs_inner=pairs_inner,
    woods_inner_hist=np.bincount((wood & inner).sum(axis=(1, 2)), minlength=118).astype(np.int64),
)
print(f"{len(seeds)} boards of {N_SEEDS} seeds ({len(crashed)} lost to a crash, {len(faulted)} faulted inside the flag pass: cell kinds kept), "
      f"{int(stray.sum())} with a stray flag draw")
print("cell kinds (passage, rigid, wood):", [(k2 == v).sum() / k2.size for v in range(3)], "expected 5/7 1/7 1/7 =", [5 / 7, 1 / 7, 1 / 7])
print("flag values 1..4 on clean boards:", flag_hist / flag_hist.sum(), " shortfall histogram:", shortfall.tolist())
print("written:", OUT, os.path.getsize(OUT), "bytes")
