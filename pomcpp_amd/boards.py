"""Synthetic start boards (host side, numpy, vectorised over envs).

The reference's InitBoardItems (/root/reference/src/bboard/bboard.cpp:346-382) draws from
std::mt19937_64 through libstdc++'s uniform_int_distribution and reads an uninitialised
queue slot; only its *distribution* is reproduced here, not its stream:
every cell i.i.d. passage 5/7, rigid 1/7, wood 1/7 (bboard.cpp:59-74,349,357-358),
ceil(woods/2) wood cells get a flag uniform in 1..4 (:367-381), then agents 0..3 are
written over the corners without clearing around them (:322-333).

kind="stress" is the kick / chain-explosion workload of SURVEY.md §8d config 5: sparse
obstacles (rigid 1/14, wood 1/14), every agent canKick, maxBombCount 5, bombStrength 4,
up to 8 pre-planted bombs with non-decreasing timers along the queue, two of them moving.
"""
from __future__ import annotations

import numpy as np

from .state import Item, MAX_BOMBS, new_states


def make_boards(n: int, seed: int = 1, kind: str = "ffa") -> np.ndarray:
    if kind not in ("ffa", "stress"):
        raise ValueError(f"unknown board kind {kind!r}")
    rng = np.random.Generator(np.random.PCG64(seed))
    s = new_states(n)
    draws = rng.integers(0, 14 if kind == "stress" else 7, size=(n, 121))
    cells = np.zeros((n, 121), dtype=np.int32)
    cells[draws == 1] = Item.RIGID
    wood = draws == 2
    cells[wood] = Item.WOOD
    # ceil(woods/2) of each env's woods, chosen uniformly, carry a flag 1..4
    keys = np.where(wood, rng.random((n, 121)), 2.0)
    order = np.argsort(keys, axis=1)
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.broadcast_to(np.arange(121), (n, 121)), axis=1)
    nwood = wood.sum(axis=1, keepdims=True)
    flagged = wood & (rank < (nwood + 1) // 2)
    cells += np.where(flagged, rng.integers(1, 5, size=(n, 121)), 0).astype(np.int32)
    s["board"] = cells.reshape(n, 11, 11)
    # PutAgentsInCorners(0, 1, 2, 3)
    s["board"][:, 0, 0] = Item.AGENT0
    s["board"][:, 0, 10] = Item.AGENT0 + 1
    s["board"][:, 10, 10] = Item.AGENT0 + 2
    s["board"][:, 10, 0] = Item.AGENT0 + 3
    s["agents"]["x"][:, 1] = 10
    s["agents"]["x"][:, 2] = 10
    s["agents"]["y"][:, 2] = 10
    s["agents"]["y"][:, 3] = 10
    if kind == "stress":
        s["agents"]["canKick"] = 1
        s["agents"]["maxBombCount"] = 5
        s["agents"]["bombStrength"] = 4
        life = np.full(n, 2, dtype=np.int64)
        env = np.arange(n)
        for k in range(8):
            x = rng.integers(0, 11, size=n)
            y = rng.integers(0, 11, size=n)
            life = np.minimum(life + rng.integers(0, 2, size=n), 10)
            direction = np.where(k < 2, rng.integers(1, 5, size=n), 0)
            ok = s["board"][env, y, x] == Item.PASSAGE
            cnt = s["bombs_count"].astype(np.int64)
            slot = cnt % MAX_BOMBS
            word = x + (y << 4) + ((k & 3) << 8) + (4 << 12) + (life << 16) + (direction << 20)
            q = s["bombs_queue"]
            q[env[ok], slot[ok]] = word[ok].astype(np.int32)
            s["board"][env[ok], y[ok], x[ok]] = Item.BOMB
            s["bombs_count"][ok] += 1
            bc = s["agents"]["bombCount"]
            bc[env[ok], k & 3] += 1
    return s
