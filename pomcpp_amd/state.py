"""numpy view of the boundary State and the small host-side State helpers.

Layout and constants restate /root/reference/include/bboard.hpp (State :356-506,
AgentInfo :228-245, Flame :342-347, FixedQueue :115-188, Item :54-71, Move :35-43,
bomb bit fields :261-335); `include/pom_state.h` is the C twin.  The helpers
mirror the State methods the reference's tests use to set boards up
(src/bboard/bboard.cpp:120-146, 313-333) — none of them is on the device path.
"""
from __future__ import annotations

import numpy as np

BOARD_SIZE = 11
AGENT_COUNT = 4
BOMB_LIFETIME = 10
FLAME_LIFETIME = 4
MAX_BOMBS = 20


class Move:  # bboard.hpp:35-43
    IDLE, UP, DOWN, LEFT, RIGHT, BOMB = range(6)


class Direction:  # bboard.hpp:45-52
    IDLE, UP, DOWN, LEFT, RIGHT = range(5)


class Item:  # bboard.hpp:54-71
    PASSAGE = 0
    RIGID = 1
    WOOD = 2 << 8
    BOMB = 3
    FLAMES = 4 << 16
    FOG = 5
    EXTRABOMB = 6
    INCRRANGE = 7
    KICK = 8
    AGENT0 = 1 << 24
    AGENT1 = AGENT0 + 1
    AGENT2 = AGENT0 + 2
    AGENT3 = AGENT0 + 3


AGENT_DTYPE = np.dtype([
    ("x", "<i4"), ("y", "<i4"), ("bombCount", "<i4"), ("maxBombCount", "<i4"),
    ("bombStrength", "<i4"), ("canKick", "u1"), ("dead", "u1"), ("pad", "u1", (2,)),
])
FLAME_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("timeLeft", "<i4"), ("strength", "<i4")])
STATE_DTYPE = np.dtype([
    ("board", "<i4", (BOARD_SIZE, BOARD_SIZE)),  # [y][x]
    ("timeStep", "<i4"),
    ("aliveAgents", "<i4"),
    ("agents", AGENT_DTYPE, (AGENT_COUNT,)),
    ("bombs_queue", "<i4", (MAX_BOMBS,)),
    ("bombs_index", "<i4"),
    ("bombs_count", "<i4"),
    ("flames_queue", FLAME_DTYPE, (MAX_BOMBS,)),
    ("flames_index", "<i4"),
    ("flames_count", "<i4"),
])
assert STATE_DTYPE.itemsize == 1004
assert STATE_DTYPE.fields["agents"][1] == 492 and STATE_DTYPE.fields["bombs_queue"][1] == 588
assert STATE_DTYPE.fields["flames_queue"][1] == 676


def new_states(n: int = 1) -> np.ndarray:
    """n states as `std::make_unique<State>()` leaves them: zero, except aliveAgents=4,
    maxBombCount=1, bombStrength=1 and every flame slot's timeLeft=4 (bboard.hpp:235-236,345,370)."""
    s = np.zeros(n, dtype=STATE_DTYPE)
    s["aliveAgents"] = AGENT_COUNT
    s["agents"]["maxBombCount"] = 1
    s["agents"]["bombStrength"] = 1
    s["flames_queue"]["timeLeft"] = FLAME_LIFETIME
    return s


# ---- item predicates, bboard.hpp:73-109 -------------------------------------
def is_wood(v):
    return (np.asarray(v) >> 8) == 2


def is_powerup(v):
    v = np.asarray(v)
    return (v > 5) & (v < 9)


def is_flame(v):
    return (np.asarray(v) >> 16) == 4


def is_agent(v):
    return np.asarray(v) >= (1 << 24)


# ---- bomb bit fields, bboard.hpp:261-335 -------------------------------------
def bomb_x(b):
    return int(b) & 0xF


def bomb_y(b):
    return (int(b) >> 4) & 0xF


def bomb_id(b):
    return (int(b) >> 8) & 0xF


def bomb_strength(b):
    return (int(b) >> 12) & 0xF


def bomb_time(b):
    return (int(b) >> 16) & 0xF


def bomb_dir(b):
    return (int(b) >> 20) & 0xF


def _i32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def queue_get(s, which: str, offset: int):
    """FixedQueue::operator[] (bboard.hpp:180-187): the offset-th live element."""
    idx = (int(s[f"{which}_index"]) + offset) % MAX_BOMBS
    return s[f"{which}_queue"][idx]


# ---- State methods used for test / demo setup ----------------------------------
def put_item(s, x: int, y: int, item: int) -> None:  # bboard.hpp:460-463
    s["board"][y, x] = item


def put_agent(s, x: int, y: int, agent: int) -> None:  # bboard.cpp:313-320
    s["board"][y, x] = Item.AGENT0 + agent
    s["agents"][agent]["x"] = x
    s["agents"][agent]["y"] = y


def put_agents_in_corners(s, a0: int = 0, a1: int = 1, a2: int = 2, a3: int = 3) -> None:  # bboard.cpp:322-333
    e = BOARD_SIZE - 1
    s["board"][0, 0] = Item.AGENT0 + a0
    s["board"][0, e] = Item.AGENT0 + a1
    s["board"][e, e] = Item.AGENT0 + a2
    s["board"][e, 0] = Item.AGENT0 + a3
    s["agents"][a1]["x"] = e
    s["agents"][a2]["x"] = e
    s["agents"][a2]["y"] = e
    s["agents"][a3]["y"] = e


def kill(s, *agents: int) -> None:  # bboard.hpp:474-491
    for a in agents:
        if not s["agents"][a]["dead"]:
            s["agents"][a]["dead"] = 1
            s["aliveAgents"] -= 1


def plant_bomb(s, x: int, y: int, agent: int, set_item: bool = False, life_time: int = BOMB_LIFETIME) -> None:
    """State::PlantBomb / PlantBombModifiedLife (bboard.cpp:120-146): writes id, position,
    strength and time nibbles into the next queue slot and leaves its other bits alone."""
    ag = s["agents"][agent]
    if ag["bombCount"] >= ag["maxBombCount"]:
        return
    slot = (int(s["bombs_index"]) + int(s["bombs_count"])) % MAX_BOMBS
    b = int(s["bombs_queue"][slot]) & 0xFFFFFFFF
    b = ((b & ~0xF00) + (agent << 8)) & 0xFFFFFFFF
    b = ((b & ~0xFF) + x + (y << 4)) & 0xFFFFFFFF
    b = ((b & ~0xF000) + (int(ag["bombStrength"]) << 12)) & 0xFFFFFFFF
    b = ((b & ~0xF0000) + (life_time << 16)) & 0xFFFFFFFF
    s["bombs_queue"][slot] = _i32(b)
    if set_item:
        s["board"][y, x] = Item.BOMB
    ag["bombCount"] += 1
    s["bombs_count"] += 1


def set_bomb_direction(s, offset: int, direction: int) -> None:  # SetBombDirection, bboard.hpp:328-331
    slot = (int(s["bombs_index"]) + offset) % MAX_BOMBS
    b = int(s["bombs_queue"][slot]) & 0xFFFFFFFF
    s["bombs_queue"][slot] = _i32((b & ~0xF00000) + (direction << 20))
