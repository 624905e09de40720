"""The reference's `[step function]` test cases restated as backend-neutral scripts.

Each function below follows one TEST_CASE / SECTION of
/root/reference/unit_test/bboard/board_logic.cpp (lines cited per case) against a small `api`:

    s = api.make()                          std::make_unique<bboard::State>()
    api.corners / put_agent / put_item / kill / plant_bomb / set_bomb_direction / spawn_flame
    api.step(s, m) / api.several_steps(n, s, m)
    api.require(cond) / api.require_agent(s, agent, x, y)      REQUIRE / REQUIRE_AGENT (:11-17)

The same script runs (a) against the compiled reference when tests/golden/gen_golden.py records the
golden trace, (b) against the oracle on CPU and (c) against the HIP path through the C-ABI — so the
parity tests read like the reference's own.  `EXTRA_CASES` are directed vectors for the quirks of
SURVEY.md §9 that the reference's suite does not reach.
"""
from __future__ import annotations

from pomcpp_amd.state import Direction, Item, Move, bomb_x, bomb_y, is_flame, queue_get, BOMB_LIFETIME, FLAME_LIFETIME

IDLE = Move.IDLE


def idle():
    return [IDLE, IDLE, IDLE, IDLE]


def place_bombs_horizontally(api, s, agent, bombs):  # board_logic.cpp:34-46
    m = idle()
    for _ in range(bombs):
        m[agent] = Move.BOMB
        api.step(s, m)
        m[agent] = Move.RIGHT
        api.step(s, m)


# ---- TEST_CASE("Basic Non-Obstacle Movement") :55-83 -------------------------------------------
def basic_movement(api):
    s = api.make()
    api.corners(s, 0, 1, 2, 3)
    m = idle()
    m[0] = Move.RIGHT
    api.step(s, m)
    api.require_agent(s, 0, 1, 0)
    m[0] = Move.DOWN
    api.step(s, m)
    api.require_agent(s, 0, 1, 1)
    m[0] = Move.LEFT
    api.step(s, m)
    api.require_agent(s, 0, 0, 1)
    m[0] = Move.UP
    api.step(s, m)
    api.require_agent(s, 0, 0, 0)
    m[3] = Move.UP
    api.step(s, m)
    api.require_agent(s, 3, 0, 9)


# ---- TEST_CASE("Basic Obstacle Collision") :85-102 ----------------------------------------------
def obstacle_collision(api):
    s = api.make()
    api.corners(s, 0, 1, 2, 3)
    m = idle()
    api.put_item(s, 1, 0, Item.RIGID)
    m[0] = Move.RIGHT
    api.step(s, m)
    api.require_agent(s, 0, 0, 0)
    m[0] = Move.DOWN
    api.step(s, m)
    api.require_agent(s, 0, 0, 1)


# ---- TEST_CASE("Movement Against Flames") :104-119 ----------------------------------------------
def movement_against_flames(api):
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    api.spawn_flame(s, 1, 1, 2)
    m[0] = Move.RIGHT
    api.step(s, m)
    api.require(s["agents"][0, 0]["dead"])
    api.require(s["board"][0, 0, 0] == Item.PASSAGE)


# ---- TEST_CASE("Destination Collision") :121-171 ------------------------------------------------
def _dest_collision_setup(api):
    s = api.make()
    api.put_agent(s, 0, 1, 0)
    api.put_agent(s, 2, 1, 1)
    api.kill(s, 2, 3)
    return s, idle()


def dest_collision_two_agents(api):
    s, m = _dest_collision_setup(api)
    m[0], m[1] = Move.RIGHT, Move.LEFT
    api.step(s, m)
    api.require_agent(s, 0, 0, 1)
    api.require_agent(s, 1, 2, 1)


def dest_collision_dead(api):
    s, m = _dest_collision_setup(api)
    m[0], m[1] = Move.RIGHT, Move.LEFT
    api.kill(s, 1)
    api.step(s, m)
    api.require_agent(s, 0, 1, 1)


def dest_collision_four_agents(api):
    s, m = _dest_collision_setup(api)
    api.put_agent(s, 1, 0, 2)
    api.put_agent(s, 1, 2, 3)
    m[0], m[1], m[2], m[3] = Move.RIGHT, Move.LEFT, Move.DOWN, Move.UP
    api.step(s, m)
    api.require_agent(s, 0, 0, 1)
    api.require_agent(s, 1, 2, 1)
    api.require_agent(s, 2, 1, 0)
    api.require_agent(s, 3, 1, 2)


# ---- TEST_CASE("Movement Dependency Handling") :173-239 -----------------------------------------
def chain_against_obstacle(api):
    s = api.make()
    m = idle()
    for i in range(4):
        api.put_agent(s, i, 0, i)
    api.put_item(s, 4, 0, Item.RIGID)
    m[0] = m[1] = m[2] = m[3] = Move.RIGHT
    api.step(s, m)
    for i in range(4):
        api.require_agent(s, i, i, 0)


def two_on_one(api):  # :198-220 — triggers the reference's moves[-1] read (SURVEY Q-UB1)
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 0, 0)
    api.put_agent(s, 2, 0, 1)
    api.put_agent(s, 1, 0, 2)
    api.put_agent(s, 1, 1, 3)
    m[0], m[1] = Move.RIGHT, Move.LEFT
    m[2] = m[3] = Move.DOWN
    api.step(s, m)
    api.require_agent(s, 0, 0, 0)
    api.require_agent(s, 1, 2, 0)
    api.require_agent(s, 2, 1, 1)
    api.require_agent(s, 3, 1, 2)


def move_ouroboros(api):
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 0, 0)
    api.put_agent(s, 1, 0, 1)
    api.put_agent(s, 1, 1, 2)
    api.put_agent(s, 0, 1, 3)
    m[0], m[1], m[2], m[3] = Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP
    api.step(s, m)
    api.require_agent(s, 3, 0, 0)
    api.require_agent(s, 0, 1, 0)
    api.require_agent(s, 1, 1, 1)
    api.require_agent(s, 2, 0, 1)


# ---- TEST_CASE("Bomb Mechanics") :241-307 ---------------------------------------------------------
def standard_bomb_laying(api):
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    m[0] = Move.BOMB
    api.step(s, m)
    api.require(s["board"][0, 0, 0] == Item.AGENT0)
    m[0] = Move.DOWN
    api.step(s, m)
    api.require(s["board"][0, 0, 0] == Item.BOMB)


def bomb_block_simple(api):
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    api.plant_bomb(s, 1, 0, 0)
    m[0] = Move.RIGHT
    api.step(s, m)
    api.require_agent(s, 0, 0, 0)


def bomb_block_complex(api):
    s = api.make()
    m = idle()
    for i in range(4):
        api.put_agent(s, i, 0, i)
    m[0] = m[1] = m[2] = Move.RIGHT
    m[3] = Move.BOMB
    api.step(s, m)
    api.require_agent(s, 0, 0, 0)
    api.require_agent(s, 1, 1, 0)
    api.require_agent(s, 2, 2, 0)
    m[0] = m[1] = m[2] = IDLE
    m[3] = Move.RIGHT
    api.step(s, m)
    api.require_agent(s, 3, 4, 0)


def bomb_ouroboros_block(api):
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 0, 0)
    api.put_agent(s, 1, 0, 1)
    api.put_agent(s, 1, 1, 2)
    api.put_agent(s, 0, 1, 3)
    m[0] = m[1] = m[2] = m[3] = Move.BOMB
    api.step(s, m)
    m[0], m[1], m[2], m[3] = Move.RIGHT, Move.DOWN, Move.LEFT, Move.UP
    api.step(s, m)
    api.require_agent(s, 0, 0, 0)
    api.require_agent(s, 1, 1, 0)
    api.require_agent(s, 2, 1, 1)
    api.require_agent(s, 3, 0, 1)


# ---- TEST_CASE("Bomb Explosion") :310-382 -----------------------------------------------------------
def _explosion_setup(api):
    s = api.make()
    api.kill(s, 2, 3)
    api.put_agent(s, 5, 5, 0)
    return s, idle()


def bomb_goes_off(api):
    s, m = _explosion_setup(api)
    m[0] = Move.BOMB
    api.step(s, m)
    m[0] = Move.UP
    api.several_steps(BOMB_LIFETIME - 1, s, m)
    api.require(s["board"][0, 5, 5] == Item.BOMB)
    api.step(s, m)
    api.require(is_flame(s["board"][0, 5, 5]))


def destroy_objects_and_agents(api):
    s, m = _explosion_setup(api)
    api.put_item(s, 6, 5, Item.WOOD)
    api.put_agent(s, 4, 5, 1)
    m[0] = Move.BOMB
    api.step(s, m)
    m[0] = Move.UP
    api.several_steps(BOMB_LIFETIME, s, m)
    api.require(s["agents"][0, 1]["dead"])
    api.require(is_flame(s["board"][0, 5, 4]))
    api.require(is_flame(s["board"][0, 5, 6]))


def keep_rigid(api):
    s, m = _explosion_setup(api)
    api.put_item(s, 6, 5, Item.RIGID)
    m[0] = Move.BOMB
    api.step(s, m)
    m[0] = Move.UP
    api.several_steps(BOMB_LIFETIME, s, m)
    api.require(s["board"][0, 5, 6] == Item.RIGID)


def kill_only_one_wood(api):
    s, m = _explosion_setup(api)
    api.put_item(s, 7, 5, Item.WOOD)
    api.put_item(s, 8, 5, Item.WOOD)
    s["agents"][0, 0]["bombStrength"] = 5
    api.plant_bomb(s, 6, 5, 0, True)
    api.several_steps(BOMB_LIFETIME, s, m)
    api.require(is_flame(s["board"][0, 5, 7]))
    api.require(not is_flame(s["board"][0, 5, 8]))


def max_agent_bomb_limit(api):
    s, m = _explosion_setup(api)
    s["agents"][0, 0]["maxBombCount"] = 2
    api.require(s["agents"][0, 0]["bombCount"] == 0)
    place_bombs_horizontally(api, s, 0, 4)
    api.require(s["board"][0, 5, 5] == Item.BOMB)
    api.require(s["board"][0, 5, 6] == Item.BOMB)
    api.require(s["board"][0, 5, 7] == Item.PASSAGE)
    api.require(s["agents"][0, 0]["bombCount"] == 2)


# ---- TEST_CASE("Flame Mechanics") :384-427 ------------------------------------------------------------
def flame_lifetime(api):
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    api.spawn_flame(s, 5, 5, 4)
    api.step(s, m)
    api.several_steps(FLAME_LIFETIME - 2, s, m)
    api.require(is_flame(s["board"][0, 5, 5]))
    api.step(s, m)
    api.require(not is_flame(s["board"][0, 5, 5]))


def flame_vanish_completely(api):
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    api.spawn_flame(s, 5, 5, 4)
    api.step(s, m)
    for i in range(5):
        api.require(is_flame(s["board"][0, 5, 5 + i]))
        api.require(is_flame(s["board"][0, 5, 5 - i]))
        api.require(is_flame(s["board"][0, 5 + i, 5]))
        api.require(is_flame(s["board"][0, 5 - i, 5]))


def flame_only_vanish_own(api):
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    api.spawn_flame(s, 5, 5, 4)
    api.step(s, m)
    api.spawn_flame(s, 6, 6, 4)
    api.several_steps(FLAME_LIFETIME - 1, s, m)
    api.require(is_flame(s["board"][0, 5, 6]))
    api.require(is_flame(s["board"][0, 6, 5]))
    api.require(not is_flame(s["board"][0, 5, 5]))


# ---- TEST_CASE("Chained Explosions") :429-472 -----------------------------------------------------------
def chained_two_bombs(api):
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    api.plant_bomb(s, 5, 5, 0, True)
    api.step(s, m)
    api.plant_bomb(s, 4, 5, 1, True)
    api.several_steps(BOMB_LIFETIME - 1, s, m)
    api.require(s["bombs_count"][0] == 0)
    api.require(is_flame(s["board"][0, 5, 6]))


def chained_two_bombs_covered(api):
    s = api.make()
    m = idle()
    api.put_agent(s, 5, 5, 0)
    api.put_agent(s, 4, 5, 1)
    api.kill(s, 2, 3)
    m[0] = Move.BOMB
    api.step(s, m)
    m[1] = Move.BOMB
    api.step(s, m)
    m[0] = m[1] = Move.DOWN
    api.several_steps(BOMB_LIFETIME - 2, s, m)
    api.require(s["bombs_count"][0] == 2)
    api.step(s, m)
    api.require(s["bombs_count"][0] == 0)
    api.require(s["flames_count"][0] == 2)


# ---- TEST_CASE("Bomb Kick Mechanics") :474-634 ------------------------------------------------------------
def _kick_setup(api):
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 1, 0)
    s["agents"][0, 0]["canKick"] = 1
    api.plant_bomb(s, 1, 1, 0, True)
    s["agents"][0, 0]["maxBombCount"] = 5  # MAX_BOMBS_PER_AGENT
    m[0] = Move.RIGHT
    return s, m


def kick_one_agent_one_bomb(api):
    s, m = _kick_setup(api)
    api.kill(s, 1, 2, 3)
    api.step(s, m)
    api.require_agent(s, 0, 1, 1)
    api.require(s["board"][0, 1, 2] == Item.BOMB)
    for i in range(4):
        api.require(s["board"][0, 1, 2 + i] == Item.BOMB)
        api.step(s, m)
        m[0] = IDLE


def kick_against_flame(api):
    s, m = _kick_setup(api)
    api.kill(s, 1, 2, 3)
    api.put_item(s, 5, 1, Item.FLAMES)
    api.step(s, m)
    m[0] = IDLE
    api.several_steps(3, s, m)
    api.require(is_flame(s["board"][0, 1, 5]))
    api.require(s["bombs_count"][0] == 0)
    api.require(s["flames_count"][0] == 1)
    f = queue_get(s[0], "flames", 0)
    api.require(f["x"] == 5 and f["y"] == 1)


def kick_bomb_bomb_collision(api):
    s, m = _kick_setup(api)
    api.kill(s, 1, 2, 3)
    api.plant_bomb(s, 7, 7, 0, True)
    api.set_bomb_direction(s, 1, Direction.UP)
    for _ in range(6):
        api.step(s, m)
        m[0] = IDLE
    api.require(bomb_x(queue_get(s[0], "bombs", 0)) == 6)
    api.require(bomb_x(queue_get(s[0], "bombs", 1)) == 7)
    api.require(bomb_y(queue_get(s[0], "bombs", 1)) == 2)


def kick_bomb_bomb_static(api):
    s, m = _kick_setup(api)
    api.kill(s, 1, 2, 3)
    api.plant_bomb(s, 7, 6, 0, True)
    api.put_item(s, 7, 0, Item.WOOD)
    api.set_bomb_direction(s, 1, Direction.UP)
    for _ in range(7):
        api.step(s, m)
        m[0] = IDLE
    api.require(bomb_x(queue_get(s[0], "bombs", 0)) == 6)
    api.require(bomb_x(queue_get(s[0], "bombs", 1)) == 7)
    api.require(bomb_y(queue_get(s[0], "bombs", 1)) == 1)


def kick_bounce_back_agent(api):
    s, m = _kick_setup(api)
    api.kill(s, 2, 3)
    api.put_agent(s, 0, 2, 1)
    m[1] = Move.UP
    api.plant_bomb(s, 2, 2, 0, True)
    api.set_bomb_direction(s, 1, Direction.UP)
    api.step(s, m)
    api.require_agent(s, 0, 0, 1)
    api.require_agent(s, 1, 0, 2)
    api.require(bomb_x(queue_get(s[0], "bombs", 0)) == 1)
    api.require(bomb_x(queue_get(s[0], "bombs", 1)) == 2)


def kick_bounce_back_complex_chain(api):
    s, m = _kick_setup(api)
    api.kill(s, 2, 3)
    api.put_agent(s, 0, 2, 1)
    m[1] = Move.UP
    api.plant_bomb(s, 2, 2, 0, True)
    api.plant_bomb(s, 0, 3, 0, True)
    api.set_bomb_direction(s, 1, Direction.UP)
    api.set_bomb_direction(s, 2, Direction.UP)
    api.step(s, m)
    api.require_agent(s, 0, 0, 1)
    api.require_agent(s, 1, 0, 2)
    api.require(s["board"][0, 3, 0] == Item.BOMB)
    api.require(s["board"][0, 1, 1] == Item.BOMB)
    api.require(s["board"][0, 2, 2] == Item.BOMB)


def kick_bounce_back_super_complex_chain(api):  # :581-600, no assertions in the reference: trace only
    s, m = _kick_setup(api)
    api.kill(s, 3)
    api.put_agent(s, 0, 2, 1)
    api.put_agent(s, 1, 3, 2)
    api.put_item(s, 2, 1, Item.RIGID)
    m[1] = Move.UP
    m[2] = Move.BOMB
    api.plant_bomb(s, 0, 3, 0, True)
    api.set_bomb_direction(s, 1, Direction.UP)
    for _ in range(3):
        api.step(s, m)
        m[0] = m[1] = IDLE
        m[2] = Move.LEFT


def kick_bounce_back_wall(api):
    s, m = _kick_setup(api)
    api.kill(s, 1, 3)
    api.put_agent(s, 1, 3, 2)
    api.put_item(s, 2, 1, Item.RIGID)
    m[2] = Move.LEFT
    s["agents"][0, 2]["canKick"] = 1
    api.plant_bomb(s, 0, 3, 0, True)
    api.step(s, m)
    api.require_agent(s, 2, 1, 3)
    api.require(s["board"][0, 3, 0] == Item.BOMB)


def kick_stepping_on_bombs(api):
    s, m = _kick_setup(api)
    api.put_agent(s, 6, 3, 0)
    api.put_agent(s, 6, 4, 1)
    api.put_agent(s, 6, 5, 2)
    m[0] = m[1] = m[2] = IDLE
    api.plant_bomb(s, 5, 6, 3, True)
    api.plant_bomb(s, 6, 6, 2, True)
    api.put_agent(s, 6, 6, 3)
    m[3] = IDLE
    api.step(s, m)
    api.require_agent(s, 3, 6, 6)
    m[3] = Move.LEFT
    api.step(s, m)
    api.require_agent(s, 3, 6, 6)


CASES = {  # the 32 leaf runs of [step function]
    "basic_movement": basic_movement,
    "obstacle_collision": obstacle_collision,
    "movement_against_flames": movement_against_flames,
    "dest_collision_two_agents": dest_collision_two_agents,
    "dest_collision_dead": dest_collision_dead,
    "dest_collision_four_agents": dest_collision_four_agents,
    "chain_against_obstacle": chain_against_obstacle,
    "two_on_one": two_on_one,
    "move_ouroboros": move_ouroboros,
    "standard_bomb_laying": standard_bomb_laying,
    "bomb_block_simple": bomb_block_simple,
    "bomb_block_complex": bomb_block_complex,
    "bomb_ouroboros_block": bomb_ouroboros_block,
    "bomb_goes_off": bomb_goes_off,
    "destroy_objects_and_agents": destroy_objects_and_agents,
    "keep_rigid": keep_rigid,
    "kill_only_one_wood": kill_only_one_wood,
    "max_agent_bomb_limit": max_agent_bomb_limit,
    "flame_lifetime": flame_lifetime,
    "flame_vanish_completely": flame_vanish_completely,
    "flame_only_vanish_own": flame_only_vanish_own,
    "chained_two_bombs": chained_two_bombs,
    "chained_two_bombs_covered": chained_two_bombs_covered,
    "kick_one_agent_one_bomb": kick_one_agent_one_bomb,
    "kick_against_flame": kick_against_flame,
    "kick_bomb_bomb_collision": kick_bomb_bomb_collision,
    "kick_bomb_bomb_static": kick_bomb_bomb_static,
    "kick_bounce_back_agent": kick_bounce_back_agent,
    "kick_bounce_back_complex_chain": kick_bounce_back_complex_chain,
    "kick_bounce_back_super_complex_chain": kick_bounce_back_super_complex_chain,
    "kick_bounce_back_wall": kick_bounce_back_wall,
    "kick_stepping_on_bombs": kick_stepping_on_bombs,
}


# ---- directed vectors for SURVEY.md §9 quirks -------------------------------------------------------------
def q1_stale_slot_direction_inherited(api):
    """Q1: a bomb planted into a slot vacated by RemoveAt inherits that slot's stale direction nibble."""
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 9, 0)
    api.put_agent(s, 5, 5, 1)
    api.kill(s, 2, 3)
    s["agents"][0, 0]["maxBombCount"] = 3
    api.put_item(s, 8, 2, Item.FLAMES)          # an orphan flame cell for bomb 1 to run into
    api.plant_bomb(s, 2, 0, 0, True)            # slot 0: stays
    api.plant_bomb(s, 5, 2, 0, True)            # slot 1: moving RIGHT towards the flame cell
    api.set_bomb_direction(s, 1, Direction.RIGHT)
    api.several_steps(3, s, m)                  # reaches (8,2), explodes, RemoveAt(1) leaves a stale copy in slot 1
    api.require(s["bombs_count"][0] == 1)
    m[0] = Move.BOMB
    api.step(s, m)                              # planted into slot 1: inherits dir RIGHT and moves this very tick
    api.require(s["bombs_count"][0] == 2)
    api.require(bomb_x(queue_get(s[0], "bombs", 1)) == 1)


def q2_stale_index_after_nested_chain(api):
    """Q2: ExplodeBombAt(i) removes whatever sits at i after the nested chain shrank the queue."""
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 0, 0)
    api.put_agent(s, 10, 10, 1)
    api.kill(s, 2, 3)
    s["agents"][0, 0]["maxBombCount"] = 5
    s["agents"][0, 0]["bombStrength"] = 3
    s["agents"][0, 1]["maxBombCount"] = 5
    s["agents"][0, 1]["bombStrength"] = 3
    api.put_item(s, 8, 5, Item.FLAMES)
    api.plant_bomb(s, 3, 5, 0, True)            # index 0, reached by the chain of the kicked bomb
    api.plant_bomb(s, 5, 8, 1, True)            # index 1, bystander
    api.plant_bomb(s, 5, 5, 0, True)            # index 2: chained by bomb 3's flame, itself chains index 0
    api.plant_bomb(s, 6, 5, 1, True)            # index 3: kicked RIGHT into the flame cell
    api.set_bomb_direction(s, 3, Direction.RIGHT)
    api.several_steps(4, s, m)


def q9_dead_agent_cancels_move(api):
    """Q9: a dead agent's stale position and its Move entry still cancel a live agent's move as a 'switch'."""
    s = api.make()
    m = idle()
    api.put_agent(s, 1, 1, 0)
    api.put_agent(s, 2, 1, 1)
    api.kill(s, 1, 2, 3)
    api.put_item(s, 2, 1, Item.PASSAGE)
    m[0], m[1] = Move.RIGHT, Move.LEFT
    api.step(s, m)
    api.require_agent(s, 0, 1, 1)               # stayed
    m[1] = IDLE
    api.step(s, m)
    api.require_agent(s, 0, 2, 1)               # now moves


def q10_three_cycle_plus_one(api):
    """Q10: no roots, but not a 4-cycle: dependency[] is overwritten and one agent is lost (Q-UB1)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 1, 1, 0)
    api.put_agent(s, 2, 1, 1)
    api.put_agent(s, 2, 2, 2)
    api.put_agent(s, 0, 1, 3)
    m[0], m[1], m[2], m[3] = Move.RIGHT, Move.DOWN, Move.UP, Move.RIGHT
    api.step(s, m)
    api.step(s, m)


def q7_out_of_order_timer_underflow(api):
    """Q7: only the queue top is tested for time 0; a later bomb at 0 borrows from its direction nibble."""
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    s["agents"][0, 0]["maxBombCount"] = 3
    api.plant_bomb(s, 5, 5, 0, True, 6)
    api.plant_bomb(s, 7, 7, 0, True, 2)
    api.several_steps(8, s, m)


def q13_resting_bomb_on_item_cells(api):
    """Loop B for resting bombs (step.cpp:243-272 with target == position): PASSAGE under the bomb becomes BOMB, a power-up
    under it is a static block (the bomb is only set to rest) and stays, two bombs on one cell: only the later one looks."""
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    for a in range(4):
        s["agents"][0, a]["maxBombCount"] = 5
    api.plant_bomb(s, 3, 3, 0, False)            # no item set: the cell shows PASSAGE
    api.plant_bomb(s, 5, 5, 1, False)
    api.put_item(s, 5, 5, Item.EXTRABOMB)        # a power-up under a queued bomb
    api.plant_bomb(s, 7, 7, 2, False)
    api.plant_bomb(s, 7, 7, 3, False)            # shares the cell with the bomb before it
    api.step(s, m)
    api.require(s["board"][0, 3, 3] == Item.BOMB)
    api.require(s["board"][0, 5, 5] == Item.EXTRABOMB)
    api.require(s["board"][0, 7, 7] == Item.BOMB)
    api.set_bomb_direction(s, 0, Direction.RIGHT)   # one moving bomb: the general loop, same resting bombs
    api.several_steps(3, s, m)


def q14_agent_chain_with_planting(api):
    """Three agents in a row all step RIGHT (a dependency chain of depth 2: visited root first), the last one plants; the
    fourth walks into the tail's old cell one tick later.  Queue order of planters follows the visiting order."""
    s = api.make()
    m = idle()
    api.put_agent(s, 3, 5, 2)
    api.put_agent(s, 2, 5, 0)
    api.put_agent(s, 1, 5, 3)
    api.put_agent(s, 9, 9, 1)
    for a in range(4):
        s["agents"][0, a]["maxBombCount"] = 3
    m[0], m[2], m[3], m[1] = Move.RIGHT, Move.RIGHT, Move.RIGHT, Move.BOMB
    api.step(s, m)
    api.require_agent(s, 2, 4, 5)
    api.require_agent(s, 0, 3, 5)
    api.require_agent(s, 3, 2, 5)
    m[3], m[0] = Move.BOMB, Move.BOMB            # planters 3 and 0 in one tick: slots in visiting order
    api.step(s, m)
    m[0], m[2], m[3] = Move.LEFT, Move.LEFT, Move.IDLE   # 2 waits for 0's cell, 0 is blocked by 3: nobody moves
    api.step(s, m)
    m[3] = Move.UP                               # 3 leaves: 0 follows into his cell, 2 into 0's
    api.several_steps(2, s, m)


def q15_long_blast_chain(api):
    """Strength-4 blasts: a ray that burns wood, one stopped by rigid, agents killed on the way, a chain through three bombs
    two of which share a cell, picked up mid-ray after the nested explosions (SpawnFlameItem's tail)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 5, 1, 0)
    api.put_agent(s, 9, 5, 1)
    api.put_agent(s, 5, 9, 2)
    api.put_agent(s, 0, 0, 3)
    for a in range(4):
        s["agents"][0, a]["maxBombCount"] = 5
        s["agents"][0, a]["bombStrength"] = 4
    api.put_item(s, 2, 5, Item.WOOD + 2)
    api.put_item(s, 5, 7, Item.RIGID)
    api.plant_bomb(s, 5, 5, 0, True, 2)          # goes off first
    api.plant_bomb(s, 7, 5, 1, True, 9)          # on its +x ray
    api.plant_bomb(s, 7, 5, 2, True, 9)          # same cell: stays queued, sitting in the flame
    api.plant_bomb(s, 7, 3, 3, True, 9)          # reached by the nested blast's -y ray
    api.plant_bomb(s, 5, 3, 3, True, 9)          # on the first bomb's -y ray, after the nested ones have run
    api.several_steps(4, s, m)


def full_queues_stress(api):
    """20 live bombs (queue full) detonating in one chain: full-depth explosion stack, flame queue fills."""
    s = api.make()
    m = idle()
    api.corners(s, 0, 1, 2, 3)
    for a in range(4):
        s["agents"][0, a]["maxBombCount"] = 5
        s["agents"][0, a]["bombStrength"] = 2
    k = 0
    for y in (2, 4, 6, 8):
        for x in (1, 3, 5, 7, 9):
            api.plant_bomb(s, x, y, k % 4, True, 3 if k == 0 else 9)
            k += 1
    api.require(s["bombs_count"][0] == 20)
    m[0] = Move.BOMB                             # a 21st bomb is refused: maxBombCount reached
    api.several_steps(6, s, m)


def q3_current_vs_stored_strength(api):
    """Q3: ExplodeBombAt (a chained bomb) burns with its owner's CURRENT bombStrength (bboard.cpp:115), ExplodeTopBomb with the
    strength stored in the bomb word (bboard.cpp:194)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 0, 0)
    api.put_agent(s, 10, 10, 1)
    api.kill(s, 2, 3)
    s["agents"][0, 0]["maxBombCount"] = 5
    s["agents"][0, 1]["maxBombCount"] = 5
    api.plant_bomb(s, 2, 3, 1, True, 2)          # head of the queue, strength 1: goes off in the 2nd step
    api.plant_bomb(s, 3, 3, 0, True, 9)          # stored strength 1, on the first bomb's +x ray
    s["agents"][0, 0]["bombStrength"] = 3        # the owner collects range afterwards
    api.plant_bomb(s, 8, 8, 0, True, 4)          # stored strength 3
    api.several_steps(2, s, m)
    api.require(is_flame(s["board"][0, 3, 6]))   # the chained bomb burnt 3 cells: the owner's current strength
    api.require(is_flame(s["board"][0, 6, 3]))
    s["agents"][0, 0]["bombStrength"] = 1        # ... and loses it again (not reachable in play, legal as an input state)
    api.several_steps(2, s, m)
    api.require(is_flame(s["board"][0, 8, 5]))   # the head of the queue burns with the strength it was planted with
    api.require(is_flame(s["board"][0, 5, 8]))
    api.several_steps(5, s, m)


def q4_flame_overwrite_and_pop_rules(api):
    """Q4: a flame takes every non-rigid cell; only WOOD keeps its flag; a power-up item and the revealed-power-up flag of an
    older flame are destroyed; a chained bomb's own cell ends with the OUTER blast's signature; PopFlame clears exactly the
    cells that still carry its own id (bboard.cpp:24-57,148-180,207,218)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 0, 0)
    api.put_agent(s, 10, 10, 1)
    api.kill(s, 2, 3)
    for a in (0, 1):
        s["agents"][0, a]["maxBombCount"] = 5
        s["agents"][0, a]["bombStrength"] = 2
    api.put_item(s, 6, 5, Item.EXTRABOMB)        # a power-up on the +x ray: burnt
    api.put_item(s, 4, 5, Item.WOOD + 1)         # flagged wood on the -x ray: reveals its power-up when the flame pops
    api.put_item(s, 5, 3, Item.WOOD + 2)         # flagged wood on the -y ray, burnt again by the second blast one step later
    api.plant_bomb(s, 5, 5, 0, True, 1)          # A
    api.plant_bomb(s, 3, 3, 0, True, 2)          # E: its +x ray crosses (5,3) while that cell is A's flame with flag 2
    api.plant_bomb(s, 5, 7, 1, True, 9)          # B: chained by A's +y ray, its -y ray runs back over A's cells
    api.step(s, m)
    api.require(is_flame(s["board"][0, 5, 6]))
    api.require(is_flame(s["board"][0, 7, 5]))
    api.require(s["bombs_count"][0] == 1)
    api.step(s, m)
    api.require(s["bombs_count"][0] == 0)
    api.several_steps(4, s, m)
    api.require(s["board"][0, 5, 4] == Item.EXTRABOMB)   # the wood's flag survived its one flame
    api.require(s["board"][0, 5, 6] == Item.PASSAGE)     # the power-up item did not
    api.require(s["board"][0, 3, 5] == Item.PASSAGE)     # a flag inside a flame does not survive the next flame
    api.require(s["flames_count"][0] == 0)


def q5_ray_order_and_second_bomb_on_origin(api):
    """Q5: rays run +x, -x, +y, -y (the flame queue records the order of the chained bombs); wood stops a ray after burning,
    rigid before; agents on the way die and the ray goes on; the blast kills the agent standing on its origin but does not set
    off a second bomb queued on the same cell — that one goes off a step later, in loop B (bboard.cpp:198-263)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 0, 0, 0)
    api.put_agent(s, 10, 10, 1)
    s["agents"][0, 0]["maxBombCount"] = 5
    s["agents"][0, 0]["bombStrength"] = 3
    s["agents"][0, 1]["maxBombCount"] = 5
    api.plant_bomb(s, 5, 5, 0, True, 1)          # A, strength 3
    api.plant_bomb(s, 5, 5, 1, True, 9)          # a second bomb on A's cell
    api.plant_bomb(s, 4, 5, 1, True, 9)          # on the -x ray
    api.plant_bomb(s, 6, 5, 1, True, 9)          # on the +x ray: chained first
    api.put_agent(s, 5, 5, 2)                    # stands on the two bombs
    api.put_agent(s, 5, 3, 3)                    # on the -y ray, two cells out: dies, the ray goes on to (5,2)
    api.put_item(s, 2, 5, Item.RIGID)
    api.put_item(s, 5, 7, Item.WOOD)
    api.step(s, m)
    api.require(s["agents"][0, 2]["dead"] == 1)
    api.require(s["agents"][0, 3]["dead"] == 1)
    api.require(is_flame(s["board"][0, 2, 5]))
    api.require(is_flame(s["board"][0, 7, 5]))           # the wood burnt ...
    api.require(s["board"][0, 8, 5] == Item.PASSAGE)     # ... and stopped the ray
    api.require(s["bombs_count"][0] == 1)                # the second bomb on the origin is still queued
    f0, f1, f2 = (queue_get(s[0], "flames", k) for k in range(3))
    api.require((int(f0["x"]), int(f1["x"]), int(f2["x"])) == (5, 6, 4))   # A, then +x's bomb, then -x's
    api.step(s, m)
    api.require(s["bombs_count"][0] == 0)                # it sat in a flame cell: loop B set it off
    api.several_steps(5, s, m)


def q6_flame_timing(api):
    """Q6: timeLeft 4 at spawn, decremented at the start of every Step and popped at 0: a flame kills movers during the three
    Steps after it was spawned and is gone for the fourth; TickFlames only ever tests the head (step_utility.cpp:208-222)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 5, 9, 0)
    api.put_agent(s, 8, 5, 1)
    api.kill(s, 2, 3)
    api.spawn_flame(s, 5, 5, 2)
    m[0] = Move.UP
    api.step(s, m)
    api.require_agent(s, 0, 5, 8)
    api.spawn_flame(s, 2, 2, 1)                  # one Step younger
    m[0] = IDLE
    api.step(s, m)
    m[0] = Move.UP
    api.step(s, m)                               # third Step after the spawn: (5,7) still burns
    api.require(s["agents"][0, 0]["dead"] == 1)
    m[0], m[1] = IDLE, Move.LEFT
    api.step(s, m)                               # fourth: popped before the agents move
    api.require_agent(s, 1, 7, 5)
    api.require(is_flame(s["board"][0, 2, 2]))
    api.require(s["flames_count"][0] == 1)
    m[1] = IDLE
    api.step(s, m)
    api.require(s["board"][0, 2, 2] == Item.PASSAGE)
    api.require(s["flames_count"][0] == 0)


def q8_several_bombs_on_one_cell(api):
    """Q8: planting neither puts Item::BOMB on the board (the agent's item stays) nor looks for a bomb already there; every
    position lookup answers with the FIRST queue match: a kick moves only the first of two bombs on a cell
    (step.cpp:54, bboard.cpp:265-311)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 3, 3, 0)
    api.put_agent(s, 3, 5, 1)
    api.kill(s, 2, 3)
    s["agents"][0, 0]["maxBombCount"] = 3
    s["agents"][0, 1]["canKick"] = 1
    m[0] = Move.BOMB
    api.step(s, m)
    api.step(s, m)
    api.require(s["bombs_count"][0] == 2)
    api.require(s["board"][0, 3, 3] == Item.AGENT0)
    m[0], m[1] = Move.RIGHT, Move.UP
    api.step(s, m)
    api.require(s["board"][0, 3, 3] == Item.BOMB)
    api.step(s, m)                               # agent 1 kicks: only the first bomb leaves ...
    api.require(bomb_y(queue_get(s[0], "bombs", 0)) == 2)
    api.require(bomb_y(queue_get(s[0], "bombs", 1)) == 3)
    api.require_agent(s, 1, 3, 4)                # ... and the second, resting under him now, bounces him back in loop A
    api.require(s["board"][0, 3, 3] == Item.BOMB)
    m[0], m[1] = IDLE, IDLE
    api.several_steps(9, s, m)                   # both go off


def q11_dying_mover_vacates(api):
    """Q11: a mover that walks into a flame dies; the cell it leaves becomes BOMB if a bomb is queued there, else PASSAGE, and
    only if it still shows the mover's own item; the dead agent keeps its stale position (step.cpp:84-99,125-136)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 3, 3, 0)
    api.put_agent(s, 6, 4, 1)
    api.put_agent(s, 5, 5, 2)
    api.put_agent(s, 10, 10, 3)
    api.put_item(s, 5, 5, Item.WOOD)             # agent 2's cell shows something else
    m[0] = Move.BOMB
    api.step(s, m)
    api.spawn_flame(s, 5, 3, 1)
    m[0], m[1], m[2] = Move.RIGHT, Move.UP, Move.UP
    api.step(s, m)
    for a in (0, 1, 2):
        api.require(s["agents"][0, a]["dead"] == 1)
    api.require(s["aliveAgents"][0] == 1)
    api.require(s["board"][0, 3, 3] == Item.BOMB)
    api.require(s["board"][0, 4, 6] == Item.PASSAGE)
    api.require(s["board"][0, 5, 5] == Item.WOOD)
    api.require((int(s["agents"][0, 0]["x"]), int(s["agents"][0, 0]["y"])) == (3, 3))
    m[0], m[1], m[2] = IDLE, IDLE, IDLE
    api.several_steps(2, s, m)


def q12_kick_bounce_collision(api):
    """Q12: anybody may step onto a BOMB cell; loop A bounces a non-kicker back (the resting bomb's target, its own cell, now
    holds an agent), a kicker gives the bomb his direction and it moves in the same Step; two bombs that want the same cell
    both stop, and ResolveBombCollision bounces the kicker back — but only if his bomb comes FIRST in the queue: the tests only
    look at queue indices >= the current one; they compare bomb VALUES, so two identical words never collide with each other
    (step.cpp:147-184,195-278, step_utility.cpp:62-128,279-329)."""
    s = api.make()
    m = idle()
    api.put_agent(s, 2, 2, 0)
    api.put_agent(s, 2, 6, 1)
    api.put_agent(s, 4, 8, 2)
    api.put_agent(s, 4, 10, 3)
    for a in (1, 2, 3):
        s["agents"][0, a]["canKick"] = 1
    s["agents"][0, 0]["maxBombCount"] = 9
    api.plant_bomb(s, 3, 2, 0, True)             # 0: a non-kicker steps onto it
    api.plant_bomb(s, 3, 6, 0, True)             # 1: kicked RIGHT
    api.plant_bomb(s, 5, 8, 0, True)             # 2: kicked RIGHT, towards (6,8) ...
    api.plant_bomb(s, 7, 8, 0, True)             # 3: ... where this one is heading: the kicker is bounced
    api.set_bomb_direction(s, 3, Direction.LEFT)
    api.plant_bomb(s, 7, 10, 0, True)            # 4: heading for (6,10) ...
    api.set_bomb_direction(s, 4, Direction.LEFT)
    api.plant_bomb(s, 5, 10, 0, True)            # 5: ... as is this one once kicked: later in the queue, its kicker stays
    api.plant_bomb(s, 8, 3, 0, True)             # 6 and 7: identical words on one cell
    api.plant_bomb(s, 8, 3, 0, True)
    m[0], m[1], m[2], m[3] = Move.RIGHT, Move.RIGHT, Move.RIGHT, Move.RIGHT
    api.step(s, m)
    api.require_agent(s, 0, 2, 2)                # bounced back
    api.require(s["board"][0, 2, 3] == Item.BOMB)
    api.require_agent(s, 1, 3, 6)                # stands where the bomb was
    api.require(bomb_x(queue_get(s[0], "bombs", 1)) == 4)
    api.require_agent(s, 2, 4, 8)                # his kick collided: bounced back by ResolveBombCollision
    api.require(bomb_x(queue_get(s[0], "bombs", 2)) == 5)
    api.require(bomb_x(queue_get(s[0], "bombs", 3)) == 7)
    api.require_agent(s, 3, 5, 10)               # his did too, but the collision was found at the other bomb's turn
    api.require(bomb_x(queue_get(s[0], "bombs", 4)) == 7)
    api.require(bomb_x(queue_get(s[0], "bombs", 5)) == 5)
    m[0], m[1], m[2], m[3] = IDLE, IDLE, IDLE, IDLE
    api.several_steps(3, s, m)


EXTRA_CASES = {
    "q1_stale_slot_direction_inherited": q1_stale_slot_direction_inherited,
    "q2_stale_index_after_nested_chain": q2_stale_index_after_nested_chain,
    "q3_current_vs_stored_strength": q3_current_vs_stored_strength,
    "q4_flame_overwrite_and_pop_rules": q4_flame_overwrite_and_pop_rules,
    "q5_ray_order_and_second_bomb_on_origin": q5_ray_order_and_second_bomb_on_origin,
    "q6_flame_timing": q6_flame_timing,
    "q7_out_of_order_timer_underflow": q7_out_of_order_timer_underflow,
    "q8_several_bombs_on_one_cell": q8_several_bombs_on_one_cell,
    "q9_dead_agent_cancels_move": q9_dead_agent_cancels_move,
    "q10_three_cycle_plus_one": q10_three_cycle_plus_one,
    "q11_dying_mover_vacates": q11_dying_mover_vacates,
    "q12_kick_bounce_collision": q12_kick_bounce_collision,
    "q13_resting_bomb_on_item_cells": q13_resting_bomb_on_item_cells,
    "q14_agent_chain_with_planting": q14_agent_chain_with_planting,
    "q15_long_blast_chain": q15_long_blast_chain,
    "full_queues_stress": full_queues_stress,
}

ALL_CASES = {**CASES, **EXTRA_CASES}
