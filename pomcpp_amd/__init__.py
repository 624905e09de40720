"""pomcpp_amd — host side of the MI355X batched Pommerman stepper.

Only what the `bboard::Step` hot path needs: the ctypes binding of the C-ABI
(`include/pom_batch.h`, built from `pomcpp_amd/csrc/` for gfx950), the numpy view
of the reference's 1004-byte `bboard::State`, a board generator with the
reference's cell distribution, and `BatchEnvironment`, a batch mirror of
`bboard::Environment` (/root/reference/include/bboard.hpp:541-644).

There is no CPU stepper in this package: without the HIP library every stepping
call raises.
"""
from .state import (  # noqa: F401
    STATE_DTYPE, AGENT_DTYPE, FLAME_DTYPE, Item, Move, Direction,
    new_states, put_agent, put_agents_in_corners, put_item, kill, plant_bomb,
    set_bomb_direction, bomb_x, bomb_y, bomb_id, bomb_strength, bomb_time, bomb_dir,
    is_flame, is_wood, is_powerup, is_agent, queue_get,
)
from .boards import make_boards  # noqa: F401
from .batch import (BatchEnvironment, PomError, library_path, load_library,  # noqa: F401
                    RESET_OFF, RESET_AT_START, RESET_AT_END)
