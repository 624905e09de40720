"""TEST INFRASTRUCTURE — CPU restatement of the observation export, numpy.  Only tests/, smoke() and bench's
cpu_baseline leg may import this; the product (pomcpp_amd/, libpom_batch.so) never does.

The reference has no observation function ("parity unpinned" against reference OUTPUTS for this row): the planes restate what
its agents read off a State, and each rule below cites the reference definition it follows.  Input: boundary States
(pomcpp_amd.state.STATE_DTYPE, the reference's 1004-byte layout, bboard.hpp:356-506).
"""
import numpy as np

PLANES = 16
N = 11
FLAMES = 4 << 16   # Item::FLAMES, bboard.hpp:61
AGENT0 = 1 << 24   # Item::AGENT0, bboard.hpp:67
MAX_BOMBS = 20     # bboard.hpp:25-27 (4 agents x 5)


def observe(states: np.ndarray, per_agent: bool = False, dtype=np.uint8):
    """planes [n,16,11,11] (or [n,4,16,11,11]), agent_attrs [n,4,8] int32, board part of env_attrs: timeStep, aliveAgents"""
    n = len(states)
    board = states["board"].reshape(n, N, N).astype(np.int64)
    pl = np.zeros((n, PLANES, N, N), dtype=np.int64)
    pl[:, 0] = board == 0                        # Item::PASSAGE, bboard.hpp:56
    pl[:, 1] = board == 1                        # RIGID :57
    pl[:, 2] = (board >> 8) == 2                 # IS_WOOD :73-76
    pl[:, 3] = board == 3                        # BOMB :59
    pl[:, 4] = (board >> 16) == 4                # IS_FLAME :85-88
    pl[:, 5] = board == 6                        # EXTRABOMB :63
    pl[:, 6] = board == 7                        # INCRRANGE :64
    pl[:, 7] = board == 8                        # KICK :65
    for i in range(4):
        pl[:, 8 + i] = board == AGENT0 + i       # AGENT0.. :67-70
    bq = states["bombs_queue"].astype(np.int64)
    bidx = states["bombs_index"].astype(np.int64)
    bcnt = states["bombs_count"].astype(np.int64)
    fq = states["flames_queue"]
    fidx = states["flames_index"].astype(np.int64)
    fcnt = states["flames_count"].astype(np.int64)
    for e in range(n):
        seen = set()
        for k in range(int(bcnt[e])):                      # queue order, as State::GetBomb scans (bboard.cpp:277-287)
            b = int(bq[e, (bidx[e] + k) % MAX_BOMBS])
            x, y = b & 0xF, (b >> 4) & 0xF                 # BMB_POS_X / BMB_POS_Y, bboard.hpp:270-277
            if (x, y) in seen or x >= N or y >= N:
                continue
            seen.add((x, y))
            pl[e, 12, y, x] = (b >> 12) & 0xF              # BMB_STRENGTH :282-285
            pl[e, 13, y, x] = (b >> 16) & 0xF              # BMB_TIME :286-289
            pl[e, 14, y, x] = (b >> 20) & 0xF              # BMB_DIR :290-293
        ys, xs = np.nonzero(pl[e, 4])
        for y, x in zip(ys, xs):
            fid = (int(board[e, y, x]) & 0xFFFF) >> 3      # FLAME_ID :98-101
            for k in range(min(int(fcnt[e]), MAX_BOMBS)):
                f = fq[e, (fidx[e] + k) % MAX_BOMBS]
                if int(f["x"]) + N * int(f["y"]) == fid:   # the centre a flame was spawned at, bboard.cpp:198-205
                    pl[e, 15, y, x] = min(max(int(f["timeLeft"]), 0), 255)
                    break
    ag = states["agents"]
    attrs = np.zeros((n, 4, 8), dtype=np.int32)             # AgentInfo, bboard.hpp:228-245
    attrs[:, :, 0] = ag["x"]
    attrs[:, :, 1] = ag["y"]
    attrs[:, :, 2] = ag["dead"] == 0
    attrs[:, :, 3] = ag["maxBombCount"] - ag["bombCount"]   # what PlantBombModifiedLife checks, bboard.cpp:125-131
    attrs[:, :, 4] = ag["bombCount"]
    attrs[:, :, 5] = ag["maxBombCount"]
    attrs[:, :, 6] = ag["bombStrength"]
    attrs[:, :, 7] = ag["canKick"] != 0
    if per_agent:
        views = np.empty((n, 4, PLANES, N, N), dtype=np.int64)
        for a in range(4):
            order = list(range(8)) + [8 + ((a + j) & 3) for j in range(4)] + list(range(12, 16))
            views[:, a] = pl[:, order]
        pl = views
    return pl.astype(dtype), attrs, np.stack([states["timeStep"], states["aliveAgents"]], axis=1).astype(np.int32)


def observe_codes(states: np.ndarray) -> np.ndarray:
    """the compact form (POM_OBS_CODES), uint8 [n,5,11,11]: the board as the small numbers of the Item enum (bboard.hpp:54-71),
    then planes 12..15 of observe()"""
    n = len(states)
    board = states["board"].reshape(n, N, N).astype(np.int64)
    out = np.full((n, 5, N, N), 255, dtype=np.int64)
    for v in (0, 1, 3, 5, 6, 7, 8, 9):                    # PASSAGE RIGID BOMB FOG EXTRABOMB INCRRANGE KICK AGENTDUMMY
        out[:, 0][board == v] = v
    out[:, 0][(board >> 8) == 2] = 2                      # IS_WOOD :73-76
    out[:, 0][(board >> 16) == 4] = 4                     # IS_FLAME :85-88
    for i in range(4):
        out[:, 0][board == AGENT0 + i] = 10 + i           # AGENT0.. :67-70
    pl, _, _ = observe(states)
    out[:, 1:5] = pl[:, 12:16]
    return out.astype(np.uint8)
