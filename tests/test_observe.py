"""Observation export: pom_batch_observe against the numpy restatement oracle/pom_observe_oracle.py,
bit-exact (uint8 / half / float hold the same small integers), through the C-ABI via the ctypes wrapper."""
import importlib.util
import os

import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.state import Item

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle():
    spec = importlib.util.spec_from_file_location("pom_observe_oracle", os.path.join(ROOT, "oracle", "pom_observe_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_planes_on_a_hand_made_state():
    """the restatement itself, on a state whose planes can be read off by eye"""
    from pomcpp_amd.state import new_states
    ob = _oracle()
    s = new_states(1)
    b = s["board"][0]
    b[0, 0] = Item.AGENT0
    b[2, 3] = Item.RIGID
    b[4, 5] = Item.WOOD + 2
    b[6, 6] = Item.BOMB
    b[7, 1] = Item.KICK
    b[9, 9] = Item.FLAMES + ((9 + 11 * 9) << 3) + 1
    s["bombs_queue"][0, 0] = 6 + (6 << 4) + (1 << 8) + (3 << 12) + (7 << 16) + (4 << 20)
    s["bombs_queue"][0, 1] = 0 + (0 << 4) + (0 << 8) + (2 << 12) + (9 << 16)       # under agent 0
    s["bombs_queue"][0, 2] = 6 + (6 << 4) + (2 << 8) + (5 << 12) + (1 << 16)       # second bomb on (6,6): hidden
    s["bombs_count"][0] = 3
    s["flames_queue"][0, 0]["x"] = 9
    s["flames_queue"][0, 0]["y"] = 9
    s["flames_queue"][0, 0]["timeLeft"] = 3
    s["flames_count"][0] = 1
    s["agents"][0, 0]["bombCount"] = 1
    pl, attrs, _ = ob.observe(s)
    assert pl.shape == (1, 16, 11, 11) and pl.dtype == np.uint8
    assert pl[0, 8, 0, 0] == 1 and pl[0, 1, 2, 3] == 1 and pl[0, 2, 4, 5] == 1 and pl[0, 3, 6, 6] == 1
    assert pl[0, 7, 7, 1] == 1 and pl[0, 4, 9, 9] == 1 and pl[0, 15, 9, 9] == 3
    assert (pl[0, 12, 6, 6], pl[0, 13, 6, 6], pl[0, 14, 6, 6]) == (3, 7, 4)
    assert (pl[0, 12, 0, 0], pl[0, 13, 0, 0]) == (2, 9)
    assert pl[0, :12].sum(axis=0).max() == 1 and pl[0, 0].sum() == 121 - 6
    assert attrs[0, 0].tolist() == [0, 0, 1, 0, 1, 1, 1, 0]
    v, _, _ = ob.observe(s, per_agent=True)
    assert v.shape == (1, 4, 16, 11, 11) and v[0, 0, 8, 0, 0] == 1 and v[0, 1, 11, 0, 0] == 1 and v[0, 3, 9, 0, 0] == 1


def test_oracle_codes_on_a_hand_made_state():
    from pomcpp_amd.state import new_states
    ob = _oracle()
    s = new_states(1)
    b = s["board"][0]
    b[0, 0] = Item.AGENT0 + 3
    b[2, 3] = Item.RIGID
    b[4, 5] = Item.WOOD + 2
    b[6, 6] = Item.BOMB
    b[7, 1] = Item.KICK
    b[8, 8] = 5            # FOG
    b[8, 9] = 2            # not an Item the reference writes (WOOD is 2 << 8)
    b[9, 9] = Item.FLAMES + ((9 + 11 * 9) << 3) + 1
    s["bombs_queue"][0, 0] = 6 + (6 << 4) + (1 << 8) + (3 << 12) + (7 << 16) + (4 << 20)
    s["bombs_count"][0] = 1
    s["flames_queue"][0, 0]["x"] = 9
    s["flames_queue"][0, 0]["y"] = 9
    s["flames_queue"][0, 0]["timeLeft"] = 3
    s["flames_count"][0] = 1
    c = ob.observe_codes(s)
    assert c.shape == (1, 5, 11, 11) and c.dtype == np.uint8
    assert (c[0, 0, 0, 0], c[0, 0, 2, 3], c[0, 0, 4, 5], c[0, 0, 6, 6], c[0, 0, 7, 1], c[0, 0, 8, 8], c[0, 0, 8, 9], c[0, 0, 9, 9]) == (13, 1, 2, 3, 8, 5, 255, 4)
    assert (c[0, 1, 6, 6], c[0, 2, 6, 6], c[0, 3, 6, 6], c[0, 4, 9, 9]) == (3, 7, 4, 3)
    assert (c[0, 0] == 0).sum() == 121 - 8 and c[0, 1:].sum() == 3 + 7 + 4 + 3


def _played_states(n, ticks, kind="ffa", seed=3):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=300)
    env.make_game(pa.make_boards(n, seed=seed, kind=kind))
    env.step_random(seed, 2 if kind == "stress" else 1, ticks=ticks)
    return env


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,ticks", [("ffa", 200, 0), ("ffa", 1000, 57), ("stress", 777, 23), ("stress", 64, 5), ("ffa", 5, 120)])
def test_planes_match_the_oracle(hip_lib, kind, n, ticks):
    ob = _oracle()
    env = _played_states(n, ticks, kind)
    states = env.get_state()
    want, want_attrs, want_env = ob.observe(states)
    got, attrs, eattrs = env.observe()
    assert got.shape == (n, 16, 11, 11)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(attrs.cpu().numpy(), want_attrs)
    st = env.status()
    e = eattrs.cpu().numpy()
    assert np.array_equal(e[:, :2], want_env)
    assert np.array_equal(e[:, 2] & 1, st["done"]) and np.array_equal((e[:, 2] >> 1) & 1, st["draw"]) and np.array_equal(e[:, 3], st["winner"])
    # the stress boards must actually exercise the bomb / flame planes
    if kind == "stress" and ticks > 10:
        assert want[:, 12].any() and want[:, 14].any() and want[:, 15].any() and want[:, 4].any()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("per_agent", [False, True])
@pytest.mark.parametrize("dtype", ["uint8", "float16", "float32"])
def test_views_and_element_types(hip_lib, per_agent, dtype):
    ob = _oracle()
    env = _played_states(333, 40, "stress", seed=9)
    want, _, _ = ob.observe(env.get_state(), per_agent=per_agent, dtype=getattr(np, dtype))
    got, a, e = env.observe(per_agent=per_agent, dtype=dtype, attrs=False)
    assert a is None and e is None
    assert got.dtype.itemsize == np.dtype(dtype).itemsize
    assert np.array_equal(got.cpu().numpy(), want)
    again, _, _ = env.observe(per_agent=per_agent, dtype=dtype, attrs=False, out=got)
    assert again.data_ptr() == got.data_ptr() and np.array_equal(again.cpu().numpy(), want)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,ticks", [("ffa", 200, 0), ("ffa", 1001, 57), ("stress", 778, 23), ("stress", 67, 5), ("ffa", 5, 120), ("ffa", 16, 30)])
def test_codes_match_the_oracle(hip_lib, kind, n, ticks):
    """the compact layout (POM_OBS_CODES); batch sizes with every remainder mod 4 (a pass stages four envs and the batch's last
    bytes leave one by one), the bytes behind the last env untouched"""
    import torch
    ob = _oracle()
    env = _played_states(n, ticks, kind)
    want = ob.observe_codes(env.get_state())
    buf = torch.full((n * 605 + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    got, attrs, eattrs = env.observe(dtype="codes", out=buf[: n * 605].view(n, 5, 11, 11))
    assert np.array_equal(got.cpu().numpy(), want)
    assert (buf[n * 605:] == 0xAB).all()
    # an array that starts 4 / 8 bytes into a 16-byte line (a dword boundary is what the ABI asks for): the export's dword path —
    # the 16-byte stores need the lines
    for off in (4, 8):
        buf2 = torch.full((n * 605 + 80,), 0xCD, dtype=torch.uint8, device="cuda")
        got2, _, _ = env.observe(dtype="codes", attrs=False, out=buf2[off: off + n * 605].view(n, 5, 11, 11))
        assert np.array_equal(got2.cpu().numpy(), want), off
        assert (buf2[:off] == 0xCD).all() and (buf2[off + n * 605:] == 0xCD).all()
    _, want_attrs, want_env = ob.observe(env.get_state())
    assert np.array_equal(attrs.cpu().numpy(), want_attrs) and np.array_equal(eattrs.cpu().numpy()[:, :2], want_env)
    if kind == "stress" and ticks > 10:
        assert want[:, 1].any() and want[:, 3].any() and want[:, 4].any() and (want[:, 0] == 4).any()
    with pytest.raises(ValueError):
        env.observe(dtype="codes", per_agent=True)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["uint8", "codes"])
def test_first_in_queue_speaks_for_a_cell(hip_lib, dtype):
    """hand-made states crowded with duplicates — up to 20 live bombs on five cells, up to 20 live flames on five centres (and up to
    255 counted), queue heads anywhere in the ring, flame cells whose centre has no flame: the first
    live slot in queue order speaks (State::GetBomb's scan, bboard.cpp:277-287).  The export resolves "first" with key tables and a
    lowering loop; this is the test that makes that loop run."""
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    from pomcpp_amd.state import new_states
    ob = _oracle()
    rng = np.random.default_rng(12)
    n = 203
    s = new_states(n)
    for e in range(n):
        cells = rng.choice(121, size=5, replace=False)
        centres = rng.choice(121, size=5, replace=False)
        s["bombs_index"][e] = rng.integers(0, 20)
        s["bombs_count"][e] = rng.integers(0, 21)
        for k in range(20):
            c = int(cells[rng.integers(0, 5)])
            s["bombs_queue"][e, k] = (c % 11) | ((c // 11) << 4) | (int(rng.integers(0, 4)) << 8) | (int(rng.integers(1, 11)) << 12) \
                | (int(rng.integers(0, 10)) << 16) | (int(rng.integers(0, 5)) << 20)
        s["flames_index"][e] = rng.integers(0, 20)
        s["flames_count"][e] = rng.choice([0, 1, 5, 20, 21, 200, int(rng.integers(0, 21))])
        for k in range(20):
            c = int(centres[rng.integers(0, 5)])
            f = s["flames_queue"][e, k]
            f["x"], f["y"], f["timeLeft"], f["strength"] = c % 11, c // 11, rng.integers(-3, 12), rng.integers(0, 8)
        b = s["board"][e].reshape(-1)
        for c in cells:
            b[c] = Item.BOMB if rng.integers(0, 2) else Item.AGENT0 + int(rng.integers(0, 4))
        for c in rng.choice(121, size=30, replace=False):
            if c in cells:
                continue
            centre = int(centres[rng.integers(0, 5)]) if rng.integers(0, 4) else int(rng.integers(0, 121))  # some without a flame
            flag = int(rng.integers(0, 4))
            # a flame cell with a power-up under it is a burnt wood at the end of a ray: its centre lies on its row or column
            # (pom_packed.h; anything else no game reaches and the record does not hold)
            if flag and (centre == c or (centre % 11 != c % 11 and centre // 11 != c // 11)):
                others = [k for k in list(range(c % 11, 121, 11)) + list(range(c - c % 11, c - c % 11 + 11)) if k != c]
                centre = int(others[rng.integers(0, len(others))])
            b[c] = Item.FLAMES + (centre << 3) + flag
    with BatchEnvironment(n, mode=MODE_ENV) as env:
        env.make_game(s)
        states = env.get_state()
        assert states.tobytes() == s.tobytes()
        if dtype == "codes":
            want = ob.observe_codes(states)
        else:
            want, _, _ = ob.observe(states)
        got, _, _ = env.observe(dtype=dtype, attrs=False)
        assert np.array_equal(got.cpu().numpy(), want)
        assert want[:, -1].any() and (want[:, 1 if dtype == "codes" else 12] > 0).sum() > 2 * n


@pytest.mark.gpu
def test_observation_follows_the_game(hip_lib):
    """observe -> step -> observe on the stream the steps run on: planes always describe the state a download returns"""
    ob = _oracle()
    env = _played_states(4096 + 17, 0)
    for t in range(6):
        env.step_simple(11, 7)
        got, _, _ = env.observe()
        want, _, _ = ob.observe(env.get_state())
        assert np.array_equal(got.cpu().numpy(), want), t
    env.close()


@pytest.mark.gpu
def test_bad_arguments_are_refused(hip_lib):
    import torch
    from pomcpp_amd.batch import PomError
    env = _played_states(64, 0)
    buf = torch.empty(64 * 16 * 121 + 16, dtype=torch.uint8, device="cuda")
    from pomcpp_amd.batch import _check
    with pytest.raises(PomError):  # misaligned planes pointer
        _check(env._lib, env._lib.pom_batch_observe(env._h, buf.data_ptr() + 1, 0, 0, None, None))
    with pytest.raises(PomError):  # unknown element type
        _check(env._lib, env._lib.pom_batch_observe(env._h, buf.data_ptr(), 7, 0, None, None))
    with pytest.raises(ValueError):
        env.observe(dtype="int64")
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("at_end,fresh", [(True, False), (False, False), (True, True)])
@pytest.mark.parametrize("per_agent,dtype", [(False, "uint8"), (True, "uint8"), (False, "float32"), (True, "float16"), (False, "codes")])
def test_step_and_observation_in_one_launch(hip_lib, oracle, at_end, fresh, per_agent, dtype):
    """pom_batch_step_device_observe: the launch that plays the tick writes the observation of the state it leaves behind.  Every
    tick: planes and attributes against the numpy restatement applied to the downloaded state, and the state itself against a
    twin handle stepped by pom_batch_step_device (itself checked against the oracle every tick in tests/test_reset_modes.py);
    with the reset at the end of the tick finished envs show their next start state, marked restarted."""
    import torch
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END
    ob = _oracle()
    n, cap = 1000 + 13, 40
    kw = dict(mode=MODE_ENV, auto_reset=RESET_AT_END if at_end else True, max_steps=cap, fresh_boards=fresh, board_seed=5)
    rng = np.random.default_rng(4)
    with BatchEnvironment(n, **kw) as env, BatchEnvironment(n, **kw) as twin:
        for e in (env, twin):
            if fresh:
                e.generate(5)
            else:
                e.make_game(pa.make_boards(n, seed=8, kind="stress"))
        restarted_seen = 0
        for t in range(70):
            mv = torch.from_numpy(rng.integers(0, 6, size=(n, 4), dtype=np.int32)).to("cuda")
            got, attrs, eattrs = env.step_device_observe(mv, per_agent=per_agent, dtype=dtype)
            twin.step_device(mv)
            states = env.get_state()
            assert states.tobytes() == twin.get_state().tobytes(), t
            want, want_attrs, want_env = ob.observe(states, per_agent=per_agent, dtype=np.uint8 if dtype == "codes" else getattr(np, dtype))
            if dtype == "codes":
                want = ob.observe_codes(states)
            assert np.array_equal(got.cpu().numpy(), want), t
            assert np.array_equal(attrs.cpu().numpy(), want_attrs), t
            e = eattrs.cpu().numpy()
            st = env.status()
            assert np.array_equal(e[:, :2], want_env), t
            assert np.array_equal(e[:, 2] & 1, st["done"]) and np.array_equal(e[:, 3], st["winner"]), t
            if at_end:
                fin = env.last_results()["finished"]
                assert np.array_equal((e[:, 2] >> 3) & 1, fin), t
                restarted_seen += int(fin.sum())
        assert not at_end or restarted_seen > n // 2
        assert np.array_equal(env.counters(), twin.counters())
