"""The reference's `[step function]` suite (unit_test/bboard/board_logic.cpp:55-634), restated in
tests/step_cases.py, run against the oracle on CPU and against the HIP path through the C-ABI.
Beyond the reference's own assertions, every Step must reproduce the state the compiled reference
produced (tests/golden/step_cases.npz) bit for bit, and the setup helpers must build the very
states the reference's State methods build."""
import os

import numpy as np
import pytest

from tests.case_api import HostAPI
from tests.step_cases import ALL_CASES, CASES

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "step_cases.npz"))


def _run(name, stepper, oracle):
    api = HostAPI(stepper, oracle)
    ALL_CASES[name](api)
    before, moves, after = GOLDEN[f"{name}__before"], GOLDEN[f"{name}__moves"], GOLDEN[f"{name}__after"]
    assert len(api.trace) == len(before)
    for k, (b, m, a) in enumerate(api.trace):
        assert list(m) == moves[k].tolist()
        assert b == before[k].tobytes(), f"{name}: input state of step {k} differs from the reference's"
        assert a == after[k].tobytes(), f"{name}: state after step {k} differs from the reference's"


def test_suite_is_complete():
    assert len(CASES) == 32  # 10 TEST_CASEs, 32 leaf runs
    # one directed vector per quirk of SURVEY.md §9 (Q1 .. Q12), recorded from the compiled reference like the suite itself
    quirks = {name.split("_")[0] for name in ALL_CASES if name.startswith("q")}
    assert {f"q{k}" for k in range(1, 13)} <= quirks, sorted(quirks)
    assert all(f"{name}__after" in GOLDEN.files for name in ALL_CASES)


@pytest.mark.parametrize("name", list(ALL_CASES))
def test_case_oracle(name, oracle):
    _run(name, lambda s, m: oracle.step(s, m), oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ALL_CASES))
def test_case_gpu(name, oracle, hip_lib):
    from pomcpp_amd.batch import step_one
    _run(name, step_one, oracle)
