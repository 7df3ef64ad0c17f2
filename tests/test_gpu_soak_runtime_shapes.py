"""A short soak of the run-time-shape instances (tests/soak_runtime_shapes.py: random LV chains of width <= 16, random exposure chains
3-H1-H2-1, random Fisher-KPP reaction chains on larger grids) against the oracle, every gradient entry bit for bit; SOAK_ROUNDS=n for a longer one."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_runtime_shapes_match_the_oracle_bit_for_bit():
    import soak_runtime_shapes as S
    assert S.soak(int(os.environ.get("SOAK_ROUNDS", "25")), 77) == 0
