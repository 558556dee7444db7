"""The oracle reproduces the frozen fixtures in tests/golden bit for bit (CPU only)."""
import numpy as np
import pytest

import cases


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_oracle_matches_golden(oracle, name):
    got = cases.run_oracle(cases.CASES[name])
    want = cases.load_golden(name)
    assert cases.bits_equal(got["accum"], want["accum"])
    for key in ("rgba8", "prim_id", "chain_hash", "rng_state"):
        assert np.array_equal(got[key], want[key]), key
    assert got["counters"] == want["counters"]


def test_oracle_thread_count_does_not_change_results(oracle):
    case = cases.CASES["c3_mesh_bench_pose"]
    a = cases.run_oracle(case, n_threads=1)
    b = cases.run_oracle(case, n_threads=5)
    assert cases.bits_equal(a["accum"], b["accum"]) and a["counters"] == b["counters"]
