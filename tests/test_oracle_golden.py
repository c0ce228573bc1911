"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/*.json)."""
import pytest

from oracle import Oracle
from golden_runner import all_cases, run_case

CASES = all_cases()


@pytest.mark.parametrize("fname,case", CASES, ids=[f"{f}:{c['name']}" for f, c in CASES])
def test_oracle_golden(fname, case):
    run_case(case, lambda ints, k, all_kmers: Oracle(ints, k, all_kmers))


def test_rng_lemire_equals_this_toolchains_libstdcxx():
    """RNG_LEMIRE restates libstdc++ >= 11's uniform_int_distribution; check it against the real one."""
    import numpy as np
    rng = np.random.default_rng(7)
    for n in [1, 2, 3, 7, 10, 100, 3000, 65536, 2**31, 2**32 - 2]:
        for seed in rng.integers(0, 2**32, size=50, dtype=np.uint64):
            a = Oracle.rng_generate(int(seed), 1, n, 3, 0)
            b = Oracle.rng_generate_std(int(seed), 1, n, 3)
            assert a.tolist() == b.tolist(), (seed, n)


def test_master_seed_stream_follows_5000_per_batch_rule():
    """quasimap.cpp:120-141: every batch of <= 5000 reads draws exactly 5000 seeds."""
    raw = Oracle.rng_raw(42, 15000)
    s = Oracle.master_seeds(42, [3, 5001])
    assert s[:3].tolist() == raw[:3].tolist()              # file 1: draws 0..2 (2997 discarded)
    assert s[3:5003].tolist() == raw[5000:10000].tolist()  # file 2, batch 1
    assert s[5003] == raw[10000]                           # file 2, batch 2
