"""BatchedMT19937: the initial-condition draws of many envs as array operations, bit-identical to one
``np.random.RandomState`` per env (what the reference's ``reset`` does: pdegym/kuramoto/kuramoto.py:101,106)."""
import time

import numpy as np
import pytest

from pdegym.kuramoto.mt_batch import BatchedMT19937


def test_bit_identical_to_randomstate_across_twists_and_mixed_positions():
    seeds = [0, 1, 7, 100, 2 ** 32 - 1, 123456789, 42, 2 ** 31]
    E = len(seeds)
    ref = [np.random.RandomState(s) for s in seeds]
    mt = BatchedMT19937(E)
    mt.seed_rows(np.arange(E), seeds)
    # equal positions: several draws across the 624-word twist boundary (64 doubles = 128 words per draw)
    for _ in range(7):
        got = mt.uniform_rows(np.arange(E), -0.4, 0.4, 64)
        for i in range(E):
            np.testing.assert_array_equal(got[i], ref[i].uniform(-0.4, 0.4, size=64))
    # mixed positions: only some envs draw (an autoreset), different lengths later
    ids = np.array([5, 0, 3])
    got = mt.uniform_rows(ids, -0.4, 0.4, 256)
    for k, i in enumerate(ids):
        np.testing.assert_array_equal(got[k], ref[i].uniform(-0.4, 0.4, size=256))
    got = mt.uniform_rows(np.arange(E), -1.0, 2.5, 400)      # 800 words: more than one twist per call
    for i in range(E):
        np.testing.assert_array_equal(got[i], ref[i].uniform(-1.0, 2.5, size=400))
    # re-seeding a subset
    mt.seed_rows([2, 6], [99, 31337])
    ref[2], ref[6] = np.random.RandomState(99), np.random.RandomState(31337)
    got = mt.uniform_rows(np.arange(E), -0.4, 0.4, 48)
    for i in range(E):
        np.testing.assert_array_equal(got[i], ref[i].uniform(-0.4, 0.4, size=48))


def test_unseeded_streams_differ_and_large_batches_are_cheap():
    with pytest.raises(ValueError):
        BatchedMT19937(2).seed_rows([0], [2 ** 32])          # as RandomState: seeds live in [0, 2**32)
    mt = BatchedMT19937(4)
    mt.seed_rows(np.arange(4), [None] * 4)
    u = mt.uniform_rows(np.arange(4), -0.4, 0.4, 64)
    assert len({tuple(r) for r in u.tolist()}) == 4 and (np.abs(u) <= 0.4).all()
    E, N = 4096, 256
    mt = BatchedMT19937(E)
    t0 = time.perf_counter()
    mt.seed_rows(np.arange(E), list(range(1000, 1000 + E)))
    u = mt.uniform_rows(np.arange(E), -0.4, 0.4, N)
    dt = time.perf_counter() - t0
    np.testing.assert_array_equal(u[17], np.random.RandomState(1017).uniform(-0.4, 0.4, N))
    np.testing.assert_array_equal(u[E - 1], np.random.RandomState(1000 + E - 1).uniform(-0.4, 0.4, N))
    print(f"seed + draw of {E} x {N}: {dt * 1e3:.1f} ms")
    assert dt < 2.0
