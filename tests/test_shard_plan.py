"""kcfx_shard_plan: the partitioning of kcfx_screen_sharded, host arithmetic only (no GPU). Every (sample, window) pair lies in
exactly one segment whatever the mode, the devices' k-mer loads are even where the mode promises it, and the samples x windows
grid (1 < S < G) makes a device open at most two databases where the window-sharded mode opens all S on every device."""
import numpy as np
from hypothesis import given, settings, strategies as st

from kcftools_amd import engine

K = 31


def coverage(plan, S, W):
    seen = np.zeros((S, W), dtype=np.int32)
    for dev, s, lo, hi in plan:
        assert 0 <= s < S and 0 <= lo < hi <= W, (dev, s, lo, hi)
        seen[s, lo:hi] += 1
    return seen


@settings(max_examples=150, deadline=None)
@given(st.integers(1, 12), st.integers(1, 9), st.lists(st.integers(0, 400), min_size=1, max_size=60),
       st.sampled_from([engine.SHARD_AUTO, engine.SHARD_SAMPLES, engine.SHARD_WINDOWS, engine.SHARD_GRID]))
def test_every_window_sample_pair_is_screened_exactly_once(G, S, lens, mode):
    wl = np.array(lens, dtype=np.int32)
    plan, used = engine.shard_plan(wl, K, S, G, mode)
    assert np.all(coverage(plan, S, len(wl)) == 1)
    assert all(0 <= dev < G for dev, *_ in plan)
    if mode == engine.SHARD_AUTO:
        assert used == (engine.SHARD_SAMPLES if S >= G else engine.SHARD_WINDOWS if S == 1 else engine.SHARD_GRID)
    else:
        assert used == mode


def test_grid_opens_two_databases_per_device_and_balances_the_kmers():
    rng = np.random.default_rng(3)
    wl = rng.integers(31, 60_000, 12_000).astype(np.int32)            # gene-model-like window sizes
    kmers = np.maximum(0, wl.astype(np.int64) - K + 1)
    for S, G in ((2, 8), (3, 8), (5, 8), (7, 8), (2, 3)):
        plan, used = engine.shard_plan(wl, K, S, G)
        assert used == engine.SHARD_GRID
        assert np.all(coverage(plan, S, len(wl)) == 1)
        per_dev_samples = [len({s for d, s, *_ in plan if d == g}) for g in range(G)]
        assert max(per_dev_samples) <= 2, per_dev_samples               # ... where KCFX_SHARD_WINDOWS opens all S on every device
        load = np.zeros(G)
        for d, s, lo, hi in plan:
            load[d] += kmers[lo:hi].sum()
        assert load.max() <= 1.02 * load.mean(), load
        wplan, _ = engine.shard_plan(wl, K, S, G, engine.SHARD_WINDOWS)
        assert min(len({s for d, s, *_ in wplan if d == g}) for g in range(G)) == S


def test_sample_and_window_modes_keep_their_shapes():
    wl = np.full(1000, 10_000, dtype=np.int32)
    plan, used = engine.shard_plan(wl, K, 8, 8)
    assert used == engine.SHARD_SAMPLES and sorted(plan) == [(g, g, 0, 1000) for g in range(8)]
    plan, used = engine.shard_plan(wl, K, 11, 4)
    assert used == engine.SHARD_SAMPLES and [s for d, s, *_ in plan if d == 1] == [1, 5, 9]
    plan, used = engine.shard_plan(wl, K, 1, 4)
    assert used == engine.SHARD_WINDOWS and [(lo, hi) for _, _, lo, hi in plan] == [(0, 250), (250, 500), (500, 750), (750, 1000)]
    # nothing to screen: no segments
    assert engine.shard_plan(wl, K, 0, 4)[0] == [] and engine.shard_plan(np.zeros(0, np.int32), K, 3, 4)[0] == []


def cohort_coverage(plan, S, W):
    """a cohort segment covers the group of up to 8 consecutive samples that starts at seg_sample"""
    seen = np.zeros((S, W), dtype=np.int32)
    for dev, s, lo, hi in plan:
        assert s % engine.COHORT_MAX == 0 and 0 <= s < S and 0 <= lo < hi <= W, (dev, s, lo, hi)
        seen[s:min(S, s + engine.COHORT_MAX), lo:hi] += 1
    return seen


@settings(max_examples=120, deadline=None)
@given(st.integers(1, 12), st.integers(1, 40), st.lists(st.integers(0, 400), min_size=1, max_size=60))
def test_cohort_mode_covers_every_pair_once(G, S, lens):
    wl = np.array(lens, dtype=np.int32)
    plan, used = engine.shard_plan(wl, K, S, G, engine.SHARD_COHORT)
    assert used == engine.SHARD_COHORT                          # (never chosen by AUTO: the other test pins that)
    assert np.all(cohort_coverage(plan, S, len(wl)) == 1)
    assert all(0 <= dev < G for dev, *_ in plan)


def test_cohort_mode_shapes_of_the_baseline_configs():
    """BASELINE configs[3]: 8 samples on 8 devices = ONE table, every device an eighth of the windows; configs[4]: 32 samples =
    four tables, two devices each. Loads within 2 %, a device builds at most two tables (one when G is a multiple of the groups)."""
    rng = np.random.default_rng(9)
    wl = rng.integers(31, 60_000, 12_003).astype(np.int32)
    kmers = np.maximum(0, wl.astype(np.int64) - K + 1)
    for S, G, tables_per_dev in ((8, 8, 1), (32, 8, 1), (16, 8, 1), (12, 8, 2), (8, 3, 1), (9, 2, 2), (40, 8, 2)):
        plan, used = engine.shard_plan(wl, K, S, G, engine.SHARD_COHORT)
        assert used == engine.SHARD_COHORT and np.all(cohort_coverage(plan, S, len(wl)) == 1)
        groups = -(-S // engine.COHORT_MAX)
        load = np.zeros(G)
        for d, s, lo, hi in plan:
            load[d] += kmers[lo:hi].sum()
        assert load.max() <= 1.02 * load.mean(), (S, G, load)
        assert abs(load.sum() - groups * kmers.sum()) < 1           # a group's look-ups are one pass whatever its size
        assert max(len({s for d, s, *_ in plan if d == g}) for g in range(G)) <= tables_per_dev, (S, G)
    plan, _ = engine.shard_plan(np.full(1000, 10_000, dtype=np.int32), K, 8, 8, engine.SHARD_COHORT)
    assert sorted(plan) == [(g, 0, 125 * g, 125 * (g + 1)) for g in range(8)]


def test_table_mode_gives_every_device_every_sample_and_the_slices_partition_the_minimizers():
    """KCFX_SHARD_TABLE: the TABLE is cut, not the work list -- every device screens every (sample, window) pair against its slice
    of that sample's table, and every minimizer (the 32 bits that name a first-choice line and r) belongs to exactly one slice."""
    wl = np.array([10_000, 50, 31, 0, 7_000] * 40, dtype=np.int32)
    for S, G in ((1, 4), (3, 2), (2, 8)):
        plan, used = engine.shard_plan(wl, K, S, G, engine.SHARD_TABLE)
        assert used == engine.SHARD_TABLE and len(plan) == S * G
        assert np.all(coverage(plan, S, len(wl)) == G)                      # once per device
        assert sorted(plan) == sorted((g, s, 0, len(wl)) for g in range(G) for s in range(S))
    assert engine.shard_plan(wl, K, 5, 5)[1] != engine.SHARD_TABLE         # never chosen by AUTO
    rng = np.random.default_rng(11)
    g32 = rng.integers(0, 1 << 32, 200_000, dtype=np.uint64)
    cls = rng.integers(0, 4, 200_000)
    for n in (1, 2, 3, 4, 7, 8, 64):
        sl = np.array([engine.table_slice_of(int(g), 27, n, int(c)) for g, c in zip(g32[:20_000], cls)])
        assert sl.min() >= 0 and sl.max() < n and len(np.unique(sl)) == n
        cnt = np.bincount(sl, minlength=n)
        assert cnt.max() <= 1.25 * cnt.mean() + 40, cnt                     # even within sampling noise
        # the slice class (two bits of the minimizer's order key, which the anchor bytes of a planned reference carry) leads: a class
        # belongs to the slices of one contiguous range -- one slice when n is 2 or 4, two of eight -- so a pass against a slice
        # knows from the anchors alone which positions are not its own
        for c in range(4):
            mine = np.unique(sl[cls[:20_000] == c])
            assert len(mine) == mine.max() - mine.min() + 1 and len(mine) <= (n + 3) // 4 + 1
            assert mine.min() == (c * n) // 4
        if n in (2, 4):
            assert np.array_equal(sl, cls[:20_000] * n // 4)
    # a function of the class and the 32 bits alone, whatever the table's size says about which of them are the line
    assert all(engine.table_slice_of(int(g), 24, 8, 2) == engine.table_slice_of(int(g), 29, 8, 2) for g in g32[:2000])
    # within a class the hash of (line : r) decides, not the line number alone
    line = 0x5A5A5 << 12
    assert len({engine.table_slice_of(line | r, 20, 16, 1) for r in range(4096)}) == 4
    assert engine.table_slice_of(1, 27, 0) == -1 and engine.table_slice_of(1, 27, 65) == -1 and engine.table_slice_of(1, 5, 4) == -1
    assert engine.table_slice_of(1, 27, 4, 4) == -1 and engine.table_slice_of(1, 27, 4, -1) == -1
