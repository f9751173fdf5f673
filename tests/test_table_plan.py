"""kcfx_table_plan: the size policy of kcfx_db_open's cell tables, host arithmetic only (no GPU) -- the same two functions
(cells_lbits, cells_regions in kcfx_api.cpp) kcfx_db_open calls with the device's free memory."""
import pytest
from hypothesis import given, settings, strategies as st

from kcftools_amd import engine

GIB = 1 << 30
FREE = 260 * GIB                # an MI355X with nothing else resident


def per_cell(n, lines):
    return n / (16.0 * lines)


def test_the_sizes_the_profiles_quote():
    # the bench sample, a handle somebody keeps (expected_queries = 0): ~0.075 per cell in a power of two, one region
    assert engine.table_plan(150_079_223, 31, True, FREE) == (1 << 27, 1)
    # ... opened for ONE pass of a 30 Mb reference (the command-line driver, kcfx_screen_sharded): the first size at or under 0.3 per
    # cell -- 4.4 GB instead of 17 -- and so for a hundred passes; the sparser tables pay for their allocation past ~1e12 look-ups
    assert engine.table_plan(150_079_223, 31, True, FREE, expected_queries=30_010_326) == (1 << 25, 1)
    assert engine.table_plan(150_079_223, 31, True, FREE, expected_queries=100 * 30_010_326) == (1 << 25, 1)
    assert engine.table_plan(150_079_223, 31, True, FREE, expected_queries=10**12)[0] in (1 << 26, 1 << 27)
    assert engine.table_plan(150_079_223, 31, True, FREE, expected_queries=10**14) == (1 << 27, 1)
    # a one-pass table between the powers of two: 2^24 lines would be 0.45 per cell, the region steps stop at 1.5 x 2^24 (0.298)
    assert engine.table_plan(120_000_000, 31, True, FREE, expected_queries=1) == (3 << 23, 3)
    # the config-5 database: 2^28 lines would be 0.49 per cell; a table under 48 GiB grows up to them in quarters: 3 x 2^27
    assert engine.table_plan(2_110_000_000, 31, True, FREE) == (3 << 27, 3)
    # the wheat-scale database: 2^30 lines (the most a 32-bit mix of the minimizer addresses) = 0.47 per cell; 3 x 2^29 = 192 GiB
    # when 80 % of the free memory hold them, 5 x 2^28 = 160 GiB when not, 2^30 when that does not fit either
    assert engine.table_plan(8_080_000_000, 31, True, 245 * GIB) == (3 << 29, 3)
    assert engine.table_plan(8_080_000_000, 31, True, 210 * GIB) == (5 << 28, 5)
    assert engine.table_plan(8_080_000_000, 31, True, 180 * GIB)[1] == 1
    # a small device: the table shrinks towards one k-mer per cell, never below
    lines, _ = engine.table_plan(2_110_000_000, 31, True, 40 * GIB)
    assert 16 * lines >= 2_110_000_000 and lines * 128 <= 0.8 * 40 * GIB
    # k outside 19 .. 31, or more k-mers than any cell table has cells: a line table (lines = 0)
    assert engine.table_plan(1_000_000, 15, True, FREE) == (0, 1)
    assert engine.table_plan(1_000_000, 32, True, FREE) == (0, 1)
    assert engine.table_plan(40_000_000_000, 31, True, FREE) == (0, 1)
    with pytest.raises(engine.KcfxError):
        engine.table_plan(-1, 31, True, FREE)
    with pytest.raises(engine.KcfxError):
        engine.table_plan(1, 40, True, FREE)


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 16_000_000_000), st.sampled_from([19, 25, 31]), st.booleans(), st.integers(8, 280), st.integers(1, 10**15))
def test_expected_queries_only_ever_shrink_the_table(n, k, both, free_gib, q):
    """A stated number of look-ups never gives a larger table than "unknown", more look-ups never a smaller one, and the dense end
    stays at or under ~0.3 k-mers per cell wherever the unknown-queries policy would have been sparser than that."""
    free = free_gib * GIB
    unknown, _ = engine.table_plan(n, k, both, free)
    stated, regions = engine.table_plan(n, k, both, free, expected_queries=q)
    more, _ = engine.table_plan(n, k, both, free, expected_queries=min(10**15, 100 * q))
    if unknown == 0:
        assert stated == 0
        return
    assert 0 < stated <= unknown and 16 * stated >= n
    assert more >= stated
    assert regions in (1, 3, 5, 7)
    if per_cell(n, unknown) <= 0.30 and stated * 128 < 48 * GIB and stated >= (1 << 20):
        assert per_cell(n, stated) <= 0.30 + 1e-9 or stated == unknown


@settings(max_examples=300, deadline=None)
@given(st.integers(0, 16_000_000_000), st.sampled_from([19, 21, 25, 27, 31]), st.booleans(), st.integers(8, 280))
def test_policy_properties(n, k, both, free_gib):
    free = free_gib * GIB
    lines, regions = engine.table_plan(n, k, both, free)
    if lines == 0:
        return                                             # no cell table fits: line table
    assert regions in (1, 3, 5, 7)
    base = lines // regions
    assert base * regions == lines and base & (base - 1) == 0 and lines < 1 << 32
    assert base >= 1 << (17 if both else 18)               # r must keep its mark bits (cell_lbits_min)
    assert 16 * lines >= n                                 # a cell per k-mer at least
    d = per_cell(n, lines)
    bytes_ = lines * 128
    if regions > 1:
        # regions only where a power of two was left denser than 0.3 per cell, and never past 48 GiB / 80 % of the free memory
        pow2 = 1 << (lines.bit_length() - 1)
        assert per_cell(n, pow2) > 0.30
        assert bytes_ <= max(48 * GIB, 0.8 * free)
    if bytes_ > 48 * GIB:
        assert d > 0.25 * 0.30                             # nothing that large for a database that sparse
    if d > 0.5 and lines < 1 << 30:
        assert 2 * (1 << (lines.bit_length() - 1)) * 128 > 0.6 * free     # denser than 0.5 only because the device is small
    # more free memory never gives a smaller table
    more, _ = engine.table_plan(n, k, both, free + 64 * GIB)
    assert more >= lines or more == 0
