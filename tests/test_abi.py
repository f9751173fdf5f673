"""The C-ABI library loads and exports every symbol include/kcfx.h declares; host-only entry points
(kcfx_score, kcfx_windows) agree with the oracle.  No GPU compute here."""
import os
import re

import numpy as np
import pytest

from kcftools_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "kcfx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kcfx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = engine.lib()
    syms = header_symbols()
    assert len(syms) >= 15
    assert set(syms) == set(engine.ABI_SYMBOLS)
    for s in syms:
        assert hasattr(L, s), s
    assert b"gfx950" in L.kcfx_version()


def test_no_gpu_means_loud_failure_not_fallback():
    if engine.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(engine.KcfxError) as ei:
        engine.KmcDb.build(np.array([0], dtype=np.uint64), np.array([1], dtype=np.uint32), 31)
    assert ei.value.code == 4 and "no CPU fallback" in str(ei.value)
    with pytest.raises(engine.KcfxError):
        engine.Reference(b"ACGT" * 20, [0], [80], 31)
    with pytest.raises(engine.KcfxError) as ei:
        engine.screen_sharded(["nowhere"], 31, b"ACGT" * 20, [0], [80], devices=(0,))
    assert ei.value.code == 4 and "no CPU fallback" in str(ei.value)
    # the entry points of round 3: a golden database parses on the host, then the device is missing -- never a CPU path
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "db_canon_k31_p3_cs2_m9")
    for make in (lambda: engine.KmcDb.open(golden, expected_queries=1000), lambda: engine.Cohort([golden, golden])):
        with pytest.raises(engine.KcfxError) as ei:
            make()
        assert ei.value.code == 4 and "no CPU fallback" in str(ei.value)
    # ... while the sharding plan is host arithmetic and needs no device
    plan, used = engine.shard_plan([10_000] * 10, 31, 2, 4)
    assert used == engine.SHARD_GRID and len(plan) >= 4


def test_cache_fraction_and_table_plan_are_host_only():
    """kcfx_set_cache_fraction (the bound on the device memory the library keeps of what it has released: half of the device unless
    the host says otherwise) and kcfx_table_plan_for (the size policy as arithmetic) touch no GPU."""
    prev = engine.set_cache_fraction(0.7)
    assert prev == pytest.approx(float(os.environ.get("KCFX_POOL_FRACTION", 0.5)))
    assert engine.set_cache_fraction(2.0) == pytest.approx(0.7)          # out of range: unchanged
    assert engine.set_cache_fraction(prev) == pytest.approx(0.7)
    lines, regions = engine.table_plan(150_000_000, 31, True, 200 << 30, expected_queries=30_000_000)
    assert (lines, regions) == (1 << 25, 1)
    assert engine.shard_slice_io() == (0, 0, 0)


def test_db_header_is_host_only(tmp_path):
    """kcfx_db_header reads the KMC header fields (KMC.java:124-139) without a GPU."""
    from kcftools_amd import synth
    prm = synth.KmcParams(k=31, p=3, counter_size=2, sig_len=7, n_bins=5, both_strands=False, min_count=2, max_count=1234)
    prefix = str(tmp_path / "db")
    kmers = np.array([4, 8, 1 << 40], dtype=np.uint64)
    synth.write_kmc(prefix, kmers, np.array([1, 2, 3]), prm)
    i = engine.db_header(prefix)
    assert (i.kmer_length, i.lut_prefix_length, i.counter_size, i.signature_length, i.n_bins) == (31, 3, 2, 7, 5)
    assert (i.both_strands, i.min_count, i.max_count, i.total_kmers, i.version) == (0, 2, 1234, 3, 0x200)
    assert i.table_bytes == 0 and i.device == -1
    with pytest.raises(engine.KcfxError) as ei:
        engine.db_header(str(tmp_path / "missing"))
    assert ei.value.code == 2


def test_error_codes_without_gpu(tmp_path):
    with pytest.raises(engine.KcfxError) as ei:
        engine.KmcDb.open(str(tmp_path / "missing"))
    assert ei.value.code == 2
    # a KMC 1.x style database (version 0) is rejected like KMC.java:139-142
    from kcftools_amd import synth
    prm = synth.KmcParams(k=31, p=3, counter_size=1, sig_len=5, n_bins=2)
    prefix = str(tmp_path / "db")
    synth.write_kmc(prefix, np.array([4, 8], dtype=np.uint64), np.array([1, 2]), prm)
    raw = bytearray(open(prefix + ".kmc_pre", "rb").read())
    raw[-12:-8] = (0).to_bytes(4, "little")
    open(prefix + ".kmc_pre", "wb").write(raw)
    with pytest.raises(engine.KcfxError) as ei:
        engine.KmcDb.open(prefix)
    assert ei.value.code == 3 and "0x200" in str(ei.value)
    # (k - p) % 4 != 0 cannot be searched by the reference (HelperFunctions.java:232-243)
    with pytest.raises(engine.KcfxError):
        engine.KmcDb.build(np.zeros(1, dtype=np.uint64), np.ones(1, dtype=np.uint32), 33)
    # argument errors of the round-4 entry points are found before any device is asked for
    with pytest.raises(engine.KcfxError) as ei:
        engine.Reference(b"ACGT" * 20, [0], [80], 31, expected_screens=-1)
    assert ei.value.code == 1 and "expected_screens" in str(ei.value)            # KCFX_ERR_ARG
    with pytest.raises(engine.KcfxError) as ei:
        engine.Reference(b"ACGT" * 20, [70], [80], 31, expected_screens=1)            # a window past the text (FastaIndex.java:131-134)
    assert "outside the text" in str(ei.value)
    with pytest.raises(engine.KcfxError):
        engine.table_plan(10, 0, True, 1 << 30)


def test_score_matches_oracle(oracle):
    rng = np.random.default_rng(3)
    n = 500
    res = np.zeros(n, dtype=engine.RESULT_DTYPE)
    res["total_kmers"] = rng.integers(0, 10000, n)
    res["eff_length"] = res["total_kmers"] + 30 * rng.integers(0, 3, n)
    res["obs"] = (res["total_kmers"] * rng.random(n)).astype(np.int32)
    res["inner"] = rng.integers(0, 3000, n)
    res["left"] = rng.integers(0, 200, n)
    res["right"] = rng.integers(0, 200, n)
    res["kmer_count_sum"] = res["obs"].astype(np.int64) * rng.integers(1, 40, n)
    res[:5] = 0
    sc, md = engine.score(res)
    osc, omd = oracle.score_array(res)
    assert np.array_equal(sc, osc)          # same IEEE operations in the same order: bit-identical
    assert np.array_equal(md, omd)
    with pytest.raises(engine.KcfxError) as ei:
        engine.score(res, 0.3, 0.3, 0.5)
    assert ei.value.code == 5


def test_score_kats():
    # total = 100, eff = 130
    rows = [(100, 0, 0, 0, 100.0), (69, 1, 0, 0, 87.36923076923078), (88, 0, 5, 7, 92.43076923076923),
            (99, 28, 0, 0, 93.13846153846154), (71, 0, 0, 0, 88.39999999999999), (7, 33, 0, 0, 55.184615384615384),
            (69, 0, 31, 0, 80.44615384615383), (50, 1372, 0, 1, -236.84615384615378), (0, 0, 0, 100, 0.0)]
    res = np.zeros(len(rows), dtype=engine.RESULT_DTYPE)
    for i, (obs, inner, left, right, _) in enumerate(rows):
        res[i] = (100, 130, obs, 0, inner, left, right, 0, obs * 3)
    sc, md = engine.score(res)
    for i, row in enumerate(rows):
        assert sc[i] == pytest.approx(row[4], abs=1e-12)
        assert md[i] == (3.0 if row[0] else 0.0)


def test_windows_match_oracle(oracle):
    for L, w, step in [(1_000_000, 10_000, 0), (30_427_671, 10_000, 0), (12_345, 700, 0), (100_000, 10_000, 2_500),
                       (31, 50, 0), (30, 50, 0), (5000, 31, 0), (999, 100, 33)]:
        s, e = engine.windows(L, w, step, 31)
        os_, oe = oracle.windows(L, w, step, 31)
        assert np.array_equal(s, os_) and np.array_equal(e, oe)
    with pytest.raises(engine.KcfxError):
        engine.windows(1000, 30, 0, 31)
    with pytest.raises(engine.KcfxError):
        engine.windows(1000, 0, 0, 31)
