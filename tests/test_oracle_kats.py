"""Pin the oracle against the hand-derived known-answer vectors of SURVEY.md section 8(c).

The reference has no tests or fixtures for this path (README.md:470) and cannot run here, so these
vectors -- each derived by tracing the cited Java lines -- are the only external anchor
("parity unpinned", see oracle/kcf_oracle.h).
"""
import numpy as np
import pytest

from oracle import pymodel

K = 31
H, M = 1, 0

# (hit/miss pattern over 100 valid k-mers) -> obs, var, inner, left, right, score   (total=100, eff=130)
KATS = [
    ([H] * 100, (100, 0, 0, 0, 0), 100.0),
    ([H] * 40 + [M] * 31 + [H] * 29, (69, 1, 1, 0, 0), 87.36923076923078),
    ([M] * 5 + [H] * 88 + [M] * 7, (88, 2, 0, 5, 7), 92.43076923076923),
    ([M] * 100, (0, 1, 0, 0, 100), 0.0),
    ([H] * 50 + [M] * 1 + [H] * 49, (99, 1, 28, 0, 0), 93.13846153846154),
    ([H] * 10 + [M] * 29 + [H] * 61, (71, 1, 0, 0, 0), 88.39999999999999),
    ([H] * 10 + [M] * 30 + [H] * 60, (70, 1, 1, 0, 0), 87.76923076923077),
    ([H] * 3 + [M] * 31 + [H] * 2 + [M] * 62 + [H] * 2, (7, 2, 33, 0, 0), 55.184615384615384),
    ([M] * 31 + [H] * 69, (69, 1, 0, 31, 0), 80.44615384615383),
    ([H, M] * 50, (50, 50, 1372, 0, 1), -236.84615384615378),
]


@pytest.mark.parametrize("pattern,ints,score", KATS)
def test_accumulator_kats(oracle, pattern, ints, score):
    assert len(pattern) == 100
    res = oracle.accumulate(pattern, K, 1)
    assert (res["obs"], res["var"], res["inner"], res["left"], res["right"]) == ints
    assert res["total_kmers"] == 100
    res["eff_length"] = 130
    s, mean, bad = oracle.score(res)
    assert not bad
    assert s == pytest.approx(score, abs=1e-12)
    # the independent model agrees on the same pattern
    pm = pymodel.reduce_counts(pattern, K, 1)
    pm["eff_length"] = 130
    for f in ("obs", "var", "inner", "left", "right", "total_kmers", "kmer_count_sum"):
        assert pm[f] == res[f]
    assert pymodel.score(pm)[0] == pytest.approx(score, abs=1e-12)


def test_distance_kats(oracle):
    # GetVariants.java:267-273 for k = 31
    want = {1: 28, 2: 27, 28: 1, 29: 0, 30: 1, 31: 1, 32: 2, 100: 70}
    for g, d in want.items():
        assert oracle.lib().kcfo_distance(K, g) == d
        assert pymodel.distance(K, g) == d


def test_empty_list_is_all_zero(oracle):
    res = oracle.accumulate([], K, 1)
    assert all(v == 0 for v in res.values())
    res["eff_length"] = 0
    assert oracle.score(res)[0] == 0.0


def test_weights_sum_check(oracle):
    assert 0.3 + 0.3 + 0.4 == 1.0           # the default weights pass Data.java:101
    res = oracle.accumulate([H] * 100, K, 1)
    res["eff_length"] = 130
    assert oracle.score(res, 0.3, 0.3, 0.4)[2] == 0
    assert oracle.score(res, 0.3, 0.3, 0.5)[2] == 1


def test_min_count_and_count_sum(oracle):
    counts = [5, 1, 0, 7, 2, 2, 9]
    r1 = oracle.accumulate(counts, K, 1)
    assert (r1["obs"], r1["kmer_count_sum"], r1["var"], r1["inner"]) == (6, 26, 1, 28)
    r3 = oracle.accumulate(counts, K, 3)      # counts below -c are misses and not summed (Appendix B.7)
    assert (r3["obs"], r3["kmer_count_sum"]) == (3, 21)
    assert (r3["var"], r3["inner"], r3["left"], r3["right"]) == (2, 27 + 27, 0, 0)
    assert pymodel.reduce_counts(counts, K, 3)["inner"] == 54


def test_signature_norm_table_kats(oracle):
    from kcftools_amd import synth
    norm = oracle.norm_table(9)
    special = 4 ** 9
    allowed = np.array([oracle.lib().kcfo_sig_is_allowed(i, 9) for i in range(special)])
    assert int(allowed.sum()) == 159912
    assert int((norm == special).sum()) == 53560
    assert len(np.unique(norm)) == 104293

    def mm(s):
        v = 0
        for ch in s:
            v = (v << 2) | "ACGT".index(ch)
        return v

    assert norm[mm("AAAAAAAAA")] == 262144
    assert norm[mm("TTTTTTTTT")] == 262144
    assert norm[mm("ACACACACA")] == 262144
    assert norm[mm("ACGTACGTA")] == 27756
    assert norm[mm("CCCCCCCCC")] == 87381
    assert norm[mm("GATTACAGA")] == 146504
    assert norm[mm("CGCGCGCGC")] == 104857
    # the vectorised twin used by the synthetic KMC writer is identical
    assert np.array_equal(norm, synth.norm_table(9))
    for m in (3, 5, 7):
        assert np.array_equal(oracle.norm_table(m), synth.norm_table(m))


def test_window_tiling_counts(oracle):
    # SURVEY 8 a-1
    for L, w, n, last in [(1_000_000, 10_000, 101, (997000, 1000000)), (30_427_671, 10_000, 3052, None),
                          (119_667_750, 10_000, 12003, None)]:
        s, e = oracle.windows(L, w, 0, K)
        assert len(s) == n
        if last:
            assert (s[-1], e[-1]) == last
        # every k-mer of the contig belongs to exactly one window
        assert int(((e - s) - K + 1).sum()) == L - K + 1
        assert np.all(s[1:] == e[:-1] - K + 1)
    with pytest.raises(ValueError):
        oracle.windows(1000, 30, 0, K)        # w <= k-1 never advances in the reference
    s, e = oracle.windows(100_000, 10_000, 2_500, K)
    assert np.all(s == np.arange(0, 100_000, 2_500)[: len(s)])
    assert np.all(e - s >= K)


def test_kmer_encode_and_revcomp(oracle):
    import ctypes as C
    from kcftools_amd import synth
    L = oracle.lib()
    s = "ACGTTGCAAGGCTTAACCGGTTAACGATCGA"
    assert len(s) == 31
    w = C.c_uint64()
    assert L.kcfo_kmer_encode(s.encode(), 31, C.byref(w)) == 0
    assert w.value == synth.str_to_kmer(s)
    assert w.value & 3 == 0                                   # k = 31: low 2 bits empty (Kmer.java:243)
    rc = L.kcfo_kmer_revcomp(w.value, 31)
    assert synth.kmer_to_str(rc, 31) == pymodel.revcomp_str(s)
    can = L.kcfo_kmer_canonical(w.value, 31)
    assert synth.kmer_to_str(can, 31) == min(s, pymodel.revcomp_str(s))
    assert L.kcfo_kmer_prefix(w.value, 7) == int("".join(str("ACGT".index(c)) for c in s[:7]), 4)
    buf = (C.c_uint8 * 8)()
    L.kcfo_kmer_suffix(w.value, 31, 7, buf)
    suf = s[7:]
    want = [sum("ACGT".index(suf[4 * b + j]) << (6 - 2 * j) for j in range(4)) for b in range(6)]
    assert list(buf)[:6] == want
    assert L.kcfo_kmer_encode(b"ACGTNACGTACGTACGTACGTACGTACGTACG", 31, C.byref(w)) == -1
