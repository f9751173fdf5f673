"""The two independent restatements (C, line-by-line; Python, set-membership) must agree.

The C oracle probes the on-disk KMC layout exactly like KMC.java:292-326 (signature -> bin -> LUT ->
bisection); the Python model parses the same files into a dict and works on strings.
"""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from kcftools_amd import synth
from oracle import pymodel

INT_FIELDS = ("total_kmers", "eff_length", "obs", "var", "inner", "left", "right", "kmer_count_sum")

CONFIGS = [
    # k,  p, cs, m, bins, both
    (31, 3, 2, 9, 16, True),
    (31, 7, 1, 9, 8, True),
    (31, 3, 4, 7, 5, False),
    (23, 3, 2, 5, 7, True),
    (15, 3, 1, 5, 3, False),
    (32, 4, 2, 7, 4, True),
    (31, 3, 3, 9, 2, True),
]


def make_case(tmp_path, seed, k, p, cs, m, bins, both, ref_len=6000, noise=300):
    ref = synth.random_reference(ref_len, seed, n_run_every=1500, max_n_run=90, lower_frac=0.2, iupac_rate=2e-3)
    sample = synth.mutate_sample(ref, seed + 1, sub_rate=1.5e-2, indel_rate=2e-3, big_del_per_mb=400, big_del_len=300)
    prm = synth.KmcParams(k=k, p=p, counter_size=cs, sig_len=m, n_bins=bins, both_strands=both,
                          max_count=(1 << (8 * cs)) - 1)
    kmers, counts = synth.build_sample_db(sample, prm, seed + 2, noise_kmers=noise)
    prefix = str(tmp_path / f"db_{seed}_{k}_{p}_{cs}_{m}_{int(both)}")
    synth.write_kmc(prefix, kmers, counts, prm)
    return ref, prefix, kmers, counts, prm


@pytest.mark.parametrize("k,p,cs,m,bins,both", CONFIGS)
def test_db_roundtrip_and_lookup(oracle, tmp_path, k, p, cs, m, bins, both):
    ref, prefix, kmers, counts, prm = make_case(tmp_path, 11, k, p, cs, m, bins, both)
    db = oracle.OracleDb(prefix)
    assert (db.info.kmer_length, db.info.lut_prefix_length, db.info.counter_size) == (k, p, cs)
    assert (db.info.signature_length, db.info.n_bins, bool(db.info.both_strands)) == (m, bins, both)
    assert db.info.total_kmers == len(kmers)
    # dumpKmerTable enumeration == what was written
    dk, dc = db.dump()
    order = np.argsort(dk)
    o2 = np.argsort(kmers)
    assert np.array_equal(dk[order], kmers[o2])
    assert np.array_equal(dc[order].astype(np.int64), counts[o2].astype(np.int64).astype(np.int32).astype(np.int64))
    # every stored k-mer is found with its count, by the reference's own search (getCount)
    for i in range(0, len(kmers), max(1, len(kmers) // 400)):
        assert db.get_count(int(kmers[i])) == int(np.int32(np.uint32(counts[i])))
    # the Python parser sees the same table
    table, hdr = pymodel.read_kmc(prefix)
    assert hdr["k"] == k and hdr["both_strands"] == both and len(table) == len(kmers)
    # absent k-mers give 0
    rng = np.random.default_rng(5)
    present = set(int(x) for x in kmers)
    mask = (0xFFFFFFFFFFFFFFFF << (64 - 2 * k)) & 0xFFFFFFFFFFFFFFFF
    for _ in range(300):
        w = int(rng.integers(0, 2 ** 63)) * 2 & mask
        if both:
            w = oracle.lib().kcfo_kmer_canonical(w, k)
        if w not in present:
            assert db.get_count(w) == 0


@pytest.mark.parametrize("k,p,cs,m,bins,both", CONFIGS)
def test_process_window_agrees(oracle, tmp_path, k, p, cs, m, bins, both):
    ref, prefix, *_ = make_case(tmp_path, 23, k, p, cs, m, bins, both)
    db = oracle.OracleDb(prefix)
    table, hdr = pymodel.read_kmc(prefix)
    starts, ends = synth.tiling_windows(len(ref), 700, k)
    assert len(starts) > 5
    for s, e in zip(starts, ends):
        text = ref[s:e]
        for mc in (1, 3):
            a = db.process_window(text, mc)
            b = pymodel.process_window(text.decode(), table, k, both, mc)
            for f in INT_FIELDS:
                assert a[f] == b[f], (f, s, e, mc)
            sa = oracle.score(a)[:2]
            sb = pymodel.score(b)
            assert sa[0] == pytest.approx(sb[0], abs=1e-9) and sa[1] == pytest.approx(sb[1], abs=1e-9)
    # batch driver (thread pool) == per-window calls
    out = db.screen(ref, starts, ends - starts, 1, threads=3)
    for i, (s, e) in enumerate(zip(starts, ends)):
        a = db.process_window(ref[s:e], 1)
        for f in INT_FIELDS:
            assert int(out[f][i]) == a[f]


def test_edge_windows(oracle, tmp_path):
    ref, prefix, kmers, counts, prm = make_case(tmp_path, 31, 31, 3, 2, 9, 4, True, ref_len=3000, noise=50)
    db = oracle.OracleDb(prefix)
    table, _ = pymodel.read_kmc(prefix)
    cases = [b"", b"ACGT", b"N" * 100, b"A" * 30, b"acgt" * 10, b"ACGTN" * 40,
             b"ACGTACGTACGTACGTACGTACGTACGTACG",             # exactly k
             b"ACGTACGTACGTACGTACGTACGTACGTACGN" + b"T" * 31,
             b"RYKM" + b"ACGTTGCA" * 8 + b"n" + b"ggcc" * 12]
    for text in cases:
        a = db.process_window(text, 1)
        b = pymodel.process_window(text.decode(), table, 31, True, 1)
        for f in INT_FIELDS:
            assert a[f] == b[f], (text, f)
    # identity total = sum(runLen - k + 1) over ACGT runs >= k (SURVEY a-8)
    a = db.process_window(b"ACGTACGTACGTACGTACGTACGTACGTACGN" + b"T" * 35, 1)
    assert a["total_kmers"] == 1 + 5 and a["eff_length"] == 31 + 35


@settings(max_examples=150, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=4), min_size=0, max_size=300), st.integers(1, 3), st.sampled_from([5, 15, 31]))
def test_accumulator_property(counts, min_count, k):
    from oracle import binding
    a = binding.accumulate(counts, k, min_count)
    b = pymodel.reduce_counts(counts, k, min_count)
    for f in ("total_kmers", "obs", "var", "inner", "left", "right", "kmer_count_sum"):
        assert a[f] == b[f]


@settings(max_examples=100, deadline=None)
@given(st.text(alphabet="ACGTacgtNnRY", min_size=0, max_size=200), st.sampled_from([3, 7, 15, 31]))
def test_kmer_list_property(text, k):
    from oracle import binding
    raw = text.encode()
    n = binding.lib().kcfo_kmers_list(raw, len(raw), k, None)
    want = [text[i:i + k].upper() for i in range(len(text) - k + 1) if not text[i:i + k].upper().strip("ACGT")]
    assert n == len(want)
    out = np.zeros(max(n, 1), dtype=np.uint64)
    binding.lib().kcfo_kmers_list(raw, len(raw), k, out.ctypes.data)
    assert [synth.kmer_to_str(int(w), k) for w in out[:n]] == want
    assert binding.lib().kcfo_eff_length(raw, len(raw), k) == pymodel.eff_length(text, k)
