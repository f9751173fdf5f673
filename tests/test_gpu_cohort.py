"""Cohort tables (kcfx_cohort_open: S samples in ONE cell table, SURVEY.md 8 f-4 taken into the table; the reference does S
getVariations runs and a `cohort` merge, utils/run_kcftools.sh:124-160, Plugins/Cohort.java:71-119): every (window, sample)
record must equal what the sample's own database gives -- through the single-sample GPU path AND the oracle -- for k-mers
shared by all samples, by some, by one; counters of 1 and 2 bytes; -c thresholds; crowded tables whose stash carries real
weight; forward-only databases; and the inputs the entry point must refuse."""
import os

import numpy as np
import pytest

from kcftools_amd import engine, synth

pytestmark = pytest.mark.gpu
FIELDS = engine.INT_FIELDS
K = 31


def make_sample(tmp_path, name, ref, seed, cs=2, both=True, noise=3000, k=K, boost=False, **mut):
    sample = synth.mutate_sample(ref, seed, **mut)
    p = max(q for q in range(2, 8) if (k - q) % 4 == 0)
    prm = synth.KmcParams(k=k, p=p, counter_size=cs, sig_len=7, n_bins=8, both_strands=both, max_count=(1 << (8 * cs)) - 1)
    kmers, counts = synth.build_sample_db(sample, prm, seed + 1, noise_kmers=noise)
    if boost:                                   # counters around and beyond the 255 a cohort cell holds per sample (capped at the database's own)
        rng = np.random.default_rng(seed)
        sel = rng.random(len(counts)) < 0.15
        counts = counts.astype(np.int64)
        counts[sel] = np.minimum(counts[sel] * rng.integers(8, 60, int(sel.sum())), prm.max_count)
        counts = counts.astype(np.uint32)
    prefix = str(tmp_path / name)
    synth.write_kmc(prefix, kmers, counts, prm)
    return prefix


def check(cohort_rows, prefixes, ref_text, s, lens, min_count, oracle, what):
    for i, pfx in enumerate(prefixes):
        want = oracle.OracleDb(pfx).screen(ref_text, s, lens, min_count, threads=8)
        single = engine.KmcDb.open(pfx).screen(ref_text, s, lens, min_count)
        for f in FIELDS:
            assert np.array_equal(single[f], want[f]), (what, i, f, "single-sample path vs oracle")
            bad = np.nonzero(cohort_rows[i][f] != want[f])[0]
            assert len(bad) == 0, f"{what}: sample {i} field {f} differs at windows {bad[:6]}: {cohort_rows[i][f][bad[:6]]} vs {want[f][bad[:6]]}"


@pytest.mark.parametrize("n_samples", [1, 3, 8])
def test_cohort_rows_equal_the_single_sample_screens(oracle, tmp_path, monkeypatch, n_samples):
    ref = synth.random_reference(400_000, 31, n_run_every=40_000, max_n_run=300)
    prefixes = [make_sample(tmp_path, f"s{i}", ref, 100 + 7 * i, cs=1 + (i % 2), noise=2000 + 900 * i,
                            sub_rate=0.004 * (1 + i % 4)) for i in range(n_samples)]
    s, e = engine.windows(len(ref), 10_000, 0, K)
    lens = e - s
    co = engine.Cohort(prefixes)
    assert co.info.n_samples == n_samples and co.info.kmer_length == K and co.info.table_bytes > 0
    r = engine.Reference(ref, s, lens, K)
    for c in (1, 2, 5):
        got = co.screen(r, c)
        assert got.shape == (n_samples, len(s))
        check(got, prefixes, ref, s, lens, c, oracle, f"S={n_samples} -c {c}")
    if n_samples > 1:
        assert len({int(got[i]["obs"].sum()) for i in range(n_samples)}) > 1       # the samples do differ
    # the same from a reference without a plan (KCFX_REF_PLAN=0: the kernel derives minimizers from the text per launch)
    monkeypatch.setenv("KCFX_REF_PLAN", "0")
    r_text = engine.Reference(ref, s, lens, K)
    monkeypatch.delenv("KCFX_REF_PLAN")
    got_text = co.screen(r_text, 2)
    got_plan = co.screen(r, 2)
    for f in FIELDS:
        assert np.array_equal(got_text[f], got_plan[f]), ("planned vs text-only reference", f)
    # irregular windows (gene-model-like: 0 bp .. 40 kbp, overlapping, some shorter than k)
    rng = np.random.default_rng(5)
    wb = np.sort(rng.integers(0, len(ref) - 40_000, 300)).astype(np.int64)
    wl = rng.integers(0, 40_000, 300).astype(np.int32)
    wl[::17] = rng.integers(0, K, len(wl[::17]))
    r2 = engine.Reference(ref, wb, wl, K)
    check(co.screen(r2, 1), prefixes, ref, wb, wl, 1, oracle, "irregular windows")


@pytest.mark.parametrize("k,both", [(19, True), (22, True), (25, True), (25, False), (28, True), (30, True), (27, False)])
def test_cohort_tables_for_the_other_kmer_lengths(oracle, tmp_path, monkeypatch, k, both):
    """k = 19 .. 30 (KMC's default 25 among them): a k-mer spans W = k - 15 minimizer positions, second choices go to the cells
    nobody has as first choice (or by xor on small tables) -- the cohort kernel takes both schemes from the key table."""
    ref = synth.random_reference(250_000, 90 + k, n_run_every=50_000, max_n_run=200)
    prefixes = [make_sample(tmp_path, f"k{k}_{i}", ref, 500 + 11 * i + k, cs=1 + (i % 2), both=both, noise=4000, k=k) for i in range(3)]
    s, e = engine.windows(len(ref), 10_000, 0, k)
    for lbits in (None, "18"):                               # roomy (free-cell second choices) and crowded (few spare r bits: xor scheme)
        if lbits:
            monkeypatch.setenv("KCFX_TABLE_LBITS", lbits)
        co = engine.Cohort(prefixes)
        monkeypatch.delenv("KCFX_TABLE_LBITS", raising=False)
        assert co.info.kmer_length == k
        got = co.screen(engine.Reference(ref, s, e - s, k), 1)
        for i, pfx in enumerate(prefixes):
            want = oracle.OracleDb(pfx).screen(ref, s, e - s, 1, threads=8)
            for f in FIELDS:
                assert np.array_equal(got[i][f], want[f]), (k, both, lbits, i, f)


def test_cohort_counters_wider_than_a_cell_holds(oracle, tmp_path):
    """A cohort cell has 8 bits per sample (one more line request per minimizer run instead of 1.75 for u16). A k-mer that SOME
    sample counted more than 255 times lives in the stash with 16-bit counters for all samples -- also for the samples that hold it
    with a small counter or not at all. Counters of exactly 255 and 256, 65 535, whole runs of wide k-mers (a repeat: every k-mer of
    a window wide in one sample), -c thresholds above 255, KD sums: every (window, sample) against the oracle."""
    ref = synth.random_reference(300_000, 77, n_run_every=60_000, max_n_run=200)
    s, e = engine.windows(len(ref), 10_000, 0, K)
    lens = e - s
    prm = synth.KmcParams(k=K, p=3, counter_size=2, sig_len=7, n_bins=8, both_strands=True, max_count=65535)
    prefixes = []
    rng = np.random.default_rng(9)
    for i in range(4):
        kmers, counts = synth.build_sample_db(synth.mutate_sample(ref, 300 + i, sub_rate=0.004), prm, 400 + i, noise_kmers=3000)
        counts = counts.astype(np.int64)
        if i == 0:                                           # scattered wide counters, the boundary values among them
            idx = rng.choice(len(kmers), 4000, replace=False)
            counts[idx] = rng.choice([255, 256, 257, 300, 1000, 65535, 254], size=len(idx))
        elif i == 1:                                         # a repeat: every k-mer of the first 25 kb of the reference is wide here
            sel = np.isin(kmers, synth.kmers_of_text(ref[:25_000], K, True))
            counts[sel] = 500 + (np.arange(int(sel.sum())) % 7) * 100
        elif i == 2:
            counts[:] = np.minimum(counts * 60, 65535)       # a deep sample: many counters near and above 255
        prefix = str(tmp_path / f"w{i}")
        synth.write_kmc(prefix, kmers, counts.astype(np.uint32), prm)
        prefixes.append(prefix)
    co = engine.Cohort(prefixes)
    assert co.info.stash_records > 3000                      # the wide k-mers went there
    r = engine.Reference(ref, s, lens, K)
    for c in (1, 255, 256, 300, 1001):
        check(co.screen(r, c), prefixes, ref, s, lens, c, oracle, f"wide counters, -c {c}")
    # the same through a text-only reference and irregular windows
    r1 = engine.Reference(ref, s, lens, K, expected_screens=1)
    got_text, got_plan = co.screen(r1, 2), co.screen(r, 2)
    for f in FIELDS:
        assert np.array_equal(got_text[f], got_plan[f]), f


@pytest.mark.parametrize("lbits,ext", [(17, 0), (17, 1), (20, 1)])
def test_cohort_with_a_crowded_table_and_a_loaded_stash(oracle, tmp_path, monkeypatch, lbits, ext):
    """2^17 lines for ~2 M records: second choices and the stash (with its own counters) carry real weight; 2^20: third lines too
    (their cells carry counters like any other). ext = 1: an extended key table (what >= 0.5 G records get, cell_first_ext)."""
    ref = synth.random_reference(500_000, 41)
    prefixes = [make_sample(tmp_path, f"d{i}", ref, 200 + i, cs=2, noise=60_000) for i in range(4)]
    s, e = engine.windows(len(ref), 10_000, 0, K)
    monkeypatch.setenv("KCFX_TABLE_LBITS", str(lbits))
    monkeypatch.setenv("KCFX_TABLE_EXT", str(ext))
    co = engine.Cohort(prefixes)
    monkeypatch.delenv("KCFX_TABLE_LBITS")
    monkeypatch.delenv("KCFX_TABLE_EXT")
    assert co.info.table_lines == 1 << lbits and (lbits > 17 or co.info.stash_records > 0.05 * co.info.total_records)
    r = engine.Reference(ref, s, e - s, K)
    check(co.screen(r, 1), prefixes, ref, s, e - s, 1, oracle, "crowded")
    check(co.screen(r, 3), prefixes, ref, s, e - s, 3, oracle, "crowded -c 3")


def test_cohort_of_forward_only_databases(oracle, tmp_path):
    ref = synth.random_reference(200_000, 51)
    prefixes = [make_sample(tmp_path, f"f{i}", ref, 300 + i, both=False) for i in range(3)]
    s, e = engine.windows(len(ref), 10_000, 0, K)
    co = engine.Cohort(prefixes)
    assert co.info.both_strands == 0
    check(co.screen(engine.Reference(ref, s, e - s, K), 1), prefixes, ref, s, e - s, 1, oracle, "forward-only")


def test_cohort_async_on_a_torch_stream_and_what_it_refuses(oracle, tmp_path):
    import torch
    ref = synth.random_reference(150_000, 61)
    prefixes = [make_sample(tmp_path, f"a{i}", ref, 400 + i) for i in range(2)]
    s, e = engine.windows(len(ref), 10_000, 0, K)
    co = engine.Cohort(prefixes, expected_queries=int((e - s - K + 1).sum()))
    r = engine.Reference(ref, s, e - s, K)
    out = torch.zeros(2 * len(s) * 10, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            co.screen_async(r, out.data_ptr(), st.cuda_stream, 1)
    st.synchronize()
    got = out.cpu().numpy().view(engine.RESULT_DTYPE).reshape(2, len(s))
    check(got, prefixes, ref, s, e - s, 1, oracle, "async")
    # refused: more than 8 samples, k outside 19 .. 31, mixed k, 4-byte counters, mixed strandedness, -c 0
    with pytest.raises(engine.KcfxError) as err:
        engine.Cohort([prefixes[0]] * 9)
    assert err.value.code == 1
    prm = synth.KmcParams(k=15, p=3, counter_size=2, sig_len=7, n_bins=4)              # k outside 19 .. 31: no cell table
    km, ct = synth.build_sample_db(synth.mutate_sample(ref, 9), prm, 10, noise_kmers=100)
    synth.write_kmc(str(tmp_path / "k15"), km, ct, prm)
    k25 = make_sample(tmp_path, "k25", ref, 78, k=25)                                   # fine alone, not next to k = 31
    prm4 = synth.KmcParams(k=K, p=3, counter_size=4, sig_len=7, n_bins=4, max_count=2**31 - 1)
    km, ct = synth.build_sample_db(synth.mutate_sample(ref, 9), prm4, 10, noise_kmers=100)
    synth.write_kmc(str(tmp_path / "cs4"), km, ct, prm4)
    fwd = make_sample(tmp_path, "fwd", ref, 77, both=False)
    for bad in ([str(tmp_path / "k15")], [prefixes[0], k25], [prefixes[0], str(tmp_path / "cs4")], [prefixes[0], fwd]):
        with pytest.raises(engine.KcfxError) as err:
            engine.Cohort(bad)
        assert err.value.code == 1, bad
    with pytest.raises(engine.KcfxError):
        co.screen(r, 0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("KCFX_FUZZ_SEEDS", "6"))))       # soak: KCFX_FUZZ_SEEDS=200
def test_cohort_fuzz_against_the_single_sample_path(oracle, tmp_path, monkeypatch, seed):
    """Random cohorts: 1 .. 8 samples with their own mutation rates, noise and counter widths (one database may appear twice),
    table sizes from roomy to crowded (2^17 lines: second choices and the stash with its counters carry weight), -c 1 .. 4,
    windows on the kernel's internal borders, N runs. Every (window, sample) record against the sample's own table; one sample
    per case against the oracle."""
    rng = np.random.default_rng(7000 + seed)
    L = 150_000
    ref = bytearray(synth.random_reference(L, 800 + seed))
    for _ in range(12):
        pos = int(rng.integers(0, L - 3000))
        n = int(rng.choice([1, 2, 31, 64, 700]))
        ref[pos:pos + n] = b"N" * n
    ref = bytes(ref)
    S = int(rng.integers(1, 9))
    both = bool(seed % 4)
    k = (31, 25, 29, 31, 27, 23, 31, 30, 21, 19)[seed % 10]
    prefixes = [make_sample(tmp_path, f"z{i}", ref, 9000 + 31 * seed + i, cs=int(rng.integers(1, 3)), both=both, k=k,
                            noise=int(rng.choice([500, 20_000, 150_000])), sub_rate=float(rng.choice([0.001, 0.01, 0.03])),
                            boost=(seed + i) % 3 == 1)
                for i in range(S)]
    if S > 2 and seed % 3 == 0:
        prefixes[-1] = prefixes[0]                          # the same database twice: two identical columns
    if seed % 2:
        monkeypatch.setenv("KCFX_TABLE_LBITS", "17" if both else "18")
    co = engine.Cohort(prefixes)
    monkeypatch.delenv("KCFX_TABLE_LBITS", raising=False)
    # window lengths on the borders: k-mer counts 0 / 1 / 63 / 64 / 65, one tile +- 1, several tiles +- 1
    base = [0, k - 1, k, k + 62, k + 63, k + 64, 2047 + k - 1, 2048 + k - 1, 2049 + k - 1, 4096 + k - 1, 4097 + k - 1, 10_000, 33_333]
    wl = np.array([base[int(rng.integers(0, len(base)))] for _ in range(120)], dtype=np.int32)
    wb = np.array([int(rng.integers(0, L - int(x) + 1)) for x in wl], dtype=np.int64)
    r = engine.Reference(ref, wb, wl, k)
    c = int(rng.integers(1, 5))
    got = co.screen(r, c)
    for i, pfx in enumerate(prefixes):
        single = engine.KmcDb.open(pfx).screen(ref, wb, wl, c)
        for f in FIELDS:
            bad = np.nonzero(got[i][f] != single[f])[0]
            assert len(bad) == 0, (seed, i, f, bad[:5], got[i][f][bad[:5]], single[f][bad[:5]])
    j = int(rng.integers(0, S))
    want = oracle.OracleDb(prefixes[j]).screen(ref, wb, wl, c, threads=4)
    for f in FIELDS:
        assert np.array_equal(got[j][f], want[f]), (seed, j, f)


def test_cohort_of_empty_and_tiny_databases(oracle, tmp_path):
    """Databases without a record, with one record, and a normal one in the same cohort: the empty sample's column is all misses."""
    ref = synth.random_reference(60_000, 71)
    prm = synth.KmcParams(k=K, p=3, counter_size=2, sig_len=7, n_bins=4, max_count=65535)
    empty, one = str(tmp_path / "empty"), str(tmp_path / "one")
    synth.write_kmc(empty, np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32), prm)
    km, ct = synth.build_sample_db(synth.mutate_sample(ref, 72), prm, 73, noise_kmers=10)
    synth.write_kmc(one, km[:1], ct[:1], prm)
    normal = make_sample(tmp_path, "normal", ref, 74)
    s, e = engine.windows(len(ref), 10_000, 0, K)
    r = engine.Reference(ref, s, e - s, K)
    for prefixes in ([empty], [empty, empty], [empty, one, normal], [normal, empty]):
        got = engine.Cohort(prefixes).screen(r, 1)
        check(got, prefixes, ref, s, e - s, 1, oracle, str([os.path.basename(p) for p in prefixes]))
    assert int(engine.Cohort([empty, normal]).screen(r, 1)[0]["obs"].sum()) == 0
