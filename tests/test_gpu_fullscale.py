"""BASELINE.json full sizes on the GPU (configs[2]-[4] shapes): A. thaliana-sized five-contig genome (119.7 Mb), one
150 M-k-mer sample database, gene-model windows from a 27 000-gene GTF (irregular sizes), 10 kb and 50 kb tilings.
Checked through size-independent properties AND against the oracle on every window of the 10 kb / 50 kb tilings and on
every (window, sample) pair of the 8-sample cohort (the multi-threaded C oracle needs ~5 s per 119 M k-mers on the GPU box's
host cores); the gene-model and 3 Gb cases compare a few thousand sampled windows. Inputs come from the native generator."""
import json
import os
import subprocess
import time

import numpy as np
import pytest

from kcftools_amd import engine, synth
from oracle import pyhost

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYNTH = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
CLI = os.path.join(ROOT, "kcftools_amd", "bin", "kcftools-amd")
LENS = [30_427_671, 19_698_289, 23_459_830, 18_585_056, 26_975_502]      # TAIR10 chromosome lengths
FIELDS = engine.INT_FIELDS
K = 31
HOST_THREADS = os.cpu_count() or 8


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    d = tmp_path_factory.mktemp("full")
    prefix = str(d / "ath")
    out = subprocess.run([SYNTH, "--out", prefix, "--contig-lens", ",".join(map(str, LENS)), "--seed", "3", "--sample-seed", "4",
                          "--noise", "44000000", "--fasta"], check=True, capture_output=True, text=True).stdout
    meta = json.loads(out)
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    assert len(text) == sum(LENS) == 119_146_348
    db = engine.KmcDb.open(prefix)
    assert db.info.total_kmers == meta["db_kmers"] > 140_000_000
    return dict(prefix=prefix, text=text, db=db, meta=meta, dir=str(d))


@pytest.fixture(scope="module")
def cohort(genome):
    """Eight samples of the genome: the fixture's own plus seven more seeds (about 100 M k-mers each)."""
    prefixes = [genome["prefix"]]
    for seed in range(11, 18):
        pfx = os.path.join(genome["dir"], f"s{seed}")
        subprocess.run([SYNTH, "--out", pfx, "--contig-lens", ",".join(map(str, LENS)), "--seed", "3", "--sample-seed", str(seed),
                        "--noise", "1000000"], check=True, capture_output=True, text=True)
        prefixes.append(pfx)
    return prefixes


def tiling(window):
    wb, wl = [], []
    off = 0
    for L in LENS:
        s, e = engine.windows(L, window, 0, K)
        wb.append(s.astype(np.int64) + off)
        wl.append((e - s).astype(np.int32))
        off += L
    return np.concatenate(wb), np.concatenate(wl)


def test_tilings_partition_the_genome(oracle, genome):
    db, text = genome["db"], genome["text"]
    wb10, wl10 = tiling(10_000)
    wb50, wl50 = tiling(50_000)
    assert len(wb10) == sum(len(engine.windows(L, 10_000, 0, K)[0]) for L in LENS) and len(wb10) > 11_900
    ref10 = engine.Reference(text, wb10, wl10, K)
    t0 = time.time()
    r10 = ref10.screen(db, 1)
    t10 = time.time() - t0
    r50 = engine.Reference(text, wb50, wl50, K).screen(db, 1)
    whole_b = np.cumsum([0] + LENS[:-1])
    rw = db.screen(text, whole_b, LENS, 1)
    # every k-mer of a contig lies in exactly one tiling window, whatever the window size (SURVEY a-1)
    for f in ("total_kmers", "obs", "kmer_count_sum"):
        assert int(r10[f].sum()) == int(r50[f].sum()) == int(rw[f].sum()), f
    assert 0.5 < r10["obs"].sum() / r10["total_kmers"].sum() < 0.8
    # EFFLEN / TOTAL identity: eff - total = (k-1) x (number of valid runs) >= 0, and total <= len - k + 1
    assert np.all(r10["eff_length"] >= r10["total_kmers"]) and np.all((r10["eff_length"] - r10["total_kmers"]) % (K - 1) == 0)
    assert np.all(r10["total_kmers"] <= wl10 - K + 1)
    # accounting identities of the gap state machine: obs + misses = total; every miss-run is counted once
    misses = r10["total_kmers"] - r10["obs"]
    assert np.all((r10["var"] == 0) == (misses == 0))
    assert np.all(r10["left"] + r10["right"] <= misses)
    # idempotent, and -c is monotone
    assert all(np.array_equal(ref10.screen(db, 1)[f], r10[f]) for f in FIELDS)
    r3 = ref10.screen(db, 3)
    assert np.all(r3["obs"] <= r10["obs"]) and np.array_equal(r3["total_kmers"], r10["total_kmers"])
    # EVERY window of both tilings against the oracle (2 x 119 M k-mers on all host threads: seconds)
    odb = oracle.OracleDb(genome["prefix"])
    t0 = time.time()
    for wb, wl, res in ((wb10, wl10, r10), (wb50, wl50, r50)):
        want = odb.screen(text, wb, wl, 1, threads=HOST_THREADS)
        for f in FIELDS:
            assert np.array_equal(res[f], want[f]), f
    print(f"oracle on all {len(wb10)} + {len(wb50)} windows: {time.time() - t0:.1f} s on {HOST_THREADS} threads")
    print(f"12k windows x 119 M k-mers screened in {t10 * 1e3:.1f} ms (host call, results copied back)")


def test_gene_model_windows_end_to_end(oracle, genome):
    """configs[2]: GTF gene windows through the command-line driver; EVERY gene's integers against the oracle, the full row text
    (Python writer) for 2 000 of them."""
    contigs = [(f"chr{i + 1}", L) for i, L in enumerate(LENS)]
    gtf = os.path.join(genome["dir"], "genes.gtf")
    synth.random_gtf(gtf, contigs, 11, genes_per_mb=227.0)
    out = os.path.join(genome["dir"], "genes.kcf")
    t0 = time.time()
    p = subprocess.run([CLI, "getVariations", "-r", genome["prefix"] + ".fa", "-k", genome["prefix"], "-o", out, "-s", "S1", "-f", "gene",
                        "-g", gtf], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
    wall = time.time() - t0
    rows = [l.split("\t") for l in open(out) if not l.startswith("#")]
    assert len(rows) > 25_000
    model = pyhost.Gtf(gtf)
    fa = open(genome["prefix"] + ".fa", "rb").read()
    entries = {e[0]: e for e in pyhost.generate_faidx(genome["prefix"] + ".fa")}
    odb = oracle.OracleDb(genome["prefix"])
    by_id = {r[3]: r for r in rows}
    rng = np.random.default_rng(1)
    genes = [g for c, _ in contigs for g in model.kids(c)]
    assert len(genes) == len(rows)
    # EVERY gene against the oracle: the spliced window texts (merged, start-sorted exon loci, no separator: GTF.java:223-293)
    # concatenated into one buffer and screened in one multi-threaded call; every integer of every row must agree
    pieces, wb, wl, off = [], [], [], 0
    for g in genes:
        loci = model.merged_exon_loci(g, True)
        text = b"".join(pyhost.get_sequence(fa, entries[l[0]], l[1] - 1, l[2] - l[1] + 1) for l in loci)
        pieces.append(text)
        wb.append(off)
        wl.append(len(text))
        off += len(text)
    t1 = time.time()
    want = odb.screen(b"".join(pieces), np.array(wb, dtype=np.int64), np.array(wl, dtype=np.int32), 1, threads=HOST_THREADS)
    t_oracle = time.time() - t1
    for i, g in enumerate(genes):
        r = by_id[g]
        info = dict(kv.split("=", 1) for kv in r[5].split(";") if "=" in kv)
        _, va, ob, inner, ld, rd, _, _ = r[7].rstrip("\n").split(":")
        got = (int(r[4]), int(info["EFFLEN"]), int(ob), int(va), int(inner), int(ld), int(rd))
        exp = tuple(int(want[f][i]) for f in ("total_kmers", "eff_length", "obs", "var", "inner", "left", "right"))
        assert got == exp, (g, got, exp)
    # ... and the whole text of the row (INFO statistics, KD / SC formatting) for a sample of them
    for i in rng.choice(len(genes), 2000, replace=False):
        g = genes[i]
        c, s, e, _ = model.loci(g)
        res = {f: int(want[f][i]) for f in FIELDS}
        sc, md, _ = oracle.score(res)
        assert "\t".join(by_id[g]).rstrip("\n") == pyhost.kcf_row(c, s, e, g, res, sc, md)
    print(f"oracle on all {len(genes)} gene windows ({off / 1e6:.0f} Mb of spliced text): {t_oracle:.1f} s on {HOST_THREADS} threads")
    # rows are grouped by contig in .faidx order and sorted by start inside a contig
    order = [r[0] for r in rows]
    assert [c for i, c in enumerate(order) if i == 0 or order[i - 1] != c] == [c for c, _ in contigs]
    for c, _ in contigs:
        st = [int(r[1]) for r in rows if r[0] == c]
        assert st == sorted(st)
    print(f"getVariations -f gene on 119 Mb / {len(rows)} genes: {wall:.1f} s wall (FASTA + GTF parse, DB load, screen, KCF write)")


def test_eight_samples_in_one_pass(oracle, genome, cohort):
    """configs[3] shape: the 119 Mb genome tiled at 10 kb against 8 resident sample databases, one pass over the reference
    per step (kcfx_screen_ref_multi) -- 12 k windows x 8 samples. Every slice must equal the single-sample pass; two
    samples are also sampled against the oracle; then the same through the command-line driver for three of them."""
    prefixes = cohort
    dbs = [genome["db"]] + [engine.KmcDb.open(p) for p in prefixes[1:]]
    assert all(d.info.total_kmers > 90_000_000 for d in dbs)
    text = genome["text"]
    wb, wl = tiling(10_000)
    ref = engine.Reference(text, wb, wl, K)
    t0 = time.time()
    got = ref.screen_multi(dbs, 1)
    wall = time.time() - t0
    assert got.shape == (8, len(wb))
    for s, db in enumerate(dbs):
        single = ref.screen(db, 1)
        for f in FIELDS:
            assert np.array_equal(got[s][f], single[f]), (s, f)
    assert len({int(got[s]["obs"].sum()) for s in range(8)}) == 8          # eight different samples
    # every (window, sample) pair against the oracle: 8 x 119 M k-mers on all host threads
    t0 = time.time()
    for s in range(8):
        want = oracle.OracleDb(prefixes[s]).screen(text, wb, wl, 1, threads=HOST_THREADS)
        for f in FIELDS:
            assert np.array_equal(got[s][f], want[f]), (s, f)
    print(f"oracle on all 8 x {len(wb)} window-samples: {time.time() - t0:.1f} s on {HOST_THREADS} threads")
    n_kmers = int(got["total_kmers"].sum())
    print(f"8 samples x {len(wb)} windows = {n_kmers / 1e9:.2f} G k-mer probes in {wall * 1e3:.1f} ms (host call, results copied back)")
    # the same eight samples in ONE cohort table (BASELINE configs[3] is this: whole genome, 8 samples): every (window, sample)
    # record identical to the per-sample tables' -- which the oracle has just confirmed
    del dbs
    engine.release_cached()
    t0 = time.time()
    co = engine.Cohort(prefixes)
    t_open = time.time() - t0
    t0 = time.time()
    cgot = co.screen(ref, 1)
    t_co = time.time() - t0
    for s in range(8):
        for f in FIELDS:
            assert np.array_equal(cgot[s][f], got[s][f]), ("cohort", s, f)
    ci = co.info
    print(f"cohort table of 8 samples: {ci.total_records / 1e6:.0f} M records, 2^{int(np.log2(ci.table_lines))} lines, "
          f"{ci.table_bytes / 2**30:.1f} GiB, {100.0 * ci.stash_records / ci.total_records:.1f} % of the records in the stash, opened in "
          f"{t_open:.1f} s; 8 x {len(wb)} window-samples in {t_co * 1e3:.1f} ms (host call) against {wall * 1e3:.1f} ms for 8 tables")
    del co
    # the command-line driver on three of them: one multi-sample KCF, rows = windows, 3 sample columns
    out = os.path.join(genome["dir"], "cohort.kcf")
    p = subprocess.run([CLI, "getVariations", "-r", genome["prefix"] + ".fa", "-k", ",".join(prefixes[:3]), "-o", out,
                        "-s", "A,B,C", "-f", "window", "-w", "10000"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
    rows = [l.rstrip("\n").split("\t") for l in open(out) if not l.startswith("#")]
    assert len(rows) == len(wb) and all(len(r) == 7 + 3 for r in rows)
    for s in range(3):
        obs_col = np.array([int(r[7 + s].split(":")[2]) for r in rows])
        assert np.array_equal(np.sort(obs_col), np.sort(got[s]["obs"]))


def test_table_slices_at_genome_scale(genome, cohort):
    """KCFX_SHARD_TABLE at the size of BASELINE configs[3]: two samples of the 119 Mb genome (150 M and 100 M k-mers), each one's
    TABLE cut into three slices by minimizer (three workers on the one GPU), every slice screened against all 11 953 windows, the hit
    bitmaps merged: every record equal to the unsharded call's; and the gene-like short windows of the sharding fixture's kind."""
    wb, wl = tiling(10_000)
    prefixes = [genome["prefix"], cohort[3]]
    t0 = time.time()
    got, st = engine.screen_sharded(prefixes, K, genome["text"], wb, wl, 1, devices=(0, 0, 0), mode=engine.SHARD_TABLE)
    t_tab = time.time() - t0
    assert st.shard_mode_used == engine.SHARD_TABLE and got.shape == (2, len(wb))
    want0 = genome["db"].screen(genome["text"], wb, wl, 1)
    want1 = engine.KmcDb.open(cohort[3]).screen(genome["text"], wb, wl, 1)
    for f in FIELDS:
        assert np.array_equal(got[0][f], want0[f]) and np.array_equal(got[1][f], want1[f]), f
    assert int(got[0]["total_kmers"].sum()) > 0.97 * len(genome["text"]) and 0.5 < got[0]["obs"].sum() / got[0]["total_kmers"].sum() < 0.9
    print(f"table slices: 2 samples x 3 slices x {len(wb)} windows in {t_tab:.2f} s (loads {st.seconds_db_open_sum:.2f} s summed, "
          f"screening + merge {st.seconds_screen_sum:.3f} s)")


def test_repeat_structured_genome_at_chr1_scale(oracle, tmp_path_factory):
    """Inputs with genome structure (kcfx_synth --repeats 0.45: TE families at 80 - 95 % identity, tandem satellite arrays,
    segmental duplications, microsatellites -- in the reference AND the sample genome, which is derived from it) at the size of
    BASELINE configs[1]: repeats crowd minimizers (one 16-mer is the minimizer of thousands of database k-mers), fill lines of
    the cell table and load the stash -- none of which uniform text exercises (Fasta.java:90-127, Kmer.java:105-118 see such
    data in every real genome). Every window against the oracle, for the sparse and a dense table; kernel time and table
    statistics are logged for profiles/."""
    import torch
    d = tmp_path_factory.mktemp("repeats")
    prefix = str(d / "rep")
    meta = json.loads(subprocess.run([SYNTH, "--out", prefix, "--ref-len", "30427671", "--seed", "5", "--sample-seed", "6",
                                      "--repeats", "0.45", "--noise", "20000000"], check=True, capture_output=True, text=True).stdout)
    assert meta["repeat_bases"]["te"] > 5_000_000 and meta["db_kmers"] - meta["noise_kmers"] < 0.95 * meta["sample_kmers"]
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    s, e = engine.windows(len(text), 10_000, 0, K)
    wb, wl = s.astype(np.int64), (e - s).astype(np.int32)
    want = oracle.OracleDb(prefix).screen(text, wb, wl, 1, threads=HOST_THREADS)
    ref = engine.Reference(text, wb, wl, K)
    out = torch.zeros(len(wb) * 10, dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream()
    lines = []
    for name, kw in (("sparse", {}), ("dense", {"expected_queries": int((wl - K + 1).clip(0).sum())})):
        db = engine.KmcDb.open(prefix, **kw)
        got = ref.screen(db, 1)
        for f in FIELDS:
            assert np.array_equal(got[f], want[f]), (name, f)
        for _ in range(5):
            ref.screen_async(db, out.data_ptr(), stream.cuda_stream, 1)
        torch.cuda.synchronize()
        ref.prof_enable(True)
        ref.prof_read(reset=True)
        for _ in range(50):
            ref.screen_async(db, out.data_ptr(), stream.cuda_stream, 1)
        torch.cuda.synchronize()
        ms, cnt = ref.prof_read(reset=True)
        ref.prof_enable(False)
        i = db.info
        nk = int(got["total_kmers"].sum())
        lines.append(f"repeat-structured genome ({name} table: 2^{int(np.log2(i.table_lines))} lines, {i.total_kmers / (16.0 * i.table_lines):.3f} per cell, "
                     f"{100.0 * i.stash_kmers / i.total_kmers:.2f} % of {i.total_kmers / 1e6:.1f} M k-mers in the stash, "
                     f"{100.0 * i.second_choice_kmers / i.total_kmers:.2f} % second-choice, {i.multi_cell_kmers} in several cells): "
                     f"kernel {ms / cnt:.4f} ms per pass of {nk / 1e6:.1f} M k-mers = {nk / (ms / cnt * 1e-3) / 1e9:.0f} G k-mers/s, "
                     f"hit rate {got['obs'].sum() / nk:.3f}; all {len(wb)} windows bit-exact")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "repeats_genome.log"), "a") as fh:
        for ln in lines:
            print(ln)
            fh.write(ln + "\n")


def _mem_available_bytes():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return 0


_SIZE_DECISIONS = {}


def _size_case(name, size, decide):
    """The two size-adaptive tests run as `[full]` and `[fallback]`: exactly one of the two runs, the other SKIPS with the reason,
    so that `N passed, M skipped` (and -rs) says which size a box ran. `decide()` -> (bases, why) is evaluated once per test."""
    if name not in _SIZE_DECISIONS:
        _SIZE_DECISIONS[name] = decide()
    L, why = _SIZE_DECISIONS[name]
    ran = "fallback" if why.startswith("FALLBACK") else "full"
    if size != ran:
        pytest.skip(f"the [{ran}] case runs on this box: {L / 1e9:.1f} Gb ({why})")
    return L, why


@pytest.mark.parametrize("size", ["full", "fallback"])
def test_config5_shape_wheat_scale(oracle, genome, cohort, monkeypatch, size):
    """BASELINE.json configs[4] as it is stated: a wheat-scale 14.5 Gb reference, 50 kb tiling windows (290 k of them), THIRTY-TWO
    samples, k = 31 -- through kcfx_screen_sharded (what the command-line driver runs; two workers on the one GPU, the next
    database loaded while the current one is screened). The reference is ONE 14.5 Gb genome (kcfx_synth --ref-only: the fixture's
    119 Mb genome is its first 119 Mb); the samples are THIRTY-ONE distinct samples of 119 Mb stretches of it (eight of the first
    stretch -- the cohort fixture -- and 23 of stretches spread over the rest, --ref-offset; ~100 M k-mers each; the cohort's eight in rotation only where the
    scratch space does not take 23 more -- the log line says which) plus ONE LARGE database of 2.1 G k-mers (17 GB .kmc_suf; a cell
    table of 3 x 2^27 lines at 0.33 k-mers per cell). Falls back to a 3.2 Gb reference only when the box lacks
    the host memory or scratch disk (KCFX_HUGE=<bases> overrides) and says which size ran. Also: the large database
    * as resident handles through kcfx_screen_ref_multi (ten > 8 forces a second batch),
    * rebuilt by re-streaming its file, and as a 12-slot line table (>= 2^29 lines, which the multi-sample kernel refuses),
    * DENSE: a 2^27-line cell table = 0.98 k-mers per cell (KCFX_TABLE_LBITS), the regime a wheat-size sample forces -- same
      results on every window, kernel time logged beside the 0.49-per-cell table and a sparse 150 M-k-mer sample.
    10 000 sampled windows (0.5 G k-mers) of four samples, the large (and dense) one among them, against the oracle."""
    import shutil
    big_kmers = int(float(os.environ.get("KCFX_BIG_DB", "2.0e9")))
    def decide():
        L = int(float(os.environ.get("KCFX_HUGE", "14.5e9")))
        why = "KCFX_HUGE" if os.environ.get("KCFX_HUGE") else "BASELINE configs[4] size"
        if not os.environ.get("KCFX_HUGE"):
            need_ram, have_ram = 6 * L, _mem_available_bytes()
            if have_ram and have_ram < need_ram:
                L, why = int(3.2e9), f"FALLBACK: {have_ram / 1e9:.0f} GB of host memory available, 14.5 Gb needs {need_ram / 1e9:.0f} GB"
        return L, why
    L, why = _size_case("config5", size, decide)
    need = big_kmers * 9 + L * 3
    if shutil.disk_usage(genome["dir"]).free < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of scratch disk for the large database")
    N_SAMPLES = 32
    rng = np.random.default_rng(7)
    t_all = time.time()
    # ---- the large database: the cohort's first sample genome + 2 G random k-mers -------------------------------------
    big = os.path.join(genome["dir"], "big")
    t0 = time.time()
    subprocess.run([SYNTH, "--out", big, "--contig-lens", ",".join(map(str, LENS)), "--seed", "3", "--sample-seed", "21",
                    "--noise", str(big_kmers)], check=True, capture_output=True, text=True)
    t_gen = time.time() - t0
    # ---- the reference ------------------------------------------------------------------------------------------------
    # ONE genome of L bases, not the small one tiled (round 5): kcfx_synth's reference is a function of (seed, position), so the
    # fixture's 119 Mb genome (seed 3) is the first 119 Mb of the seed's 14.5 Gb genome, and a sample of any other stretch of it is
    # `--ref-offset`. N-runs, soft-masked blocks and IUPAC codes as everywhere.
    big_ref = os.path.join(genome["dir"], "huge")
    subprocess.run([SYNTH, "--out", big_ref, "--ref-len", str(L), "--seed", "3", "--ref-only"], check=True, capture_output=True, text=True)
    tbytes = np.fromfile(big_ref + ".ref.txt", dtype=np.uint8).tobytes()
    os.remove(big_ref + ".ref.txt")
    assert tbytes[:len(genome["text"])] == genome["text"]
    W = 50_000
    starts = np.arange(0, L - K + 1, W - K + 1, dtype=np.int64)
    lens = np.minimum(W, L - starts).astype(np.int32)
    starts, lens = starts[lens >= K], lens[lens >= K]
    # 32 samples: the eight cohort databases, 23 more samples of the same genome (their own mutations; ~100 M k-mers, 0.85 GB each)
    # where the scratch space takes them -- else the eight in rotation, and the log line says so -- and the large one last
    prefixes = list(cohort)
    extra_need = (N_SAMPLES - 1 - 8) * int(1.0e9) + (4 << 30)
    distinct = shutil.disk_usage(genome["dir"]).free > extra_need and not os.environ.get("KCFX_ROTATE_SAMPLES")
    t0 = time.time()
    small = sum(LENS)
    region = {s_: 0 for s_ in range(8)}                                  # where in the reference a sample's genome lies
    for i in range(8, N_SAMPLES - 1):
        if distinct:
            # its own stretch of the genome: 23 stretches spread over what lies behind the fixture's
            off = (small + (i - 8) * max(0, L - 2 * small) // 22) // 100000 * 100000
            pfx = os.path.join(genome["dir"], f"x{i}")
            subprocess.run([SYNTH, "--out", pfx, "--ref-len", str(min(small, L - off)), "--ref-offset", str(off), "--seed", "3",
                            "--sample-seed", str(100 + i), "--noise", "1000000"], check=True, capture_output=True, text=True)
            os.remove(pfx + ".ref.txt")
            prefixes.append(pfx)
            region[i] = off
        else:
            prefixes.append(cohort[i % 8])
            region[i] = 0
    t_gen_samples = time.time() - t0
    prefixes.append(big)
    BIG = N_SAMPLES - 1
    # ---- every GPU entry point the CLI uses: sharded over "two devices" (the one GPU twice), next database loaded during
    #      the current screening ---------------------------------------------------------------------------------------
    t0 = time.time()
    got, st = engine.screen_sharded(prefixes, K, tbytes, starts, lens, 1, devices=(0, 0))
    t_sharded = time.time() - t0
    assert got.shape == (N_SAMPLES, len(starts)) and st.shard_mode_used == engine.SHARD_SAMPLES
    n = int(got[0]["total_kmers"].sum())
    assert n > 0.95 * L
    if distinct:
        assert len({int(got[s]["obs"].sum()) for s in range(N_SAMPLES)}) == N_SAMPLES    # thirty-two different samples
    else:
        for i in range(8, N_SAMPLES - 1):                                               # the same database again: the same records
            for f in FIELDS:
                assert np.array_equal(got[i][f], got[i % 8][f]), (i, f)
        assert len({int(got[s]["obs"].sum()) for s in (0, 1, 2, 3, 4, 5, 6, 7, BIG)}) == 9   # nine different samples
    assert int(got[BIG]["obs"].sum()) > int(got[0]["obs"].sum()) * 0.5
    # partition identity at full size: every k-mer of the reference lies in exactly one tiling window (SURVEY a-1)
    assert n == int(got[BIG]["total_kmers"].sum()) and np.all(got[0]["total_kmers"] <= lens - K + 1)
    # loading overlapped screening: what the screening threads waited for is less than what the loads took
    assert st.seconds_exposed_load < st.seconds_db_open_sum
    # ---- ten of them through resident handles and kcfx_screen_ref_multi -------------------------------------------------
    ref = engine.Reference(tbytes, starts, lens, K)
    t0 = time.time()
    big_db = engine.KmcDb.open(big)
    t_big_open = time.time() - t0
    bi = big_db.info
    assert bi.total_kmers > big_kmers and bi.table_format == 1 and bi.table_lines >= 1 << 28
    # half a k-mer per cell, and 2 G k-mers share ~0.6 G distinct 16-mer minimizers: second choices and the stash carry weight
    assert 0 < bi.stash_kmers < 0.4 * bi.total_kmers and bi.second_choice_kmers > 0
    monkeypatch.setenv("KCFX_LOAD_STREAM_TWICE", "1")
    restreamed = engine.KmcDb.open(cohort[1])
    monkeypatch.delenv("KCFX_LOAD_STREAM_TWICE")
    dbs = [genome["db"]] + [engine.KmcDb.open(p) for p in cohort[1:]] + [restreamed, big_db]
    which = list(range(8)) + [1, BIG]
    t0 = time.time()
    multi = ref.screen_multi(dbs, 1)
    t_multi = time.time() - t0
    for j, s in enumerate(which):
        for f in FIELDS:
            assert np.array_equal(multi[j][f], got[s][f]), (s, f)

    def kernel_ms(db, steps=3):
        """the screening kernel alone (HIP events on the launch, as bench.py), per pass over the whole reference"""
        import torch
        out = torch.zeros(len(starts) * 10, dtype=torch.int32, device="cuda:0")
        stream = torch.cuda.current_stream()
        ref.screen_async(db, out.data_ptr(), stream.cuda_stream, 1)
        torch.cuda.synchronize()
        ref.prof_enable(True)
        ref.prof_read(reset=True)
        for _ in range(steps):
            ref.screen_async(db, out.data_ptr(), stream.cuda_stream, 1)
        torch.cuda.synchronize()
        ms, cnt = ref.prof_read(reset=True)
        ref.prof_enable(False)
        return ms / max(1, cnt)

    ms_sparse = kernel_ms(dbs[3])
    ms_big = kernel_ms(big_db)
    del dbs[1:9]                                            # free HBM for the other forms of the large database
    restreamed = None
    # ---- the large database as a 12-slot line table: >= 2^29 lines, which the multi-sample kernel refuses ---------------
    monkeypatch.setenv("KCFX_TABLE_FORMAT", "lines")
    monkeypatch.setenv("KCFX_TABLE_LOAD", "0.3")
    t0 = time.time()
    big_lines = engine.KmcDb.open(big)
    t_lines_open = time.time() - t0
    small_lines = engine.KmcDb.open(cohort[2])
    monkeypatch.delenv("KCFX_TABLE_FORMAT")
    monkeypatch.delenv("KCFX_TABLE_LOAD")
    assert big_lines.info.table_format == 0 and big_lines.info.table_lines >= 1 << 29
    pair = ref.screen_multi([small_lines, big_lines], 1)
    for f in FIELDS:
        assert np.array_equal(pair[0][f], got[2][f]) and np.array_equal(pair[1][f], got[BIG][f]), f
    del big_lines, small_lines, pair
    engine.release_cached()
    # ---- DENSE: the large database at ~1 k-mer per cell (what a wheat-size sample forces) ------------------------------------
    dense_bits = int(np.ceil(np.log2(bi.total_kmers / 16.0)))
    monkeypatch.setenv("KCFX_TABLE_LBITS", str(dense_bits))
    t0 = time.time()
    dense = engine.KmcDb.open(big, no_fallback=True)
    t_dense_open = time.time() - t0
    monkeypatch.delenv("KCFX_TABLE_LBITS")
    di = dense.info
    assert di.table_format == 1 and di.table_lines == 1 << dense_bits and di.total_kmers / (16.0 * di.table_lines) > 0.9
    dres = ref.screen(dense, 1)
    for f in FIELDS:
        assert np.array_equal(dres[f], got[BIG][f]), f      # every window, every field: the dense table changes nothing
    ms_dense = kernel_ms(dense)
    del dense
    # ---- 10 000 sampled windows (0.5 G k-mers) of four samples against the oracle, the large / dense one among them ----------
    t0 = time.time()
    n_checked = 0
    for s, pfx, n_check in ((0, prefixes[0], 3000), (3, prefixes[3], 3000), (29, prefixes[29], 3000), (BIG, big, 1000)):
        # (sampled over the whole reference + 400 windows of the stretch the sample was drawn from: elsewhere it has nothing to find)
        w0 = int(region.get(s, 0) // (W - K + 1))
        own = np.arange(w0, min(len(starts), w0 + 400))
        idx = np.unique(np.concatenate([rng.choice(len(starts), min(n_check - 400, len(starts)), replace=False), own, [0, len(starts) - 1]]))
        want = oracle.OracleDb(pfx).screen(tbytes, starts[idx], lens[idx], 1, threads=HOST_THREADS)
        for f in FIELDS:
            assert np.array_equal(want[f], got[s][f][idx]), (s, f)
        n_checked += len(idx)
    t_oracle = time.time() - t0
    gk = n / 1e9
    line = (f"config-5 ({why}): {L / 1e9:.1f} Gb reference (ONE genome, not tiled), {len(starts)} windows x {N_SAMPLES} samples ({'all distinct: 23 generated in %.0f s' % t_gen_samples if distinct else 'eight databases in rotation + the large one'}) = {N_SAMPLES * n / 1e9:.0f} G k-mer probes; "
            f"kcfx_screen_sharded (2 workers on one GPU, load overlapped): {t_sharded:.1f} s wall, loads {st.seconds_db_open_sum:.1f} s of "
            f"which {st.seconds_exposed_load:.1f} s exposed, screening {st.seconds_screen_sum:.2f} s; large database {bi.total_kmers / 1e9:.2f} G "
            f"k-mers generated in {t_gen:.0f} s, opened in {t_big_open:.1f} s ({bi.regions} x 2^{int(np.log2(bi.table_lines // bi.regions))} lines = "
            f"{bi.total_kmers / (16.0 * bi.table_lines):.2f} per cell, {bi.table_bytes / 2**30:.0f} GiB incl. stash, "
            f"{100.0 * bi.stash_kmers / bi.total_kmers:.0f} % in the stash; as a line table {t_lines_open:.1f} s); DENSE 2^{dense_bits} lines = "
            f"{di.total_kmers / (16.0 * di.table_lines):.2f} per cell: opened in {t_dense_open:.1f} s, {di.table_bytes / 2**30:.0f} GiB, "
            f"{100.0 * di.stash_kmers / di.total_kmers:.0f} % in the stash, {100.0 * di.second_choice_kmers / di.total_kmers:.0f} % second-choice, "
            f"every window identical; screening kernel per pass of {gk:.1f} G k-mers: sparse 150 M-k-mer sample {ms_sparse:.1f} ms "
            f"({gk / ms_sparse * 1e3:.0f} G k-mers/s), large at {bi.total_kmers / (16.0 * bi.table_lines):.2f} per cell {ms_big:.1f} ms "
            f"({gk / ms_big * 1e3:.0f} G/s), dense {ms_dense:.1f} ms ({gk / ms_dense * 1e3:.0f} G/s); kcfx_screen_ref_multi on 10 resident "
            f"databases: {t_multi * 1e3:.0f} ms = {10 * n / t_multi / 1e9:.0f} G k-mers/s; oracle on {n_checked} sampled windows of 4 samples "
            f"{t_oracle:.0f} s; test wall {time.time() - t_all:.0f} s")
    print(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "config5_shape.log"), "a") as fh:
        fh.write(line + "\n")


def _cgroup_memory_limit():
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(path).read().strip()
            if v != "max":
                return int(v)
        except (OSError, ValueError):
            pass
    return 0


@pytest.mark.parametrize("size", ["full", "fallback"])
def test_wheat_scale_genomic_database(oracle, monkeypatch, size):
    """A database at wheat scale made of GENOMIC k-mers (runs that share minimizers, as a real `kmc` count of a 14.5 Gb genome is),
    not of random ones: one non-tiled synthetic genome as large as this box's memory and scratch space allow (9.1 Gb = 8 G distinct
    31-mers where the cgroup grants ~300 GB; the size that ran is printed), its mutated sample counted into ONE KMC database (64 GB
    .kmc_suf), opened (a) as the size policy opens it -- ~0.5 k-mers per cell, 128 GiB of lines for 8 G k-mers -- and (b) at half the
    lines (~1 per cell, which a quarter of the k-mers pays for with a place in the stash: allowed to fail on the stash's size), and
    the genome's own 50 kb tiling screened against both. Both tables must give identical records on every window, >= 10 000
    sampled windows must equal the oracle's, and the log line carries what the review asked for: stash share, second- and
    third-line share, open time, k-mers/s of the pass. (KCFX_WHEAT=<bases> overrides the size; the inputs live in /dev/shm when
    the scratch disk is too small and are removed afterwards.)"""
    import shutil
    import tempfile
    import torch
    t_all = time.time()
    mem = _cgroup_memory_limit() or _mem_available_bytes()
    mem = min(mem, _mem_available_bytes() or mem)
    def decide():
        L = int(float(os.environ.get("KCFX_WHEAT", "9.1e9")))
        why = "KCFX_WHEAT" if os.environ.get("KCFX_WHEAT") else "8 G genomic k-mers"
        if not os.environ.get("KCFX_WHEAT") and mem and 30.0 * L > 0.85 * mem:    # the generator peaks at ~25 bytes per base, the files add 9 in /dev/shm
            L = int(0.85 * mem / 30.0)
            why = f"FALLBACK: {mem / 1e9:.0f} GB of memory granted"
        return L, why
    L, why = _size_case("wheat", size, decide)
    if L < 1.0e9:
        pytest.skip(f"needs ~30 bytes of host memory per base; {mem / 1e9:.0f} GB granted")
    need_disk = int(9.2 * L)
    base = None
    for cand in (tempfile.gettempdir(), "/dev/shm"):
        if os.path.isdir(cand) and shutil.disk_usage(cand).free > need_disk + (8 << 30):
            base = cand
            break
    if base is None:
        pytest.skip(f"needs {need_disk / 1e9:.0f} GB of scratch space")
    d = tempfile.mkdtemp(prefix="kcfx_wheat_", dir=base)
    try:
        prefix = os.path.join(d, "wheat")
        t0 = time.time()
        out = subprocess.run([SYNTH, "--out", prefix, "--ref-len", str(L), "--seed", "41", "--sample-seed", "42", "--noise", "0"],
                             check=True, capture_output=True, text=True).stdout
        meta = json.loads(out.strip().splitlines()[-1])
        t_gen = time.time() - t0
        n_db = int(meta["db_kmers"])
        text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
        os.remove(prefix + ".ref.txt")
        W = 50_000
        starts = np.arange(0, len(text) - K + 1, W - K + 1, dtype=np.int64)
        lens = np.minimum(W, len(text) - starts).astype(np.int32)
        starts, lens = starts[lens >= K], lens[lens >= K]
        engine.release_cached()
        ref = engine.Reference(text, starts, lens, K)
        stream = torch.cuda.current_stream()
        outbuf = torch.zeros(len(starts) * 10, dtype=torch.int32, device="cuda:0")

        # the file once through the page cache: what the generator left in /dev/shm reads at a third of the speed the first time
        # (13 GB/s against 45), and a database that is opened has normally been written some time before
        t0 = time.time()
        with open(prefix + ".kmc_suf", "rb") as fh:
            while fh.read(1 << 26):
                pass
        t_read = time.time() - t0

        def open_and_screen(lbits, again=False):
            if lbits:
                monkeypatch.setenv("KCFX_TABLE_LBITS", str(lbits))
            t1 = time.time()
            db = engine.KmcDb.open(prefix, no_fallback=True)
            t_open = time.time() - t1
            t_again = None

            def pass_ms(handle):
                ref.screen(handle, 1)
                ref.prof_enable(True)
                ref.prof_read(reset=True)
                for _ in range(2):
                    ref.screen_async(handle, outbuf.data_ptr(), stream.cuda_stream, 1)
                torch.cuda.synchronize()
                ms_, cnt_ = ref.prof_read(reset=True)
                ref.prof_enable(False)
                return ms_ / max(1, cnt_)
            ms_first = None
            if again:                       # ... and once more in this process: the table's memory is kept between the two (kcfx_pool.cpp)
                ms_first = pass_ms(db)
                # (a job that opens one large database after the other says so: since round 6 the library keeps at most half of the
                # device of what it has released unless asked -- a 206 GB table that goes back to the driver makes the next open wait
                # for the driver's clearing: 12.7 s instead of 7)
                prev_fraction = engine.set_cache_fraction(0.9)
                db.close()
                t1 = time.time()
                db = engine.KmcDb.open(prefix, no_fallback=True)
                t_again = time.time() - t1
                engine.set_cache_fraction(prev_fraction)
            if lbits:
                monkeypatch.delenv("KCFX_TABLE_LBITS")
            info = db.info
            assert info.table_format == 1 and info.total_kmers == n_db
            res = ref.screen(db, 1)
            ref.prof_enable(True)
            ref.prof_read(reset=True)
            for _ in range(2):
                ref.screen_async(db, outbuf.data_ptr(), stream.cuda_stream, 1)
            torch.cuda.synchronize()
            ms, cnt = ref.prof_read(reset=True)
            ref.prof_enable(False)
            d_ = dict(lines=int(info.table_lines), regions=int(info.regions), extended=int(info.extended), per_cell=info.stored_kmers / (16.0 * info.table_lines), gib=info.table_bytes / 2 ** 30,
                      stash=info.stash_kmers / max(1, info.stored_kmers), moved=info.second_choice_kmers / max(1, info.stored_kmers),
                      open_s=t_open, again_s=t_again, kernel_ms=ms / max(1, cnt), first_ms=ms_first)
            db.close()
            engine.release_cached()
            return res, d_

        res_policy, a = open_and_screen(0, again=True)
        n = int(res_policy["total_kmers"].sum())
        assert n > 0.97 * len(text)
        hit = res_policy["obs"].sum() / n
        assert 0.5 < hit < 0.85                                  # the sample is the genome at 1 % divergence with 10 % deleted
        res_dense, b, b_err = None, None, "not tried: the policy's table is already at ~1 per cell (other handles of this process hold memory)"
        if 16 * (a["lines"] // 2) >= n_db:
            try:
                res_dense, b = open_and_screen(int(np.log2(a["lines"])) - 1)
                for f in FIELDS:
                    assert np.array_equal(res_dense[f], res_policy[f]), f
            except engine.KcfxError as exc:                      # (a stash of two billion k-mers may not fit: said, not hidden)
                b_err = str(exc)
        # >= 10 000 sampled windows against the oracle (the C restatement bisects the 64 GB file itself)
        rng = np.random.default_rng(11)
        idx = np.unique(np.concatenate([rng.choice(len(starts), min(10_000, len(starts)), replace=False), [0, len(starts) - 1]]))
        t0 = time.time()
        want = oracle.OracleDb(prefix).screen(text, starts[idx], lens[idx], 1, threads=HOST_THREADS)
        t_oracle = time.time() - t0
        for f in FIELDS:
            assert np.array_equal(want[f], res_policy[f][idx]), f
        gk = n / 1e9

        def show(x):
            shape = f"2^{int(np.log2(x['lines']))}" if x["regions"] == 1 else f"{x['regions']} x 2^{int(np.log2(x['lines'] // x['regions']))}"
            return (f"{shape} lines ({'extended' if x['extended'] else 'plain'}) = {x['per_cell']:.2f} per cell, {x['gib']:.0f} GiB incl. stash, opened in {x['open_s']:.1f} s" + (f" (again in this process: {x['again_s']:.1f} s; pass on the first handle {x['first_ms']:.1f} ms)" if x['again_s'] else "") + ", "
                    f"{100 * x['stash']:.1f} % in the stash, {100 * x['moved']:.1f} % in second / third lines, pass {x['kernel_ms']:.1f} ms = "
                    f"{gk / x['kernel_ms'] * 1e3:.0f} G k-mers/s")
        line = (f"wheat-scale genomic database ({why}): {len(text) / 1e9:.2f} Gb non-tiled genome, {n_db / 1e9:.2f} G distinct 31-mers of its sample "
                f"({os.path.getsize(prefix + '.kmc_suf') / 1e9:.0f} GB .kmc_suf in {base}, generated in {t_gen:.0f} s, read once in {t_read:.0f} s), {len(starts)} windows of 50 kb, hit rate "
                f"{hit:.3f}; size policy: {show(a)}" + (f"; half the lines: {show(b)}, every window identical" if b else f"; half the lines: {b_err}") +
                f"; oracle on {len(idx)} sampled windows {t_oracle:.0f} s; test wall {time.time() - t_all:.0f} s")
        print(line)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "wheat_genomic.log"), "a") as fh:
            fh.write(line + "\n")
        # what the third lines are for: a genomic database at ~1 k-mer per cell used to leave a quarter to a third of itself in the stash
        # (which size the policy picks depends on what else this process holds on the device: 2^30 lines for 8 G k-mers when 60 % of
        # the free memory can take them, else 2^29)
        assert a["extended"] == 1                                # (the build found > 15 % of the k-mers outside their first-choice cell and started over)
        assert a["stash"] < (0.10 if a["per_cell"] < 0.6 else 0.22)
        # (3 x 2^29 lines: a pass takes 42.6 - 43.0 ms = 209 - 211 G k-mers/s in six of the round's ten runs of this test and 51.6 - 53.0 ms
        # = 169 - 174 G in the other four, on both handles of the process alike -- round 4's "two speeds of a table", 20 % at 206 GB: where
        # the driver happened to place the table, not what the kernel does; the bound is under the slow mode)
        if a["per_cell"] < 0.4:
            assert gk / a["kernel_ms"] * 1e3 > 150
    finally:
        shutil.rmtree(d, ignore_errors=True)
