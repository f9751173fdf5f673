"""End-to-end: the `kcftools-amd getVariations` driver (option set of GetVariants.java:21-61) on the golden FASTA + KMC
database, against KCF text assembled from the oracle's integers with the Python restatement of the KCF writer.
Everything except the ##date and ##CMD header lines must be byte-identical."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from kcftools_amd import synth
from oracle import pyhost

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "kcftools_amd", "bin", "kcftools-amd")
DB = "db_canon_k31_p3_cs2_m9"


def run_cli(args, expect_rc=0):
    p = subprocess.run([CLI, "getVariations"] + args, capture_output=True, text=True)
    assert p.returncode == expect_rc, p.stdout + p.stderr
    return p.stdout


def strip_volatile(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("##date=") and not l.startswith("##CMD="))


def expected_kcf(oracle, fa, windows, sample, window, step, min_count, weights=(0.3, 0.3, 0.4)):
    """windows: list of (seq, start, end, id, text_bytes) in output order"""
    odb = oracle.OracleDb(os.path.join(G, DB))
    entries = pyhost.generate_faidx(fa)
    rows = []
    for seq, s, e, wid, text in windows:
        res = odb.process_window(text, min_count)
        sc, md, _ = oracle.score(res, *weights)
        rows.append(pyhost.kcf_row(seq, s, e, wid, res, sc, md))
    hdr = pyhost.kcf_header(fa, [(n, l) for n, l, *_ in entries], sample, window, step, 31, len(windows), *weights, "x", "d")
    return strip_volatile(hdr + "\n".join(rows) + "\n")


def test_window_mode(oracle, tmp_path):
    fa = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(G, "ref.fa"), fa)
    data = open(fa, "rb").read()
    entries = pyhost.generate_faidx(fa)
    for window, step, mc in ((2000, 0, 1), (5000, 1250, 2)):
        out = str(tmp_path / f"o_{window}_{step}.kcf")
        args = ["-r", fa, "-k", os.path.join(G, DB), "-o", out, "-s", "my/sample", "-f", "window", "-w", str(window), "-c", str(mc)]
        if step:
            args += ["-p", str(step)]
        log = run_cli(args)
        assert "changed to: my_sample" in log
        wins = []
        for ent in entries:
            s, e = synth.tiling_windows(ent[1], window, 31, step)
            for a, b in zip(s, e):
                wins.append((ent[0], int(a), int(b), f"{ent[0]}_{a}", pyhost.get_sequence(data, ent, int(a), int(b - a))))
        assert strip_volatile(open(out).read()) == expected_kcf(oracle, fa, wins, "my_sample", window, step, mc)
        head = open(out).read().split("\n")
        assert head[0] == "##format=KCF0.4.0" and any(l.startswith("##CMD=kcftools-amd getVariations -r ") for l in head)


@pytest.mark.parametrize("feature", ["gene", "transcript"])
def test_gene_and_transcript_mode(oracle, tmp_path, feature):
    fa = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(G, "ref.fa"), fa)
    data = open(fa, "rb").read()
    entries = pyhost.generate_faidx(fa)
    gtf = str(tmp_path / "a.gtf")
    synth.random_gtf(gtf, [(e[0], e[1]) for e in entries], 9, genes_per_mb=1500, with_gene_lines=(feature == "gene"))
    out = str(tmp_path / "o.kcf")
    run_cli(["-r", fa, "-k", os.path.join(G, DB), "-o", out, "-s", "S", "-f", feature, "-g", gtf, "--wi", "0.25", "--wt", "0.25", "--wr", "0.5"])
    model = pyhost.Gtf(gtf)
    by_name = {e[0]: e for e in entries}
    wins = []
    for ent in entries:
        cw = []
        for gene in model.kids(ent[0]):
            for fid in ([gene] if feature == "gene" else model.kids(gene)):
                c, s, e, _ = model.loci(fid)
                loci = model.merged_exon_loci(fid, feature == "gene")
                text = b"".join(pyhost.get_sequence(data, by_name[l[0]], l[1] - 1, l[2] - l[1] + 1) for l in loci)
                cw.append((c, s, e, fid, text))
        cw.sort(key=lambda w: w[1])            # GetVariants.java:169-171 (stable; ties keep generation order here)
        wins += cw
    assert len(wins) > 20
    assert strip_volatile(open(out).read()) == expected_kcf(oracle, fa, wins, "S", 0, 0, 1, (0.25, 0.25, 0.5))


def test_cli_validation_errors(tmp_path):
    fa = os.path.join(G, "ref.fa")
    base = ["-r", fa, "-k", os.path.join(G, DB), "-o", str(tmp_path / "x.kcf"), "-s", "S"]
    for extra, msg in ((["-f", "window"], "Window size is required"), (["-f", "window", "-w", "1000", "-g", "a.gtf"], "GTF file is not valid"),
                       (["-f", "gene"], "GTF file is required"), (["-f", "gene", "-g", "a.gtf", "-w", "5"], "Window size is not valid"),
                       (["-f", "exon", "-w", "10"], "Invalid model type"), (["-f", "window", "-w", "1000", "-c", "0"], "at least 1"),
                       (["-f", "window", "-w", "1000", "-t", "0"], "greater than 0"), (["-f", "window", "-w", "20"], "window size must exceed"),
                       (["-f", "window", "-w", "1000", "--wi", "0.5"], "Weights should sum to 1.0")):
        log = run_cli(base + extra, expect_rc=1)
        assert msg in log and " - ERROR " in log
    log = run_cli(["-r", fa, "-k", str(tmp_path / "nodb"), "-o", str(tmp_path / "x.kcf"), "-s", "S", "-f", "window", "-w", "1000"], expect_rc=1)
    assert "Error reading prefix file" in log


def test_multi_sample_equals_cohort_of_single_sample_runs(oracle, tmp_path):
    """`-k a,b -s x,y` screens the resident reference against every sample and writes what `cohort`
    (Plugins/Cohort.java) would produce from the single-sample KCFs: KD re-derived from its 2-decimal text, scores from the
    integers, min / max / mean statistics over the samples (float accumulators, Float.MIN_VALUE start of XS)."""
    fa = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(G, "ref.fa"), fa)
    data = open(fa, "rb").read()
    entries = pyhost.generate_faidx(fa)
    dbs = ["db_canon_k31_p3_cs2_m9", "db_fwd_k31_p7_cs1_m7"]
    out = str(tmp_path / "multi.kcf")
    run_cli(["-r", fa, "-k", ",".join(os.path.join(G, d) for d in dbs), "-o", out, "-s", "S1,S2", "-f", "window", "-w", "3000", "-c", "2"])
    singles = []
    for i, d in enumerate(dbs):
        o1 = str(tmp_path / f"single{i}.kcf")
        run_cli(["-r", fa, "-k", os.path.join(G, d), "-o", o1, "-s", f"S{i + 1}", "-f", "window", "-w", "3000", "-c", "2"])
        singles.append([l.rstrip("\n").split("\t") for l in open(o1) if not l.startswith("#")])
    want_rows = []
    for r0, r1 in zip(*singles):
        assert r0[:5] == r1[:5]
        total = int(r0[4])
        eff = int(r0[5].split(";")[0].split("=")[1])
        cols = [pyhost.cohort_sample_from_text(r[7], total, eff, 0.3, 0.3, 0.4) for r in (r0, r1)]
        want_rows.append(pyhost.cohort_row(r0[0], int(r0[1]), int(r0[2]), r0[3], total, eff, cols))
    got = open(out).read().split("\n")
    got_rows = [l for l in got if l and not l.startswith("#")]
    assert got_rows == want_rows and len(got_rows) > 10
    assert [l for l in got if l.startswith("#CHROM")][0].endswith("\tFORMAT\tS1\tS2")
    # and the single-sample files themselves are the oracle's numbers (spot check of the first sample)
    odb = oracle.OracleDb(os.path.join(G, dbs[0]))
    ent = entries[0]
    s, e = synth.tiling_windows(ent[1], 3000, 31)
    res = odb.process_window(pyhost.get_sequence(data, ent, int(s[2]), int(e[2] - s[2])), 2)
    sc, md, _ = oracle.score(res)
    assert "\t".join(singles[0][2]) == pyhost.kcf_row(ent[0], int(s[2]), int(e[2]), f"{ent[0]}_{s[2]}", res, sc, md)
    log = run_cli(["-r", fa, "-k", os.path.join(G, dbs[0]), "-o", out, "-s", "A,B", "-f", "window", "-w", "3000"], expect_rc=1)
    assert "differ" in log


def test_cohort_table_option_gives_the_same_file(tmp_path):
    """`--cohort-table` (= `--shard cohort`): the samples in ONE table per group of eight (kcfx_cohort_open), one look-up per
    reference k-mer for the group, tables x window ranges over the listed devices -- the multi-sample KCF must be the file the
    per-sample tables give, byte for byte (##CMD / ##date aside), on one worker and on three. A pair that cannot share a table
    (mixed strandedness) is screened sample by sample inside the same call: same file again."""
    fa = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(G, "ref.fa"), fa)
    ref = open(os.path.join(G, "ref.fa"), "rb").read()
    text = b"".join(l for l in ref.split(b"\n") if not l.startswith(b">"))
    prm = synth.KmcParams(k=31, p=3, counter_size=2, sig_len=7, n_bins=4, max_count=65535)
    prefixes = []
    for i in range(3):
        km, ct = synth.build_sample_db(synth.mutate_sample(text, 40 + i), prm, 50 + i, noise_kmers=500)
        pfx = str(tmp_path / f"c{i}")
        synth.write_kmc(pfx, km, ct, prm)
        prefixes.append(pfx)
    outs = []
    def body(path):
        return [l for l in open(path).read().split("\n") if not l.startswith("##CMD") and not l.startswith("##date")]
    for extra in ([], ["--cohort-table"], ["--shard", "cohort", "--devices", "0,0,0"]):
        out = str(tmp_path / f"o{len(outs)}.kcf")
        log = run_cli(["-r", fa, "-k", ",".join(prefixes), "-o", out, "-s", "A,B,C", "-f", "window", "-w", "3000"] + extra)
        assert ("cohort-table x window-sharded" in log) == bool(extra)
        outs.append(body(out))
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 20
    mixed = ["-r", fa, "-k", prefixes[0] + "," + os.path.join(G, "db_fwd_k31_p7_cs1_m7"), "-s", "A,B", "-f", "window", "-w", "3000"]
    run_cli(mixed + ["-o", str(tmp_path / "x0.kcf")])
    run_cli(mixed + ["-o", str(tmp_path / "x1.kcf"), "--cohort-table"])
    assert body(str(tmp_path / "x0.kcf")) == body(str(tmp_path / "x1.kcf"))


def test_samples_sharded_over_devices_give_the_same_file(tmp_path):
    """`--devices a,b,...` shards the samples over GPUs (sample s on device s mod G, one worker thread each, own resident
    reference, no exchange). A one-GPU box can only check the threading: `--devices 0,0,0` runs three workers against the
    same GPU concurrently and must write byte-identical output (header lines with the command line aside)."""
    fa = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(G, "ref.fa"), fa)
    dbs = ["db_canon_k31_p3_cs2_m9", "db_fwd_k31_p7_cs1_m7", "db_canon_k31_p3_cs2_m9", "db_fwd_k31_p7_cs1_m7",
           "db_canon_k31_p3_cs2_m9"]
    k = ",".join(os.path.join(G, d) for d in dbs)
    outs = []
    # (... and `--shard table`: every sample's TABLE cut into one slice per listed device by minimizer, every device screens every
    # window against its slice, the hit bitmaps merged -- the same file again)
    for i, dev in enumerate((["--device", "0"], ["--devices", "0,0,0"], ["--devices", "0,0"], ["--devices", "0,0,0", "--shard", "table"],
                             ["--device", "0", "--shard", "table"])):
        out = str(tmp_path / f"m{i}.kcf")
        run_cli(["-r", fa, "-k", k, "-o", out, "-s", "A,B,C,D,E", "-f", "window", "-w", "3000"] + dev)
        outs.append([l for l in open(out) if not l.startswith("##CMD") and not l.startswith("##date") and not l.startswith("##fileDate")])
    assert all(o == outs[0] for o in outs[1:]) and sum(1 for l in outs[0] if not l.startswith("#")) > 10


def test_pin_helper_accepts_the_driver_output_for_every_golden_case(tmp_path):
    """tools/pin_against_jar.sh compares the REAL kcftools jar's KCF with tests/golden/expected.json through
    tools/kcf_vs_expected.py. No JVM exists here, but the helper itself can be exercised: the KCF our own driver writes for
    each of the nine golden cases (three databases -- canonical k=31, forward-only k=31, canonical k=23 -- times three window
    geometries incl. -p and -c) must pass it, and a corrupted row must not."""
    import json
    helper = os.path.join(ROOT, "tools", "kcf_vs_expected.py")
    exp_path = os.path.join(G, "expected.json")
    fa = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(G, "ref.fa"), fa)
    cases = json.load(open(exp_path))["cases"]
    assert len(cases) == 9
    for idx, c in enumerate(cases):
        out = str(tmp_path / f"case{idx}.kcf")
        args = ["-r", fa, "-k", os.path.join(G, c["db"]), "-o", out, "-s", "S", "-f", "window", "-w", str(c["window"]), "-c", str(c["min_count"])]
        if c["step"]:
            args += ["-p", str(c["step"])]
        run_cli(args)
        p = subprocess.run(["python3", helper, out, exp_path, str(idx)], capture_output=True, text=True)
        assert p.returncode == 0, (idx, p.stdout)
    lines = open(out).read().split("\n")
    row = next(i for i, l in enumerate(lines) if l and not l.startswith("#"))
    f = lines[row].split("\t")
    f[4] = str(int(f[4]) + 1)
    lines[row] = "\t".join(f)
    open(out, "w").write("\n".join(lines))
    assert subprocess.run(["python3", helper, out, exp_path, str(idx)], capture_output=True, text=True).returncode == 1
