#!/usr/bin/env python3
"""Regenerates the committed fixtures of tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference (Java 17 + Maven) cannot run in the build image and ships no test data, so these are NOT outputs of
the reference: inputs come from kcftools_amd/synth.py (seeded), expected outputs from oracle/kcf_oracle.c, and the
script refuses to write them unless the independent Python model (oracle/pymodel.py) agrees on every window.
Files: ref.fa (+ .faidx text it should index to), db_*.kmc_pre/.kmc_suf, expected.json.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kcftools_amd import synth  # noqa: E402
from oracle import binding as oracle, pymodel  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
FIELDS = ("total_kmers", "eff_length", "obs", "var", "inner", "left", "right", "kmer_count_sum")


def main():
    contigs = [("chrA", synth.random_reference(23_456, 11, n_run_every=4000, max_n_run=150, lower_frac=0.2, iupac_rate=1e-3)),
               ("chrB", synth.random_reference(9_031, 12, n_run_every=2500, max_n_run=60, lower_frac=0.1, iupac_rate=2e-3)),
               ("scaffold_3 extra description", synth.random_reference(1_207, 13, n_run_every=0, lower_frac=0.5, iupac_rate=0.0))]
    synth.write_fasta(os.path.join(OUT, "ref.fa"), contigs, line_bases=60)
    text = b"".join(seq for _, seq in contigs)
    dbs = {
        "db_canon_k31_p3_cs2_m9": synth.KmcParams(k=31, p=3, counter_size=2, sig_len=9, n_bins=4, both_strands=True),
        "db_fwd_k31_p7_cs1_m7": synth.KmcParams(k=31, p=7, counter_size=1, sig_len=7, n_bins=3, both_strands=False, max_count=255),
        "db_canon_k23_p3_cs4_m5": synth.KmcParams(k=23, p=3, counter_size=4, sig_len=5, n_bins=5, both_strands=True, max_count=0x7FFFFFFF),
    }
    expected = {"contigs": [[n.split(" ")[0], len(s)] for n, s in contigs], "cases": []}
    for i, (name, prm) in enumerate(dbs.items()):
        sample = synth.mutate_sample(text, 40 + i, sub_rate=1.2e-2, indel_rate=1.5e-3, big_del_per_mb=150, big_del_len=1200)
        kmers, counts = synth.build_sample_db(sample, prm, 50 + i, noise_kmers=400)
        prefix = os.path.join(OUT, name)
        synth.write_kmc(prefix, kmers, counts, prm)
        odb = oracle.OracleDb(prefix)
        table, hdr = pymodel.read_kmc(prefix)
        for window, step, min_count in ((2_000, 0, 1), (5_000, 1_250, 2), (700, 0, 1)):
            wb, wl, ids = [], [], []
            off = 0
            for cname, seq in contigs:
                s, e = synth.tiling_windows(len(seq), window, prm.k, step)
                wb += list(s + off)
                wl += list(e - s)
                ids += [f"{cname.split(' ')[0]}_{x}" for x in s]
                off += len(seq)
            res = odb.screen(text, wb, wl, min_count, threads=2)
            for j in range(len(wb)):
                pm = pymodel.process_window(text[wb[j]:wb[j] + wl[j]].decode("latin1"), table, prm.k, prm.both_strands, min_count)
                for f in FIELDS:
                    assert pm[f] == int(res[f][j]), (name, window, j, f)
            sc, md = oracle.score_array(res)
            expected["cases"].append({
                "db": name, "k": prm.k, "window": window, "step": step, "min_count": min_count,
                "win_begin": [int(x) for x in wb], "win_len": [int(x) for x in wl], "ids": ids,
                **{f: [int(x) for x in res[f]] for f in FIELDS},
                "score": [float(x) for x in sc], "mean_count": [float(x) for x in md]})
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(expected, f)
    print("wrote", len(expected["cases"]), "cases;", sum(len(c["ids"]) for c in expected["cases"]), "windows")


if __name__ == "__main__":
    main()
