#!/usr/bin/env python3
"""BASELINE configs[2] as a NUMBER (run on the GPU box): the 119 Mb five-contig genome, 27 000 GTF gene models (merged, start-sorted
exon loci spliced without separator: GTF.java:223-293), one 150 M-k-mer sample. Screening-kernel time and k-mers/s on the gene
windows against the 10 kb tiling of the same genome and database, with the shape of the tile plan (tiles per window, words per
wave). Prints one JSON object.

    python tools/genes_probe.py [--workdir /tmp/kcfx_genes] [--steps 100]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kcftools_amd import engine, synth  # noqa: E402
from oracle import pyhost               # noqa: E402  (GTF model + .faidx arithmetic: test infrastructure, used here to cut the windows)

LENS = [30_427_671, 19_698_289, 23_459_830, 18_585_056, 26_975_502]
K = 31
TILE_K = 2048


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workdir", default="/tmp/kcfx_genes")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    import torch
    os.makedirs(args.workdir, exist_ok=True)
    prefix = os.path.join(args.workdir, "ath")
    synth_bin = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
    if not os.path.exists(prefix + ".kmc_suf"):
        subprocess.run([synth_bin, "--out", prefix, "--contig-lens", ",".join(map(str, LENS)), "--seed", "3", "--sample-seed", "4",
                        "--noise", "44000000", "--fasta"], check=True, capture_output=True, text=True)
    contigs = [(f"chr{i + 1}", L) for i, L in enumerate(LENS)]
    gtf = os.path.join(args.workdir, "genes.gtf")
    synth.random_gtf(gtf, contigs, 11, genes_per_mb=227.0)
    model = pyhost.Gtf(gtf)
    fa = open(prefix + ".fa", "rb").read()
    entries = {e[0]: e for e in pyhost.generate_faidx(prefix + ".fa")}
    pieces, wb, wl, off = [], [], [], 0
    for c, _ in contigs:
        for g in model.kids(c):
            text = b"".join(pyhost.get_sequence(fa, entries[l[0]], l[1] - 1, l[2] - l[1] + 1) for l in model.merged_exon_loci(g, True))
            pieces.append(text)
            wb.append(off)
            wl.append(len(text))
            off += len(text)
    gene_text = b"".join(pieces)
    wb = np.array(wb, dtype=np.int64)
    wl = np.array(wl, dtype=np.int32)
    whole = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    tb, tl, o = [], [], 0
    for L in LENS:
        s, e = engine.windows(L, 10_000, 0, K)
        tb.append(s.astype(np.int64) + o)
        tl.append((e - s).astype(np.int32))
        o += L
    tb, tl = np.concatenate(tb), np.concatenate(tl)

    stream = torch.cuda.current_stream()

    def measure(text, b, l):
        ref = engine.Reference(text, b, l, K)
        db = engine.KmcDb.open(prefix, expected_queries=int(ref.n_kmer_slots))
        out = torch.zeros(len(b) * 10, dtype=torch.int32, device="cuda:0")
        for _ in range(200):                                    # clock ramp + warm-up
            ref.screen_async(db, out.data_ptr(), stream.cuda_stream, 1)
        torch.cuda.synchronize()
        ref.prof_enable(True)
        ref.prof_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ref.screen_async(db, out.data_ptr(), stream.cuda_stream, 1)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps
        ms, n = ref.prof_read(reset=True)
        res = out.cpu().numpy().view(engine.RESULT_DTYPE)
        kmers = int(res["total_kmers"].sum())
        nk = np.maximum(0, l.astype(np.int64) - K + 1)
        tiles = -(-nk // TILE_K)
        per = np.where(tiles > 0, -(-nk // np.maximum(tiles, 1)), 0)
        words = -(-per // 64)
        d = {"windows": int(len(b)), "text_bases": int(len(text)), "kmers": kmers, "kernel_ms": ms / n, "step_ms": wall * 1e3,
             "G_kmers_per_s_kernel": kmers / (ms / n * 1e-3) / 1e9, "tiles": int(tiles.sum()), "tiles_per_window_mean": float(tiles[nk > 0].mean()),
             "kmers_per_tile_mean": float(nk.sum() / max(1, tiles.sum())), "words_per_wave_mean": float((np.ceil(words / 2) * tiles).sum() / max(1, tiles.sum())),
             "windows_with_one_word_or_less": int((nk <= 64).sum()), "window_len_quantiles_5_50_95": [int(x) for x in np.percentile(l, [5, 50, 95])],
             "table_lines": int(db.info.table_lines)}
        db.close()
        ref.close()
        return d

    genes = measure(gene_text, wb, wl)
    tiling = measure(whole, tb, tl)
    print(json.dumps({"genes": genes, "tiling_10kb": tiling,
                      "genes_rate_over_tiling_rate": genes["G_kmers_per_s_kernel"] / tiling["G_kmers_per_s_kernel"]}, indent=1))


if __name__ == "__main__":
    main()
