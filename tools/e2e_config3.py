#!/usr/bin/env python3
"""End-to-end wall of BASELINE configs[3] on one GPU: the 119 Mb five-contig genome, 10 kb windows, EIGHT samples in one
`kcftools-amd getVariations` run (FASTA read -> 8 databases loaded and screened -> one 8-column KCF written).

    python tools/e2e_config3.py [--workdir /tmp/kcfx_e2e3] [--runs 3]"""
import argparse
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENS = [30_427_671, 19_698_289, 23_459_830, 18_585_056, 26_975_502]      # TAIR10 chromosome lengths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workdir", default="/tmp/kcfx_e2e3")
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--noise", type=int, default=44_000_000, help="random k-mers per sample beside the ~100 M genomic ones")
    args = ap.parse_args()
    os.makedirs(args.workdir, exist_ok=True)
    synth = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
    cli = os.path.join(ROOT, "kcftools_amd", "bin", "kcftools-amd")
    prefixes = []
    t0 = time.time()
    for i in range(8):
        pfx = os.path.join(args.workdir, f"s{i}")
        if not os.path.exists(pfx + ".kmc_suf"):
            subprocess.run([synth, "--out", pfx, "--contig-lens", ",".join(map(str, LENS)), "--seed", "3", "--sample-seed", str(20 + i),
                            "--noise", str(args.noise)] + (["--fasta"] if i == 0 else []), check=True, capture_output=True, text=True)
        prefixes.append(pfx)
    print(f"inputs ready in {time.time() - t0:.0f} s: 8 x {os.path.getsize(prefixes[0] + '.kmc_suf') / 1e9:.2f} GB .kmc_suf", flush=True)
    out = os.path.join(args.workdir, "cohort.kcf")
    for r in range(args.runs):
        t = time.time()
        p = subprocess.run([cli, "getVariations", "-r", prefixes[0] + ".fa", "-k", ",".join(prefixes), "-o", out,
                            "-s", ",".join(f"S{i}" for i in range(8)), "-f", "window", "-w", "10000"], capture_output=True, text=True)
        wall = time.time() - t
        if p.returncode:
            print(p.stdout[-800:], p.stderr[-800:])
            sys.exit(1)
        m = re.search(r"Screened.*", p.stdout)
        rows = sum(1 for ln in open(out) if not ln.startswith("#"))
        print(f"run {r}: {wall:.2f} s wall, {rows} rows x 8 samples = {rows * 8 * 9970 / wall / 1e9:.2f} G k-mers/s end to end; {m.group(0) if m else ''}", flush=True)


if __name__ == "__main__":
    main()
