#!/usr/bin/env python3
"""How does the cell table behave on a database that looks like a k-mer count of sequencing READS (kmc -ci1, the
reference's quick-start) rather than the uniform-noise database of the bench? Read errors add several times more distinct
k-mers than the genome has; they occur once and share minimizers (and cells) with their genomic neighbours.

    python tools/realistic_db_probe.py [--ref-len 30427671] [--error-events 4500000] [--noise 0] [--steps 200]

Generates the database with kcfx_synth --error-events, opens it on cuda:0, prints the table statistics and the screening
rate (kernel time by HIP events, as bench.py), and checks every window against the oracle."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-len", type=int, default=30_427_671)
    ap.add_argument("--error-events", type=int, default=4_500_000)
    ap.add_argument("--noise", type=int, default=0)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--window", type=int, default=10_000)
    ap.add_argument("--workdir", default="/tmp/kcfx_probe")
    ap.add_argument("--min-count", type=int, default=1)
    ap.add_argument("--k", type=int, default=31, help="27 .. 31 take the cell table; (k - p) must be a multiple of 4")
    ap.add_argument("--p", type=int, default=0, help="LUT prefix length (default: 7, or the largest of 3 .. 7 that fits k)")
    ap.add_argument("--repeats", type=float, default=0.0, help="kcfx_synth --repeats: fraction of the genome with repeat structure")
    args = ap.parse_args()
    K = args.k
    p = args.p or max(q for q in range(3, 8) if (K - q) % 4 == 0)
    os.makedirs(args.workdir, exist_ok=True)
    synth = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
    prefix = os.path.join(args.workdir, f"r{args.ref_len}_e{args.error_events}_n{args.noise}_k{K}" + (f"_rep{args.repeats}" if args.repeats else ""))
    if not os.path.exists(prefix + ".kmc_suf"):
        out = subprocess.run([synth, "--out", prefix, "--ref-len", str(args.ref_len), "--seed", "1", "--sample-seed", "2",
                              "--error-events", str(args.error_events), "--noise", str(args.noise), "--k", str(K), "--p", str(p)]
                             + (["--repeats", str(args.repeats)] if args.repeats else []),
                             check=True, capture_output=True, text=True).stdout
        meta = json.loads(out.strip().splitlines()[-1])
        print("generated:", {k: meta[k] for k in ("sample_kmers", "noise_kmers", "error_kmers", "db_kmers")}, flush=True)
    import torch
    from kcftools_amd import engine
    from oracle import binding as oracle
    oracle.build()
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    starts, ends = engine.windows(len(text), args.window, 0, K)
    wb, wl = starts.astype(np.int64), (ends - starts).astype(np.int32)
    t0 = time.time()
    db = engine.KmcDb.open(prefix)
    i = db.info
    print(f"kcfx_db_open {time.time() - t0:.2f} s; table_format {i.table_format}, 2^{int(np.log2(max(1, i.table_lines)))} lines, "
          f"{i.table_bytes / 2**30:.1f} GiB incl. stash {i.stash_bytes / 2**30:.2f} GiB; {i.total_kmers / 1e6:.1f} M k-mers: "
          f"{i.second_choice_kmers / 1e6:.2f} M second-choice, {i.stash_kmers / 1e6:.2f} M in the stash "
          f"({100.0 * i.stash_kmers / max(1, i.total_kmers):.1f} %)", flush=True)
    ref = engine.Reference(text, wb, wl, K)
    out = torch.zeros(len(wb) * 10, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream()
    for _ in range(5):
        ref.screen_async(db, out.data_ptr(), st.cuda_stream, args.min_count)
    torch.cuda.synchronize()
    ref.prof_enable(True)
    ref.prof_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ref.screen_async(db, out.data_ptr(), st.cuda_stream, args.min_count)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, n = ref.prof_read(reset=True)
    res = out.cpu().numpy().view(engine.RESULT_DTYPE)[:len(wb)]
    nk = int(res["total_kmers"].sum())
    print(f"{nk / 1e6:.1f} M k-mers per pass, hit rate {res['obs'].sum() / nk:.3f}: kernel {ms / n:.4f} ms = "
          f"{nk / (ms / n * 1e-3) / 1e9:.1f} G k-mers/s, step {el / args.steps * 1e3:.4f} ms; "
          f"algorithmic roofline frac {nk * 11 / (ms / n * 1e-3) / 8e12:.3f}", flush=True)
    t0 = time.time()
    want = oracle.OracleDb(prefix).screen(text, wb, wl, args.min_count, threads=os.cpu_count() or 8)
    ok = all(np.array_equal(res[f], want[f]) for f in engine.INT_FIELDS)
    print(f"oracle on all {len(wb)} windows ({time.time() - t0:.1f} s): {'bit-exact' if ok else 'MISMATCH'}")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
