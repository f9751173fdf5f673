#!/usr/bin/env python3
"""Cell-table density sweep on the bench workload (BASELINE configs[1]): what a denser table costs the screening kernel.

    python tools/density_sweep.py [--lbits 27,26,25,24,23] [--steps 200] [--csv gpurun_out/r03_density.csv]
    python tools/density_sweep.py --one 24 --steps 30          # one density, for a rocprofv3 --pmc pass around it

For every table size 2^L lines (KCFX_TABLE_LBITS, read by kcfx_db_open) the bench database is opened, every window is
checked against the first size's result (and the first against the oracle), and the screening kernel is timed by the HIP
events bench.py uses. Columns: L, k-mers per cell, table bytes, stash bytes, second-choice %, stash %, kcfx_db_open seconds,
kernel ms, step ms, G k-mers/s, algorithmic roofline fraction. HBM requests per launch come from the --pmc passes
(tools/density_pmc.sh) and are merged in by --merge-pmc."""
import argparse
import csv
import glob
import json
import os
os.environ.setdefault("KCFX_TEST_HOOKS", "1")     # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pmc_requests(directory, kernel="screen_k31c_kernel"):
    vals = {}
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lbits", default="27,26,25,24,23")
    ap.add_argument("--one", type=int, default=0)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--csv", default="gpurun_out/r03_density.csv")
    ap.add_argument("--merge-pmc", default="", help="directory holding pmc_L<L>/ outputs of tools/density_pmc.sh")
    ap.add_argument("--workdir", default=os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"))
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()

    import bench
    import torch
    from kcftools_amd import engine
    synth_bin = bench.ensure_built()
    prefix, meta = bench.make_dataset(synth_bin, args.workdir, 30_427_671, 150_000_000, 2, os.cpu_count() or 8)
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    K = 31
    starts, ends = engine.windows(len(text), 10_000, 0, K)
    wb, wl = starts.astype(np.int64), (ends - starts).astype(np.int32)
    ref = engine.Reference(text, wb, wl, K)
    out = torch.zeros(len(wb) * 10, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream()
    Ls = [args.one] if args.one else [int(x) for x in args.lbits.split(",")]
    rows = []
    first = None
    for L in Ls:
        os.environ["KCFX_TABLE_LBITS"] = str(L)
        t0 = time.time()
        db = engine.KmcDb.open(prefix)
        t_open = time.time() - t0
        i = db.info
        for _ in range(10):
            ref.screen_async(db, out.data_ptr(), st.cuda_stream, 1)
        torch.cuda.synchronize()
        ref.prof_enable(True)
        ref.prof_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ref.screen_async(db, out.data_ptr(), st.cuda_stream, 1)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms, n = ref.prof_read(reset=True)
        ref.prof_enable(False)
        res = out.cpu().numpy().view(engine.RESULT_DTYPE)[:len(wb)].copy()
        nk = int(res["total_kmers"].sum())
        if first is None:
            first = res
            if not args.no_oracle and not args.one:
                from oracle import binding as oracle
                oracle.build()
                want = oracle.OracleDb(prefix).screen(text, wb, wl, 1, threads=os.cpu_count() or 8)
                assert all(np.array_equal(res[f], want[f]) for f in engine.INT_FIELDS), "GPU differs from the oracle"
        else:
            assert all(np.array_equal(res[f], first[f]) for f in engine.INT_FIELDS), f"L={L} differs from L={Ls[0]}"
        kms = ms / max(1, n)
        row = {"lbits": int(np.log2(max(1, i.table_lines))), "kmers_per_cell": i.total_kmers / (16.0 * i.table_lines),
               "table_bytes": int(i.table_bytes), "stash_bytes": int(i.stash_bytes),
               "second_choice_pct": 100.0 * i.second_choice_kmers / max(1, i.total_kmers),
               "stash_pct": 100.0 * i.stash_kmers / max(1, i.total_kmers), "db_open_s": round(t_open, 3),
               "kernel_ms": round(kms, 5), "step_ms": round(el / args.steps * 1e3, 5),
               "gkmers_per_s": round(nk / (kms * 1e-3) / 1e9, 1), "roofline_frac": round(nk * 11 / (kms * 1e-3) / 8e12, 4)}
        if args.merge_pmc:
            row.update({k.lower(): round(v) for k, v in pmc_requests(os.path.join(args.merge_pmc, f"pmc_L{L}")).items()})
        rows.append(row)
        print(json.dumps(row), flush=True)
        db.close()
        engine.release_cached()
    if not args.one:
        os.makedirs(os.path.dirname(args.csv) or ".", exist_ok=True)
        with open(args.csv, "w", newline="") as f:
            keys = []
            for r in rows:
                keys += [k for k in r if k not in keys]
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(rows)
        print("wrote", args.csv)


if __name__ == "__main__":
    main()
