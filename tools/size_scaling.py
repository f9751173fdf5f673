#!/usr/bin/env python3
"""Kernel time against the number of reference k-mers on the bench database (same table, prefixes of the chr1-sized text):
the intercept of the line is what a launch pays that is not per k-mer (ramp-up, the last partial round of workgroups).

    python tools/size_scaling.py [--steps 200]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--fractions", default="0.0625,0.125,0.25,0.5,0.75,1.0")
    ap.add_argument("--workdir", default=os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"))
    args = ap.parse_args()
    import bench
    import torch
    from kcftools_amd import engine
    synth_bin = bench.ensure_built()
    prefix, meta = bench.make_dataset(synth_bin, args.workdir, 30_427_671, 150_000_000, 2, os.cpu_count() or 8)
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    K = 31
    db = engine.KmcDb.open(prefix)
    st = torch.cuda.current_stream()
    rows = []
    for rep in range(2):
        for f in [float(x) for x in args.fractions.split(",")]:
            n = int(len(text) * f)
            starts, ends = engine.windows(n, 10_000, 0, K)
            wb, wl = starts.astype(np.int64), (ends - starts).astype(np.int32)
            ref = engine.Reference(text[:n], wb, wl, K)
            out = torch.zeros(len(wb) * 10, dtype=torch.int32, device="cuda:0")
            for _ in range(20):
                ref.screen_async(db, out.data_ptr(), st.cuda_stream, 1)
            torch.cuda.synchronize()
            ref.prof_enable(True)
            ref.prof_read(reset=True)
            for _ in range(args.steps):
                ref.screen_async(db, out.data_ptr(), st.cuda_stream, 1)
            torch.cuda.synchronize()
            ms, cnt = ref.prof_read(reset=True)
            ref.prof_enable(False)
            res = out.cpu().numpy().view(engine.RESULT_DTYPE)[:len(wb)]
            nk = int(res["total_kmers"].sum())
            row = {"fraction": f, "kmers": nk, "kernel_ms": round(ms / cnt, 5), "ns_per_Mkmer": round(ms / cnt * 1e6 / (nk / 1e6), 2)}
            rows.append(row)
            print(json.dumps(row), flush=True)
            ref.close() if hasattr(ref, "close") else None
    x = np.array([r["kmers"] for r in rows], dtype=np.float64)
    y = np.array([r["kernel_ms"] for r in rows])
    b, a = np.polyfit(x, y, 1)
    print(json.dumps({"intercept_us": round(a * 1e3, 2), "slope_ms_per_30M": round(b * 30.0e6, 5)}))


if __name__ == "__main__":
    main()
