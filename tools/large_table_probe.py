#!/usr/bin/env python3
"""A database of billions of GENOMIC k-mers in the table the size policy gives it, and a few passes of its genome's tiling -- the
workload of tools/pmc_large_table.sh (counters of the regime a real database lives in: 0.3 k-mers per cell, extended table).

    python tools/large_table_probe.py --prefix /dev/shm/kcfx_large/g [--passes 5] [--stats]     (--stats: KCFX_LIB = the KCFX_EXP_STATS build)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("KCFX_TEST_HOOKS", "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefix", required=True)
    ap.add_argument("--window", type=int, default=50_000)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--stats", action="store_true")
    args = ap.parse_args()
    import torch
    from kcftools_amd import engine
    K = 31
    text = np.fromfile(args.prefix + ".ref.txt", dtype=np.uint8).tobytes()
    W = args.window
    starts = np.arange(0, len(text) - K + 1, W - K + 1, dtype=np.int64)
    lens = np.minimum(W, len(text) - starts).astype(np.int32)
    starts, lens = starts[lens >= K], lens[lens >= K]
    ref = engine.Reference(text, starts, lens, K)
    t0 = time.time()
    db = engine.KmcDb.open(args.prefix)
    i = db.info
    print(json.dumps({"open_s": round(time.time() - t0, 2), "lines": int(i.table_lines), "regions": int(i.regions), "extended": int(i.extended),
                      "per_cell": round(i.stored_kmers / (16.0 * i.table_lines), 3), "table_GiB": round(i.table_bytes / 2**30, 1),
                      "outside_first_choice_pct": round(100.0 * i.second_choice_kmers / max(1, i.stored_kmers), 2),
                      "stash_pct": round(100.0 * i.stash_kmers / max(1, i.stored_kmers), 2)}), flush=True)
    out = torch.zeros(len(starts) * 10, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    L = engine.lib()
    if args.stats:
        L.kcfx_exp_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
        L.kcfx_exp_stats(None, 1)
    ref.screen_async(db, out.data_ptr(), st, 1)
    torch.cuda.synchronize()
    if args.stats:
        buf = (C.c_uint64 * 16)()
        L.kcfx_exp_stats(buf, 1)
        names = ["queued", "found_second", "sent_on_from_second", "found_third", "stash_lookups", "found_stash", "second_line_requests",
                 "words_with_queued", "found_first_choice", "deferred_rounds"]
        print(json.dumps({n: int(buf[j]) for j, n in enumerate(names)}), flush=True)
    ref.prof_enable(True)
    ref.prof_read(reset=True)
    for _ in range(args.passes):
        ref.screen_async(db, out.data_ptr(), st, 1)
    torch.cuda.synchronize()
    ms, n = ref.prof_read(reset=True)
    res = out.cpu().numpy().view(engine.RESULT_DTYPE)[:len(starts)]
    nk = int(res["total_kmers"].sum())
    print(json.dumps({"kmers_per_pass": nk, "obs": int(res["obs"].sum()), "pass_ms": round(ms / max(1, n), 3),
                      "G_kmers_per_s": round(nk / (ms / max(1, n) * 1e-3) / 1e9, 1), "frac": round(nk * 11 / (ms / max(1, n) * 1e-3) / 8e12, 3)}), flush=True)


if __name__ == "__main__":
    main()
