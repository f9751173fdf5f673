#!/usr/bin/env python3
"""What the queued look-ups of a screening pass are made of (experiment build: make fast FASTFLAGS=-DKCFX_EXP_STATS ->
build_exp/libkcfx_stats.so; run with KCFX_LIB pointing at it). Bench workload unless --prefix names another dataset.

    KCFX_LIB=$PWD/build_exp/libkcfx_stats.so python3 tools/exp_stats_probe.py --lbits 27,25 [--few]
"""
import argparse
import ctypes as C
import json
import os
os.environ.setdefault("KCFX_TEST_HOOKS", "1")     # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["queued", "found_second", "sent_on_from_second", "found_third", "stash_lookups", "found_stash", "second_line_requests",
         "words_with_queued", "found_first_choice", "deferred_rounds"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lbits", default="27,25")
    ap.add_argument("--prefix", default="")
    ap.add_argument("--few", action="store_true", help="open as the one-pass job does (expected_queries = one pass: per-line crowding counters)")
    ap.add_argument("--text-only", action="store_true")
    args = ap.parse_args()
    import bench
    from kcftools_amd import engine
    if args.prefix:
        prefix = args.prefix
    else:
        prefix, _ = bench.make_dataset(bench.ensure_built(), os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"), 30_427_671, 150_000_000, 2, os.cpu_count() or 8)
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    K = 31
    s, e = engine.windows(len(text), 10_000, 0, K)
    wb, wl = s.astype(np.int64), (e - s).astype(np.int32)
    ref = engine.Reference(text, wb, wl, K, expected_screens=1 if args.text_only else 0)
    L = engine.lib()
    L.kcfx_exp_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    L.kcfx_exp_stats.restype = C.c_int32
    buf = (C.c_uint64 * 16)()
    for lb in [int(x) for x in args.lbits.split(",")]:
        if lb > 0:
            os.environ["KCFX_TABLE_LBITS"] = str(lb)
        else:
            os.environ.pop("KCFX_TABLE_LBITS", None)
        db = engine.KmcDb.open(prefix, expected_queries=int(ref.n_kmer_slots) if args.few else 0)
        L.kcfx_exp_stats(None, 1)
        res = ref.screen(db, 1)
        L.kcfx_exp_stats(buf, 1)
        i = db.info
        row = {"lines": int(i.table_lines), "per_cell": round(i.total_kmers / (16.0 * i.table_lines), 3),
               "outside_first_choice_pct": round(100.0 * i.second_choice_kmers / max(1, i.total_kmers), 2),
               "stash_pct": round(100.0 * i.stash_kmers / max(1, i.total_kmers), 3), "kmers": int(res["total_kmers"].sum()), "obs": int(res["obs"].sum())}
        row.update({n: int(buf[j]) for j, n in enumerate(NAMES)})
        print(json.dumps(row), flush=True)
        db.close()
        engine.release_cached()


if __name__ == "__main__":
    main()
