#!/usr/bin/env python3
"""Stash share and screening rate of a GENOMIC database (runs of k-mers that share minimizers) against table density, on the GPU
box: one non-tiled synthetic genome of --len bases, its sample's KMC database, the genome's own 50 kb tiling screened against
cell tables of 2^lbits lines. (The random-k-mer databases of the config-5 test say little about a real wheat-size count: there
every k-mer is a run of its own.)

    python tools/genomic_density_probe.py [--len 2.4e9] [--lbits 27,28,29]"""
import argparse
import json
import os
os.environ.setdefault("KCFX_TEST_HOOKS", "1")     # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kcftools_amd import engine  # noqa: E402

K = 31


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--len", type=float, default=2.4e9)
    ap.add_argument("--lbits", default="27,28,29")
    ap.add_argument("--base", default="/dev/shm")
    ap.add_argument("--ext", default="0,1,auto", help="plain / extended tables (KCFX_TABLE_EXT), auto = the library decides")
    ap.add_argument("--regions", default="1", help="regions of 2^lbits lines each (KCFX_TABLE_REGIONS); with --lbits 0 the size policy decides everything")
    args = ap.parse_args()
    import torch
    d = tempfile.mkdtemp(prefix="kcfx_gd_", dir=args.base)
    try:
        prefix = os.path.join(d, "g")
        t0 = time.time()
        out = subprocess.run([os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth"), "--out", prefix, "--ref-len", str(int(args.len)), "--seed", "41",
                              "--sample-seed", "42", "--noise", "0"], check=True, capture_output=True, text=True).stdout
        meta = json.loads(out.strip().splitlines()[-1])
        print(f"genome {args.len / 1e9:.2f} Gb, {meta['db_kmers'] / 1e9:.2f} G k-mers, generated in {time.time() - t0:.0f} s", flush=True)
        text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
        W = 50_000
        starts = np.arange(0, len(text) - K + 1, W - K + 1, dtype=np.int64)
        lens = np.minimum(W, len(text) - starts).astype(np.int32)
        ref = engine.Reference(text, starts, lens, K)
        stream = torch.cuda.current_stream()
        outbuf = torch.zeros(len(starts) * 10, dtype=torch.int32, device="cuda:0")
        base = None
        for lb, ext, reg in [(int(x), y, int(z)) for x in args.lbits.split(",") for y in args.ext.split(",") for z in args.regions.split(",")]:
            if lb:
                os.environ["KCFX_TABLE_LBITS"] = str(lb)
                os.environ["KCFX_TABLE_REGIONS"] = str(reg)
            else:
                os.environ.pop("KCFX_TABLE_LBITS", None)
                os.environ.pop("KCFX_TABLE_REGIONS", None)
            if ext == "auto":
                os.environ.pop("KCFX_TABLE_EXT", None)
            else:
                os.environ["KCFX_TABLE_EXT"] = ext
            t1 = time.time()
            try:
                db = engine.KmcDb.open(prefix, no_fallback=True)
            except engine.KcfxError as exc:
                print(f"2^{lb} ext {ext}: {exc}", flush=True)
                continue
            t_open = time.time() - t1
            i = db.info
            res = ref.screen(db, 1)
            if base is None:
                base = res
            same = all(np.array_equal(res[f], base[f]) for f in engine.INT_FIELDS)
            ref.prof_enable(True)
            ref.prof_read(reset=True)
            for _ in range(3):
                ref.screen_async(db, outbuf.data_ptr(), stream.cuda_stream, 1)
            torch.cuda.synchronize()
            ms, cnt = ref.prof_read(reset=True)
            ref.prof_enable(False)
            n = int(res["total_kmers"].sum())
            print(f"{i.regions} x 2^{int(np.log2(i.table_lines // i.regions))} lines ({ext}: {'extended' if i.extended else 'plain'}): {i.stored_kmers / (16.0 * i.table_lines):.2f} per cell, {i.table_bytes / 2**30:.1f} GiB incl. stash, opened in {t_open:.1f} s, "
                  f"stash {100.0 * i.stash_kmers / i.stored_kmers:.1f} %, second / third lines {100.0 * i.second_choice_kmers / i.stored_kmers:.1f} %, "
                  f"pass {ms / cnt:.2f} ms = {n / (ms / cnt) / 1e6:.0f} G k-mers/s, hit rate {res['obs'].sum() / n:.3f}, identical to the first: {same}", flush=True)
            db.close()
            engine.release_cached()
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
