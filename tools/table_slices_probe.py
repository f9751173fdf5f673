#!/usr/bin/env python3
"""A database in table slices on ONE GPU, slice after slice: what a pass costs per slice against the whole table's pass -- the
rehearsal of KCFX_SHARD_TABLE that a one-GPU box allows (on G devices the slices' passes run side by side; here one after the other,
each slice alone on the device, as it would be on its own).

    python tools/table_slices_probe.py [--ref-len 2.3e9] [--slices 4] [--window 50000]

Generates one genome and the k-mers of its sample as ONE KMC database (kcfx_synth, genomic k-mers: runs that share minimizers),
opens it whole (the size policy's table) and as n slices, screens the genome's tiling against each, merges the slices' bitmaps and
compares every window with the whole table's records."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-len", type=float, default=2.3e9)
    ap.add_argument("--slices", type=int, default=4)
    ap.add_argument("--window", type=int, default=50_000)
    ap.add_argument("--workdir", default=None)
    args = ap.parse_args()
    os.environ["KCFX_TEST_HOOKS"] = "1"                     # (the A/B switch below is a test hook)
    import torch
    from kcftools_amd import engine
    K = 31
    L = int(args.ref_len)
    base = args.workdir or tempfile.mkdtemp(prefix="kcfx_slices_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    prefix = os.path.join(base, "g")
    synth = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
    t0 = time.time()
    out = subprocess.run([synth, "--out", prefix, "--ref-len", str(L), "--seed", "41", "--sample-seed", "42", "--noise", "0"],
                         check=True, capture_output=True, text=True).stdout
    meta = json.loads(out.strip().splitlines()[-1])
    print(f"generated {meta['db_kmers'] / 1e9:.2f} G k-mers of a {L / 1e9:.2f} Gb genome in {time.time() - t0:.0f} s", flush=True)
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    os.remove(prefix + ".ref.txt")
    W = args.window
    starts = np.arange(0, len(text) - K + 1, W - K + 1, dtype=np.int64)
    lens = np.minimum(W, len(text) - starts).astype(np.int32)
    starts, lens = starts[lens >= K], lens[lens >= K]
    ref = engine.Reference(text, starts, lens, K)                       # planned: several passes follow
    stream = torch.cuda.current_stream().cuda_stream

    def pass_ms(db, n=3):
        ref.screen_partial_async(db, stream, 1)
        torch.cuda.synchronize()
        ref.prof_enable(True)
        ref.prof_read(reset=True)
        for _ in range(n):
            ref.screen_partial_async(db, stream, 1)
        torch.cuda.synchronize()
        ms, c = ref.prof_read(reset=True)
        ref.prof_enable(False)
        return ms / max(1, c)

    def describe(i):
        return (f"{i.regions} x 2^{int(np.log2(i.table_lines // i.regions))} lines, {i.stored_kmers / (16.0 * i.table_lines):.3f} per cell, "
                f"{i.table_bytes / 2 ** 30:.0f} GiB, {100.0 * i.stash_kmers / max(1, i.stored_kmers):.1f} % in the stash, "
                f"{100.0 * i.second_choice_kmers / max(1, i.stored_kmers):.1f} % in second / third lines{', extended' if i.extended else ''}")

    t0 = time.time()
    whole = engine.KmcDb.open(prefix, no_fallback=True)
    t_open = time.time() - t0
    ms_whole = pass_ms(whole)
    want = ref.finish()
    nk = int(want["total_kmers"].sum())
    print(f"whole table: {describe(whole.info)}; opened in {t_open:.1f} s; pass of {nk / 1e9:.2f} G k-mers {ms_whole:.2f} ms = {nk / ms_whole / 1e6:.0f} G k-mers/s", flush=True)
    whole.close()
    engine.release_cached()
    n = args.slices
    nb, ns = ref.partial_bytes()
    kept = []                                               # the partial results of slices 0 .. n - 2 (520 B per tile each)
    times, times_all = [], []
    for s in range(n):
        t0 = time.time()
        db = engine.KmcDb.open(prefix, no_fallback=True, n_slices=n, slice=s)
        t_open = time.time() - t0
        # the A/B: every position walked with the other slices' lanes idle (round 5's form, what a text-only reference still does)
        os.environ["KCFX_SLICE_WALK_ALL"] = "1"
        ms_all = pass_ms(db)
        del os.environ["KCFX_SLICE_WALK_ALL"]
        ms = pass_ms(db)                                    # (last: its results are the ones exported / merged below)
        times.append(ms)
        times_all.append(ms_all)
        print(f"slice {s} of {n}: {describe(db.info)}; opened in {t_open:.1f} s; pass {ms:.2f} ms (walking every position: {ms_all:.2f} ms)", flush=True)
        if s < n - 1:
            b = torch.zeros(nb // 8, dtype=torch.int64, device="cuda:0")
            c = torch.zeros(ns // 8, dtype=torch.int64, device="cuda:0")
            ref.export_partial_async(b.data_ptr(), c.data_ptr(), stream)
            torch.cuda.synchronize()
            kept.append((b, c))
        db.close()
        engine.release_cached()
    # the reference holds the last slice's results: the others' into them (hits OR-ed, count sums added), one reduction
    t0 = time.time()
    for b, c in kept:
        ref.merge_partial_async(b.data_ptr(), c.data_ptr(), stream)
    torch.cuda.synchronize()
    t_merge = time.time() - t0
    got = ref.finish()
    same_bits = all(np.array_equal(got[f], want[f]) for f in engine.INT_FIELDS)
    print(f"the {n} slices merged ({(nb + ns) / 1e6:.0f} MB of bitmaps and sums per slice, {t_merge * 1e3:.1f} ms for the {n - 1} merges on one device) == the whole "
          f"table's records on every window, every integer field: {same_bits}", flush=True)
    print(f"per-slice pass {min(times):.2f} - {max(times):.2f} ms (mean {sum(times) / n:.2f}) against {ms_whole:.2f} ms for the whole table: "
          f"on {n} devices a pass would take the slowest slice's {max(times):.2f} ms + the merge = {ms_whole / max(times):.2f} x the one-device rate, "
          f"every device walking the positions whose minimizer's class is its own (walking every position: {min(times_all):.2f} - {max(times_all):.2f} ms)", flush=True)
    sys.exit(0 if same_bits else 1)


if __name__ == "__main__":
    main()
