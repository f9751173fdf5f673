#!/usr/bin/env python3
"""What does the FIRST screening launch on a fresh handle pay for? bench.py's cache_state leg says: ~0.19 ms against ~0.09 for the
launch after it, whatever the caches hold (a 1 GiB read-only sweep before it changes nothing, 50 ms of idle time neither). This
script makes the pair visible to rocprofv3: per cycle a fresh kcfx_db_open, then TWO launches of the one-pass form back to back, then
close -- under `rocprofv3 --kernel-trace --pmc ...` the per-dispatch counter rows of the screening kernel alternate first / second.

    python tools/first_pass_probe.py [cycles]          (prints its own event timings; tools/first_pass_pmc.sh wraps it)

Variants (FIRST_PASS_MODE): plain | same -- the SAME handle screened again after 20 other launches against another handle | other -- the first launch after the open goes to an
OLD handle, then the fresh one (is it the handle or the moment?) | idle -- no open, only time without GPU work | allocfree / bigfree --
no open, device memory taken and given back to the driver (small buffers / a 60 MB block through kcfx_release_cached), then the OLD handle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from kcftools_amd import engine  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = os.environ.get("FIRST_PASS_MODE", "plain")
synth_bin = bench.ensure_built()
prefix, meta = bench.make_dataset(synth_bin, os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"), 30_427_671, 150_000_000, 2, os.cpu_count() or 8)
text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
K = 31
s, e = engine.windows(len(text), 10_000, 0, K)
ref = engine.Reference(text, s.astype(np.int64), (e - s).astype(np.int32), K, expected_screens=1)
out = torch.zeros(len(s) * 10, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
expected = int(ref.n_kmer_slots)


def launch(db, n=1):
    res = []
    for _ in range(n):
        ref.prof_enable(True)
        ref.prof_read(reset=True)
        ref.screen_async(db, out.data_ptr(), st, 1)
        torch.cuda.synchronize()
        ms, c = ref.prof_read(reset=True)
        ref.prof_enable(False)
        res.append(ms / max(1, c))
    return res


warm = engine.KmcDb.open(prefix, expected_queries=expected)
launch(warm, 5)
for c in range(cycles):
    if mode == "idle":
        # nothing but time: the old handle after 1 / 5 / 50 ms without any GPU work
        import time
        line = []
        for ms in (0.2, 1, 5, 50):
            launch(warm, 3)
            time.sleep(ms * 1e-3)
            a, b = launch(warm, 2)
            line.append(f"{ms} ms idle: {a:.5f} / {b:.5f}")
        print(f"cycle {c}: old handle, first / second launch after " + "; ".join(line), flush=True)
        continue
    if mode == "events":
        # which kind of event in front of a launch on the OLD handle makes it slow? (no open, no library allocation)
        line = []
        host = torch.empty(64 << 20, dtype=torch.uint8)
        pinned = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        devbuf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
        big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
        for name, ev in (("nothing", lambda: None), ("H2D 64 MB pageable", lambda: devbuf.copy_(host)), ("H2D 64 MB pinned", lambda: devbuf.copy_(pinned)),
                         ("D2H 64 MB", lambda: host.copy_(devbuf)), ("D2D 1 GiB", lambda: big.copy_(big.roll(1) if False else big)),
                         ("torch malloc+free 2 GiB (empty_cache)", lambda: (torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0"), torch.cuda.empty_cache()))):
            launch(warm, 3)
            ev()
            torch.cuda.synchronize()
            a, b = launch(warm, 2)
            line.append(f"{name}: {a:.5f} / {b:.5f}")
        print(f"cycle {c}: old handle, first / second launch after " + "; ".join(line), flush=True)
        del host, pinned, devbuf, big
        continue
    if mode in ("allocfree", "bigfree"):
        # no open at all: device allocations made and given back to the driver (a small reference handle: a few hipMalloc / hipFree
        # of KB-sized buffers; "bigfree": a 1 GiB block through the library's pool and kcfx_release_cached), then the OLD handle
        if mode == "allocfree":
            r2 = engine.Reference(text[:100_000], [0], [100_000], K)
            r2.close()
        else:
            r3 = engine.Reference(text + text[:1_000_000], s[:10].astype(np.int64), (e - s)[:10].astype(np.int32), K)
            r3.close()
            engine.release_cached()
        a, b = launch(warm, 2)
        print(f"cycle {c}: old handle after {mode}: first {a:.5f} ms, second {b:.5f} ms", flush=True)
        continue
    if mode == "clocks":
        # round 6: the memory / fabric / shader clock levels the driver reports (sysfs pp_dpm_*), sampled as fast as a Python thread
        # reads them, across open -> first launch -> second launch -> 20 more
        import glob, threading, time
        files = {}
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            for nm in ("pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_sclk", "pp_dpm_socclk"):
                f = os.path.join(card, nm)
                if os.path.exists(f) and nm not in files:
                    files[nm] = f
        samples, marks, stop = [], [], [False]
        def cur(path):
            try:
                for ln in open(path).read().splitlines():
                    if ln.rstrip().endswith("*"):
                        return ln.split(":")[1].strip().rstrip("*").strip()
            except Exception:
                return "?"
            return "?"
        def sampler():
            while not stop[0]:
                samples.append((time.perf_counter(), tuple(cur(f) for f in files.values())))
        th = threading.Thread(target=sampler)
        th.start()
        time.sleep(0.02)
        marks.append((time.perf_counter(), "open begins"))
        db = engine.KmcDb.open(prefix, expected_queries=expected)
        marks.append((time.perf_counter(), "open done"))
        a = launch(db, 1)[0]
        marks.append((time.perf_counter(), f"first launch done ({a:.5f} ms)"))
        b = launch(db, 1)[0]
        marks.append((time.perf_counter(), f"second launch done ({b:.5f} ms)"))
        rest = launch(db, 20)
        marks.append((time.perf_counter(), f"20 more done (median {sorted(rest)[10]:.5f} ms)"))
        time.sleep(0.02)
        stop[0] = True
        th.join()
        db.close()
        t0 = marks[0][0]
        print(f"cycle {c}: clock files {list(files)}; {len(samples)} samples", flush=True)
        last = None
        mi = 0
        for t, v in samples:
            while mi < len(marks) and marks[mi][0] <= t:
                print(f"    {1e3 * (marks[mi][0] - t0):9.3f} ms  -- {marks[mi][1]}")
                mi += 1
            if v != last:
                print(f"    {1e3 * (t - t0):9.3f} ms  " + "  ".join(f"{n}={x}" for n, x in zip(files, v)))
                last = v
        for m in marks[mi:]:
            print(f"    {1e3 * (m[0] - t0):9.3f} ms  -- {m[1]}")
        continue
    db = engine.KmcDb.open(prefix, expected_queries=expected)
    if mode == "warmer":
        # round 6: a throw-away launch over ~1 % of the windows first (a reference of its own), then the full launch twice
        if c == 0:
            n1 = max(1, len(s) // 100)
            small = engine.Reference(text, s[:n1].astype(np.int64), (e - s)[:n1].astype(np.int32), K, expected_screens=1)
            out_s = torch.zeros(n1 * 10, dtype=torch.int32, device="cuda:0")
        small.prof_enable(True)
        small.prof_read(reset=True)
        small.screen_async(db, out_s.data_ptr(), st, 1)
        torch.cuda.synchronize()
        wms, wc = small.prof_read(reset=True)
        small.prof_enable(False)
        a, b = launch(db, 2)
        print(f"cycle {c}: warmer over {n1} windows {wms / max(1, wc):.5f} ms, then first {a:.5f} ms, second {b:.5f} ms", flush=True)
        db.close()
        continue
    if mode == "other":
        # is it the fresh HANDLE or the moment? the first launch after the open goes to the OLD handle (a table nobody has written
        # for a long time), the fresh one is screened after it
        o1 = launch(warm, 1)[0]
        a, b = launch(db, 2)
        print(f"cycle {c}: old handle right after the open {o1:.5f} ms, then the fresh handle: first {a:.5f} ms, second {b:.5f} ms", flush=True)
        db.close()
        continue
    a, b = launch(db, 2)
    line = f"cycle {c}: first {a:.5f} ms, second {b:.5f} ms"
    if mode == "same":
        launch(warm, 20)
        c2, d2 = launch(db, 2)
        line += f"; after 20 launches against another handle: {c2:.5f}, {d2:.5f}"
    print(line, flush=True)
    db.close()
