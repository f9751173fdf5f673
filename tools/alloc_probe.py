#!/usr/bin/env python3
"""Does the kernel's time depend on WHICH block of HBM the table got? Several handles of the bench database opened in one process
(each its own allocation), the screening kernel timed on each in turn, twice."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
from kcftools_amd import engine
synth_bin = bench.ensure_built()
prefix, meta = bench.make_dataset(synth_bin, os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"), 30_427_671, 150_000_000, 2, os.cpu_count() or 8)
text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
K = 31
s, e = engine.windows(len(text), 10_000, 0, K)
os.environ["KCFX_VERBOSE"] = "1"                  # (kcfx_db_open / kcfx_ref_create then say where their buffers are: stderr)
ref = engine.Reference(text, s.astype(np.int64), (e - s).astype(np.int32), K)
out = torch.zeros(len(s) * 10, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream()
dbs = [engine.KmcDb.open(prefix) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5)]
def t(db, n=300):
    for _ in range(20):
        ref.screen_async(db, out.data_ptr(), st.cuda_stream, 1)
    torch.cuda.synchronize()
    ref.prof_enable(True); ref.prof_read(reset=True)
    for _ in range(n):
        ref.screen_async(db, out.data_ptr(), st.cuda_stream, 1)
    torch.cuda.synchronize()
    ms, c = ref.prof_read(reset=True); ref.prof_enable(False)
    return ms / c
for rep in range(2):
    print("round", rep, " ".join(f"{t(db):.5f}" for db in dbs), flush=True)
# the same handles against a second reference (other addresses for text, plan and bitmaps): is a handle's speed its own?
# (the cache-state experiments -- read-only sweep, dirty sweep, first launch on a fresh handle -- live in tools/cache_state_probe.py and
# tools/first_pass_probe.py since round 5: the 1 GiB fill_ that stood here left the caches dirty and measured its own write-back)
keep = [ref]
for j, pad_mb in enumerate((300, 5000, 40000)):
    pad = torch.zeros(pad_mb * 1_000_000, dtype=torch.uint8, device="cuda:0")      # (moves the next allocations somewhere else)
    keep.append(pad)
    ref = engine.Reference(text, s.astype(np.int64), (e - s).astype(np.int32), K)
    keep.append(ref)
    for rep in range(2):
        print(f"ref {j + 2}, round", rep, " ".join(f"{t(db):.5f}" for db in dbs), flush=True)
