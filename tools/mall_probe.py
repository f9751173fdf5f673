#!/usr/bin/env python3
"""Does repeating the SAME launch help (lines of the table left in the 256 MB Infinity Cache by the previous step)? The kernel timed
on one handle repeatedly and on two / three handles of the same database in rotation (their tables are different blocks of HBM, so a
step finds nothing of its own from the step before)."""
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
ref = engine.Reference(text, s.astype(np.int64), (e - s).astype(np.int32), K)
out = torch.zeros(len(s) * 10, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream()
dbs = [engine.KmcDb.open(prefix) for _ in range(3)]
def t(hs, n=300):
    for i in range(30):
        ref.screen_async(hs[i % len(hs)], out.data_ptr(), st.cuda_stream, 1)
    torch.cuda.synchronize()
    ref.prof_enable(True); ref.prof_read(reset=True)
    for i in range(n):
        ref.screen_async(hs[i % len(hs)], out.data_ptr(), st.cuda_stream, 1)
    torch.cuda.synchronize()
    ms, c = ref.prof_read(reset=True); ref.prof_enable(False)
    return ms / c
for rep in range(2):
    print("one handle:", " ".join(f"{t([d]):.5f}" for d in dbs), "| two in rotation:", f"{t(dbs[:2]):.5f} {t(dbs[1:]):.5f}", "| three:", f"{t(dbs):.5f}", flush=True)
