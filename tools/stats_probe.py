import os, sys, ctypes, numpy as np
sys.path.insert(0, ".")
os.environ["KCFX_LIB"] = os.path.abspath("build_exp/libkcfx_stats.so")
import bench, torch
from kcftools_amd import engine
synth_bin = bench.ensure_built()
prefix, meta = bench.make_dataset(synth_bin, "/tmp/kcfx_bench", 30_427_671, 150_000_000, 2, os.cpu_count() or 8)
print({k: meta[k] for k in meta if "kmers" in k})
text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
K = 31
L = engine.lib()
for lb in (27, 26):
    os.environ["KCFX_TABLE_LBITS"] = str(lb)
    db = engine.KmcDb.open(prefix)
    i = db.info
    print("lbits", lb, "second %", 100.0*i.second_choice_kmers/i.total_kmers, "stash %", 100.0*i.stash_kmers/i.total_kmers, "stash kmers", i.stash_kmers)
    s, e = engine.windows(len(text), 10_000, 0, K)
    ref = engine.Reference(text, s.astype(np.int64), (e - s).astype(np.int32), K)
    out = (ctypes.c_ulonglong * 8)()
    L.kcfx_debug_stats(out, 1)
    res = ref.screen(db, 1)
    L.kcfx_debug_stats(out, 1)
    v = list(out)
    print("queued", v[0], "found in the second line", v[1], "second-line cell empty", v[5], "stash probes", v[2], "found in the stash", v[3], "distinct second lines (adjacent lanes)", v[6], "distinct first lines", v[7], "hits", int(res["obs"].sum()))
    db.close()
