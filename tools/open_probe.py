import os, sys, time
os.environ.setdefault("KCFX_TEST_HOOKS", "1")     # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import subprocess
from kcftools_amd import engine
import bench
synth_bin = bench.ensure_built()
d = "/tmp/kcfx_open_probe"; os.makedirs(d, exist_ok=True)
LENS = [30_427_671, 19_698_289, 23_459_830, 18_585_056, 26_975_502]
pf = []
for i in range(3):
    p = os.path.join(d, f"s{i}")
    if not os.path.exists(p + ".kmc_suf"):
        subprocess.run([os.path.join(os.path.dirname(engine.LIB_PATH), "bin", "kcfx_synth"), "--out", p, "--contig-lens", ",".join(map(str, LENS)), "--seed", "3", "--sample-seed", str(50 + i), "--noise", "1000000"], check=True, capture_output=True)
    pf.append(p)
for q in (0, 30_000_000, 14_300_000_000):
    for rep in range(3):
        t = time.time(); db = engine.KmcDb.open(pf[rep], expected_queries=q); dt = time.time() - t
        print(f"expected_queries {q:.1e}: open {dt*1e3:.0f} ms (inner {db.info.load_seconds*1e3:.0f}), lines {db.info.table_lines}, stash {db.info.stash_kmers}", flush=True)
        db.close()
os.environ["KCFX_LOAD_TRACE"] = "1"
db = engine.KmcDb.open(pf[0], expected_queries=14_300_000_000); db.close()
