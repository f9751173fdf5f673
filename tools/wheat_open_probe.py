import os, sys, time, subprocess
os.environ.setdefault("KCFX_TEST_HOOKS", "1")     # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000_000
synth = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
prefix = "/dev/shm/kcfx_wheat_open"
t0 = time.time()
subprocess.run([synth, "--out", prefix, "--ref-len", "2000000", "--seed", "1", "--sample-seed", "2", "--noise", str(N)], check=True, capture_output=True)
print(f"generated in {time.time() - t0:.0f} s, suf {os.path.getsize(prefix + '.kmc_suf') / 1e9:.1f} GB", flush=True)
t0 = time.time()
with open(prefix + ".kmc_suf", "rb") as f:
    while f.read(1 << 26): pass
print(f"read once in {time.time() - t0:.1f} s", flush=True)
from kcftools_amd import engine
os.environ["KCFX_LOAD_TRACE"] = "1"
for env in ({}, {}, {}):
    os.environ.update(env)
    print("==", env, flush=True)
    t = time.time(); db = engine.KmcDb.open(prefix); dt = time.time() - t
    i = db.info
    print(f"open {dt:.2f} s; lines {i.table_lines}, ext {i.extended}, stash {i.stash_kmers}, second {i.second_choice_kmers}", flush=True)
    db.close()
    for k in env: os.environ.pop(k)
os.unlink(prefix + ".kmc_suf"); os.unlink(prefix + ".kmc_pre")
