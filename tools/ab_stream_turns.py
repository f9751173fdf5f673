#!/usr/bin/env python3
"""Same-box A/B of KCFX_STREAM_TURNS on the sharded call with two workers on one GPU (what config-5 runs): 16 samples of the 119 Mb
genome against a 1.2 Gb reference, devices (0, 0)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kcftools_amd import engine
LENS = [30_427_671, 19_698_289, 23_459_830, 18_585_056, 26_975_502]
d = "/tmp/kcfx_ab_turns"; os.makedirs(d, exist_ok=True)
synth = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
pf = []
for i in range(16):
    p = os.path.join(d, f"s{i}")
    if not os.path.exists(p + ".kmc_suf"):
        subprocess.run([synth, "--out", p, "--contig-lens", ",".join(map(str, LENS)), "--seed", "3", "--sample-seed", str(70 + i), "--noise", "1000000"] + (["--fasta"] if i == 0 else []), check=True, capture_output=True)
    pf.append(p)
text = np.fromfile(pf[0] + ".ref.txt", dtype=np.uint8)
text = np.tile(text, 10).tobytes()
K = 31
W = 50_000
starts = np.arange(0, len(text) - K + 1, W - K + 1, dtype=np.int64)
lens = np.minimum(W, len(text) - starts).astype(np.int32)
for rep in range(3):
    for turns, loaders in (("1", "1"), ("0", "1"), ("1", "2"), ("0", "2")):
        os.environ["KCFX_STREAM_TURNS"] = turns
        os.environ["KCFX_SHARD_LOADERS"] = loaders
        t = time.time()
        got, st = engine.screen_sharded(pf, K, text, starts, lens, 1, devices=(0, 0))
        print(f"turns {turns} loaders/worker {loaders}: {time.time() - t:.2f} s wall, loads {st.seconds_db_open_sum:.2f} s, exposed {st.seconds_exposed_load:.2f} s, screening {st.seconds_screen_sum:.2f} s", flush=True)
