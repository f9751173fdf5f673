#!/usr/bin/env python3
"""What the first kernel launch of a process costs (code-object load): kcfx_db_open of a tiny database, first and second call."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from kcftools_amd import engine, synth
L = engine.lib()
t1 = time.perf_counter()
n = engine.device_count()
t2 = time.perf_counter()
d = tempfile.mkdtemp()
prm = synth.KmcParams(k=31, p=3, counter_size=2, sig_len=7, n_bins=4)
kmers = np.unique(np.random.default_rng(1).integers(0, 1 << 62, 1000, dtype=np.uint64) << np.uint64(2))
synth.write_kmc(d + "/t", kmers, np.ones(len(kmers), dtype=np.uint32), prm)
t3 = time.perf_counter()
db = engine.KmcDb.open(d + "/t")
t4 = time.perf_counter()
db2 = engine.KmcDb.open(d + "/t")
t5 = time.perf_counter()
print(f"import {t1-t0:.3f} s, device_count {t2-t1:.3f} s, first kcfx_db_open (HIP init + code object + 4 GiB scratch) {t4-t3:.3f} s, second {t5-t4:.3f} s; lib {os.path.getsize(engine.LIB_PATH)/1e6:.1f} MB")
