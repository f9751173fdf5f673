#!/usr/bin/env python3
"""Do two kcfx_db_open_opts calls on two threads overlap? (round 6: each load has a stream pair of its own and the loads take turns on
the file; tools/e2e_config3.py shows no gain from two loaders -- which phase stretches?)

    python tools/load_overlap_probe.py            (needs the eight databases of tools/e2e_config3.py in /tmp/kcfx_e2e3)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kcftools_amd import engine  # noqa: E402

D = "/tmp/kcfx_e2e3"
Q = 119_000_000


def open_close(i, out):
    t = time.perf_counter()
    db = engine.KmcDb.open(os.path.join(D, f"s{i}"), expected_queries=Q)
    out[i] = time.perf_counter() - t
    return db


dbs = [open_close(0, {}), open_close(1, {})]           # warm: runtime, pool, page cache
for d in dbs:
    d.close()
for rnd in range(3):
    times = {}
    t0 = time.perf_counter()
    a = open_close(2, times)
    b = open_close(3, times)
    seq = time.perf_counter() - t0
    a.close(); b.close()
    times2 = {}
    got = {}
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda i=i: got.__setitem__(i, open_close(i, times2))) for i in (4, 5)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    par = time.perf_counter() - t0
    for d in got.values():
        d.close()
    print(f"round {rnd}: two opens one after the other {seq * 1e3:.0f} ms ({times[2] * 1e3:.0f} + {times[3] * 1e3:.0f}); on two threads {par * 1e3:.0f} ms "
          f"({times2[4] * 1e3:.0f} and {times2[5] * 1e3:.0f})", flush=True)
