#!/usr/bin/env python3
"""How well do database loads overlap with each other on one GPU? (run on the GPU box) Opens the bench sample N times with T
loader threads (kcfx_db_open_opts releases the GIL under ctypes) and prints the wall time per thread count, then one open under
KCFX_LOAD_TRACE=1 for the phase split.

    python tools/load_overlap_probe.py [--n 8]"""
import argparse
import os
os.environ.setdefault("KCFX_TEST_HOOKS", "1")     # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    args = ap.parse_args()
    subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--cpu-baseline-seconds", "0", "--no-end-to-end"],
                   capture_output=True)
    from kcftools_amd import engine
    P = os.path.join(os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"), "r30427671_d150000000_s2")
    q = 30_000_000
    engine.KmcDb.open(P, expected_queries=q).close()              # page cache, pools
    for T in (1, 2, 3, 4, 8):
        for rep in range(2):
            todo = list(range(args.n))
            lock = threading.Lock()
            held = []

            def work():
                while True:
                    with lock:
                        if not todo:
                            return
                        todo.pop()
                    db = engine.KmcDb.open(P, expected_queries=q)
                    with lock:
                        held.append(db)
                        if len(held) > 2:                           # at most a few resident, as the sharded loader would keep
                            held.pop(0).close()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work) for _ in range(T)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            wall = time.perf_counter() - t0
            for d in held:
                d.close()
            print(f"{args.n} opens, {T} loader thread(s): {wall:.3f} s = {wall / args.n * 1e3:.1f} ms per database", flush=True)
    os.environ["KCFX_LOAD_TRACE"] = "1"
    engine.KmcDb.open(P, expected_queries=q).close()


if __name__ == "__main__":
    main()
