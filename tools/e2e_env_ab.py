#!/usr/bin/env python3
"""End-to-end A/B of environment switches under the command-line driver, ON THE GPU BOX:
    python tools/e2e_env_ab.py ROUNDS "" "VAR=1" ...       (each argument: space-separated VAR=value pairs, "" = none)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, envs = int(sys.argv[1]), sys.argv[2:]
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--cpu-baseline-seconds", "0"],
               capture_output=True)                      # data + FASTA + .fai
P = os.path.join(os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"), "r30427671_d150000000_s2")
cli = os.path.join(ROOT, "kcftools_amd", "bin", "kcftools-amd")
for i in range(n):
    for e in envs:
        env = dict(os.environ)
        for kv in e.split():
            k, v = kv.split("=", 1)
            env[k] = v
        t0 = time.perf_counter()
        pr = subprocess.run([cli, "getVariations", "-r", P + ".fa", "-k", P, "-o", P + ".ab.kcf", "-s", "bench", "-f", "window",
                             "-w", "10000", "-c", "1", "--device", "0"], capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        line = [x for x in pr.stdout.splitlines() if "Screened" in x]
        print(repr(e), round(wall, 3), "s;", line[-1].split("Screened")[-1] if line else (pr.stdout[-200:] + pr.stderr[-300:]), flush=True)
