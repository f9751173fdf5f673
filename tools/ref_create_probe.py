#!/usr/bin/env python3
"""What does kcfx_ref_create cost, and which part of it? (run on the GPU box) The bench reference (30 427 671 bp, 3 052 windows),
created several times in one process, planned and text-only (the first call pays the process's first launches: ~50 ms).

    python tools/ref_create_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from kcftools_amd import engine, synth
    L, k = 30_427_671, 31
    t0 = time.perf_counter()
    text = synth.random_reference(L, 1)
    starts, ends = synth.tiling_windows(L, 10_000, k)
    lens = (ends - starts).astype(np.int32)
    print(f"reference generated in {time.perf_counter() - t0:.2f} s, {len(starts)} windows", flush=True)
    torch.zeros(1, device="cuda:0")                                   # HIP initialised, as after kcfx_db_open
    for what, kw in (("planned", {}), ("planned", {}), ("planned", {}), ("text only", {"expected_screens": 1}),
                     ("text only", {"expected_screens": 1})):
        t0 = time.perf_counter()
        r = engine.Reference(text, starts, lens, k, **kw)
        dt = time.perf_counter() - t0
        r.close()
        print(f"kcfx_ref_create ({what:9s}): {dt * 1e3:7.1f} ms", flush=True)
    os.environ["KCFX_REF_TRACE"] = "1"
    r = engine.Reference(text, starts, lens, k)
    r.close()
    r = engine.Reference(text, starts, lens, k, expected_screens=1)
    r.close()


if __name__ == "__main__":
    main()
