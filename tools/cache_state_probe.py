#!/usr/bin/env python3
"""The one-launch screening kernel in three cache states, in an order rocprofv3's per-dispatch counter rows can be split by:
N launches back to back (hot), N with a read-only 1 GiB sweep before each, N with a 1 GiB fill before each. Prints the event
timings; under `rocprofv3 --kernel-trace --pmc ...` (tools/cache_state_pmc.sh) the screening kernel's dispatches are, in order,
5 warm-up + N hot + N read-swept + N write-swept."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from kcftools_amd import engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
synth_bin = bench.ensure_built()
prefix, meta = bench.make_dataset(synth_bin, os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"), 30_427_671, 150_000_000, 2, os.cpu_count() or 8)
text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
K = 31
s, e = engine.windows(len(text), 10_000, 0, K)
ref = engine.Reference(text, s.astype(np.int64), (e - s).astype(np.int32), K, expected_screens=1)
db = engine.KmcDb.open(prefix, expected_queries=int(ref.n_kmer_slots))
out = torch.zeros(len(s) * 10, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
sweep = torch.ones(1 << 28, dtype=torch.int32, device="cuda:0")
sink = []


def run(pre, n):
    ref.prof_enable(True)
    ref.prof_read(reset=True)
    for _ in range(n):
        if pre:
            pre()
        ref.screen_async(db, out.data_ptr(), st, 1)
    torch.cuda.synchronize()
    ms, c = ref.prof_read(reset=True)
    ref.prof_enable(False)
    return ms / max(1, c)


# (clocks: 60 ms of the launch itself first, untimed and NOT profiled apart from the 5 named warm-up launches above)
for _ in range(5):
    ref.screen_async(db, out.data_ptr(), st, 1)
torch.cuda.synchronize()
print(f"hot {run(None, N):.5f} ms, read-swept {run(lambda: sink.append(sweep.sum()), N):.5f} ms, write-swept {run(lambda: sweep.fill_(1), N):.5f} ms, "
      f"hot again {run(None, N):.5f} ms", flush=True)
