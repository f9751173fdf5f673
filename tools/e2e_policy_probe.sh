#!/bin/bash
export KCFX_TEST_HOOKS=1    # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
# Round 6: what does the table's size cost a ONE-PASS job end to end, in a fresh process, when the device's memory has just been
# given back by another process? (run on the GPU box; needs build_exp/churn = tools/probes/churn.hip)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
D=${KCFX_BENCH_DIR:-/tmp/kcfx_bench}
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from kcftools_amd import synth
sb = bench.ensure_built()
d = os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench")
p, meta = bench.make_dataset(sb, d, 30427671, 150000000, 2, bench.usable_cores())
fa = p + ".fa"
if not os.path.exists(fa):
    synth.write_fasta(fa, [("chr1", np.fromfile(p + ".ref.txt", dtype=np.uint8).tobytes())])
print("dataset", p, meta.get("db_kmers"))
PY
P=$D/r30427671_d150000000_s2
CLI=./kcftools_amd/bin/kcftools-amd
run() {   # run <label> [ENV=..]...
  local label=$1; shift
  local T0=$(date +%s.%N)
  env "$@" KCFX_CLI_TRACE=1 KCFX_LOAD_TRACE=1 $CLI getVariations -r $P.fa -k $P -o /tmp/e2e_probe.kcf -s bench -f window -w 10000 > /tmp/e2e_probe.out 2> /tmp/e2e_probe.err
  local T1=$(date +%s.%N)
  python3 - "$label" $T0 $T1 <<'PY'
import sys, re
label, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
ph = []
for ln in open("/tmp/e2e_probe.err"):
    m = re.match(r"\[(kcftools-amd|kcfx \w+)\]\s+(.*?)\s+([0-9.]+) ms", ln)
    if m: ph.append((m.group(1)[:4], m.group(2)[:40], float(m.group(3))))
big = [f"{a}:{b}={c:.0f}" for a, b, c in ph if c >= 15]
scr = [ln.split(" - ")[-1].strip() for ln in open("/tmp/e2e_probe.out") if "Screened" in ln]
print(f"{label:28s} wall {t1 - t0:6.3f} s | " + "; ".join(big) + (" | " + scr[-1][scr[-1].find(':') + 2:] if scr else ""))
PY
}
run warmup A=1
run warmup A=1
for LB in 27 26 25 24; do
  for i in 1 2 3 4; do run "lbits=$LB back-to-back #$i" KCFX_TABLE_LBITS=$LB; done
done
for LB in 27 25 24; do
  for i in 1 2 3; do ./build_exp/churn 64 2>&1 | tail -1; run "lbits=$LB after 64 GiB churn #$i" KCFX_TABLE_LBITS=$LB; done
done
# the churning process still alive (holding 64 GiB) while the job runs
./build_exp/churn 64 8000 &
sleep 2
for LB in 27 25; do for i in 1 2; do run "lbits=$LB beside a 64 GiB holder #$i" KCFX_TABLE_LBITS=$LB; done; done
wait
for LB in 27 25; do for i in 1 2; do run "lbits=$LB right after the holder #$i" KCFX_TABLE_LBITS=$LB; done; done
