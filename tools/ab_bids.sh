#!/bin/bash
# Same-box A/B of the rank a k-mer bids for its cell with (KCFX_BID_CLASSES builds, K31-only): bench sample, bench cohort,
# read-error-like database, repeat-structured genome.   tools/ab_bids.sh lib1 lib2 ...
for r in 1 2; do
for l in "$@"; do
  export KCFX_LIB=$PWD/$l
  a=$(python3 bench.py --cpu-baseline-seconds 0 --no-end-to-end --steps 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['roofline']['kernel_avg_ms'], d['roofline']['table']['stash_kmers'])")
  b=$(python3 bench.py --samples 8 --cohort --cpu-baseline-seconds 0 --no-end-to-end --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['roofline']['kernel_avg_ms'], d['config']['cohort_table']['stash_records'])")
  c=$(python3 tools/realistic_db_probe.py --error-events 4500000 --steps 100 2>/dev/null | grep "kernel" | sed 's/.*kernel \([0-9.]*\) ms.*/\1/')
  d=$(python3 tools/realistic_db_probe.py --repeats 0.45 --error-events 0 --noise 120000000 --steps 100 2>/dev/null | grep "kernel" | sed 's/.*kernel \([0-9.]*\) ms.*/\1/')
  echo "$l  bench $a | cohort $b | read-error $c | repeats $d"
done
done
