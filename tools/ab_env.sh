#!/bin/bash
# Same-box A/B of an environment switch on the bench workload: tools/ab_env.sh VAR A B [rounds]  (on the GPU box)
# e.g. tools/ab_env.sh KCFX_FUSED_REDUCE 0 1 3 -> kernel ms / step ms / G k-mers/s per run, alternating
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python3 bench.py --cpu-baseline-seconds 0 --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],5), 'step_ms', round(d['ms_per_step'],5), 'Gkmers/s', round(d['value']/1e9,1), 'frac', round(d['roofline']['frac'],4))"
  done
done
