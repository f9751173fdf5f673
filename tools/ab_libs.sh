#!/bin/bash
# Same-box A/B of library builds (KCFX_LIB=...) on the bench workload: tools/ab_libs.sh rounds lib1 lib2 ...  ("-" = the in-tree build)
# extra environment for every run: AB_ENV="VAR=val VAR2=val"; extra bench.py arguments: AB_ARGS="--steps 300"
N=$1; shift
for i in $(seq 1 $N); do
  for l in "$@"; do
    if [ "$l" = "-" ]; then L=""; else L="KCFX_LIB=$PWD/$l"; fi
    env $L $AB_ENV python3 bench.py --cpu-baseline-seconds 0 --no-end-to-end $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l $AB_ENV', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],5), 'step_ms', round(d['ms_per_step'],5), 'Gkmers/s', round(d['value']/1e9,1), 'frac', round(d['roofline']['frac'],4), 'box', (d.get('box') or {}).get('G_lines_per_s'), 'text_only==planned', (d.get('text_only_reference') or {}).get('identical_to_planned'))"
  done
done
