#!/bin/bash
# Same-box A/B of library builds (KCFX_LIB=...) on the bench workload: tools/ab_libs.sh rounds lib1 lib2 ...  ("-" = the in-tree build)
# extra environment for every run: AB_ENV="VAR=val VAR2=val"; extra bench.py arguments: AB_ARGS="--steps 300"
# prints, per run, the screening kernel's average of each form of the step: one-pass (text-only reference, K1 - K3 in one launch),
# two-launch (plan + screen) and amortised (screen alone)
N=$1; shift
for i in $(seq 1 $N); do
  for l in "$@"; do
    if [ "$l" = "-" ]; then L=""; else L="KCFX_LIB=$PWD/$l"; fi
    env $L $AB_ENV python3 bench.py --cpu-baseline-seconds 0 --no-end-to-end --no-cache-state $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; o=d.get('other_forms') or {}
f={d['step_form']: {'kernel_avg_ms': r['kernel_avg_ms'], 'ms_per_step': d['ms_per_step']}}; f.update(o)
g=lambda k, w: round((f.get(k) or {}).get(w) or 0, 5)
print('$l $AB_ENV', 'one-pass kernel', g('one-pass','kernel_avg_ms'), 'step', g('one-pass','ms_per_step'), '| amortised kernel', g('amortised','kernel_avg_ms'), 'step', g('amortised','ms_per_step'), '| plan', round(r.get('ref_plan_ms') or 0, 5), '| frac', round(r['frac'],4), 'one-pass', round(r.get('frac_one_pass') or 0,4), 'two-launch', round(r.get('frac_two_launch') or 0,4), 'amortised', round(r.get('frac_amortised') or 0,4), 'same', d.get('forms_identical'))"
  done
done
