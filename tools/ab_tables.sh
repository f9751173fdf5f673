#!/bin/bash
export KCFX_TEST_HOOKS=1    # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
# Same-box A/B of the table policies on the bench workload (run ON THE GPU BOX): the product table (what one pass gets), the sparse
# one (kcfx_db_open's default) and explicit sizes. Prints one line per configuration.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-ab_tables}
mkdir -p $OUT
B="python3 bench.py --steps 300 --warmup 5 --cpu-baseline-seconds 0 --no-end-to-end --no-realistic --no-cache-state --spin-ms 0"
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" $B ${EXTRA} > $OUT/$name.json 2> $OUT/$name.err
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; t = r["table"]
o = d.get("other_forms") or {}
print(f"{sys.argv[2]:14s} lines 2^{t['lines'].bit_length() - 1} {t['kmers_per_cell']:.3f}/cell stash {t['stash_kmers']:9d}  {d['step_form']} kernel {r['kernel_avg_ms']:.4f} ms  frac {r['frac']:.3f}  plan {r['ref_plan_ms']}  one-pass {r['frac_one_pass']}  two-launch {r['frac_two_launch']}  amortised {r['frac_amortised']} ({(o.get('amortised') or {}).get('kernel_avg_ms')} ms)")
PY
}
EXTRA="--table product" run product KCFX_DUMMY=1
EXTRA="--table sparse" run sparse KCFX_DUMMY=1
EXTRA="--table product" run lbits26 KCFX_TABLE_LBITS=26
EXTRA="--table product" run lbits27 KCFX_TABLE_LBITS=27
EXTRA="--table product" run product2 KCFX_DUMMY=1
