#!/bin/bash
export KCFX_TEST_HOOKS=1    # KCFX_TABLE_* / KCFX_LOAD_* are test hooks (kcfx_internal.h: hook())
# Where does an end-to-end getVariations run of the bench workload spend its wall time? (run on the GPU box)
# bench.py leaves the workload's files in $KCFX_BENCH_DIR; the CLI then runs with its phase trace on, and a few times without.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
python3 bench.py --steps 5 --warmup 1 --cpu-baseline-seconds 0 > /dev/null 2>&1
D=${KCFX_BENCH_DIR:-/tmp/kcfx_bench}
P=$D/r30427671_d150000000_s2
run() {
  T0=$(date +%s.%N); env "$@" ./kcftools_amd/bin/kcftools-amd getVariations -r $P.fa -k $P -o /tmp/e2e_trace.kcf -s bench -f window -w 10000 2>&1 | grep "^\[kc"
  python3 -c "import time,sys; print(\"wall %.3f s\" % (time.time() - float(sys.argv[1])), sys.argv[2:])" $T0 "$@"
}
run KCFX_CLI_TRACE=1 KCFX_LOAD_TRACE=1 KCFX_REF_TRACE=1
for i in 1 2 3 4 5 6; do run A=1; done
