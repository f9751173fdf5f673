#!/bin/bash
# HBM read requests of the screening kernel BY LEVEL of the look-up (first-choice cells / + second lines / + third lines / + stash)
# at several table densities, ON THE GPU BOX: experiment builds that cut the deferred path short after a level
# (make fast FASTFLAGS=-DKCFX_EXP_CUT=0|1|2 -> build_exp/libkcfx_cut{0,1,2}.so; libkcfx_base.so = the same build uncut).
# Wrong results by design, right request counts.   usage: pmc_levels.sh <tag> "<lbits list>" [extra density_sweep arguments]
TAG=${1:-r06_levels}; LS=${2:-"25 27"}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
for L in $LS; do
  for lib in cut0 cut1 cut2 base; do
    export KCFX_LIB=$PWD/build_exp/libkcfx_$lib.so
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pmc_${lib}_L$L -o p -- python3 tools/density_sweep.py --one $L --steps 30 "$@" > $OUT/${lib}_L$L.log 2>&1
  done
done
unset KCFX_LIB
python3 - <<PY
import csv, glob, json
for L in "$LS".split():
    for lib in ("cut0", "cut1", "cut2", "base"):
        v = []
        for f in glob.glob("$OUT/pmc_%s_L%s/**/*counter_collection.csv" % (lib, L), recursive=True):
            for r in csv.DictReader(open(f)):
                if "screen_k31c_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
                    v.append(float(r["Counter_Value"]))
        row = {}
        for ln in open("$OUT/%s_L%s.log" % (lib, L)):
            if ln.startswith("{"):
                row = json.loads(ln)
        print("lbits", L, lib, "requests/launch", round(sum(v) / max(1, len(v))), "launches", len(v), "kernel_ms", row.get("kernel_ms"), "second%", round(row.get("second_choice_pct", 0), 2), "stash%", round(row.get("stash_pct", 0), 2))
PY
