#!/bin/bash
# HBM read requests per launch of the screening kernels, for several library builds (KCFX_LIB), ON THE GPU BOX:
#   tools/pmc_requests_ab.sh - build_exp/libkcfx_x.so ...        ("-" = the in-tree build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="python3 bench.py --steps 30 --warmup 3 --cpu-baseline-seconds 0 --no-end-to-end"
for l in "$@"; do
  OUT=gpurun_out/req_ab/$(basename $l .so); rm -rf $OUT; mkdir -p $OUT
  if [ "$l" = "-" ]; then unset KCFX_LIB; else export KCFX_LIB=$PWD/$l; fi
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/tcc -o t -- $Q > /dev/null 2> $OUT/err
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/tcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "screen_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("$l", k, "TCC_EA0_RDREQ_sum per launch", round(sum(v) / len(v)), "launches", len(v))
PY
done
