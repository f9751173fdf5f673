#!/bin/bash
# Counters of the first and the second screening launch on a fresh handle (tools/first_pass_probe.py), ON THE GPU BOX:
#   gpurun -- 'bash tools/first_pass_pmc.sh r05_first'
TAG=${1:-r05_first}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 tools/first_pass_probe.py 4 > $OUT/plain.log 2>&1
FIRST_PASS_MODE=same python3 tools/first_pass_probe.py 3 > $OUT/same.log 2>&1
for m in other idle allocfree bigfree; do FIRST_PASS_MODE=$m python3 tools/first_pass_probe.py 3 > $OUT/$m.log 2>&1; done
for set in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$n -o p -- python3 tools/first_pass_probe.py 4 > $OUT/pmc_$n.log 2> $OUT/pmc_$n.err
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_*/")):
    rows = collections.defaultdict(dict)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "screen_k31c_kernel" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    print(d)
    for i in ids[-8:]:                                   # the last four cycles: first, second, first, second, ...
        print("  dispatch", i, {k: round(v) for k, v in rows[i].items()})
PY
cat $OUT/plain.log $OUT/same.log $OUT/other.log $OUT/idle.log $OUT/allocfree.log $OUT/bigfree.log | grep cycle
