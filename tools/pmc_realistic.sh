#!/bin/bash
# HBM requests and instruction counts of the screening kernel (planned reference) on the inputs of bench.py's `realistic` leg, ON THE
# GPU BOX: gpurun -- 'bash tools/pmc_realistic.sh r05_real'. One rocprofv3 --pmc pass per input (tools/realistic_db_probe.py).
TAG=${1:-r05_real}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() {  # name, probe arguments...
  name=$1; shift
  timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python3 tools/realistic_db_probe.py --steps 30 "$@" > $OUT/$name.log 2> $OUT/$name.err
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmq_$name -o p -- python3 tools/realistic_db_probe.py --steps 30 "$@" > /dev/null 2>&1
}
run headline --error-events 0 --noise 123000000
run read_error --error-events 4500000 --noise 0
run k25 --error-events 0 --noise 123000000 --k 25 --p 5
run repeats --error-events 0 --noise 123000000 --repeats 0.3
python3 - <<PY
import csv, glob, collections
for name in ("headline", "read_error", "k25", "repeats"):
    agg = collections.defaultdict(list)
    for d in ("pmc_", "pmq_"):
        for f in glob.glob("$OUT/%s%s/**/*counter_collection.csv" % (d, name), recursive=True):
            for r in csv.DictReader(open(f)):
                if "screen_k31c_kernel" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    line = open("$OUT/%s.log" % name).read().strip().splitlines()
    print(name, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())}, "launches", len(next(iter(agg.values()), [])))
    for ln in line:
        if "kernel" in ln or "kcfx_db_open" in ln:
            print("   ", ln[:260])
PY
