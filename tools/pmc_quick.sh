#!/bin/bash
# Quick counter passes around the bench workload, ON THE GPU BOX: HBM requests + instruction mix of the screening kernel.
TAG=${1:-quick}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
Q="python3 bench.py --steps 40 --warmup 3 --cpu-baseline-seconds 0 --no-end-to-end"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o t -- $Q > $OUT/bench_tcc.json 2> $OUT/pmc_tcc.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $Q > /dev/null 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o q -- $Q > /dev/null 2> $OUT/pmc_sq2.err
python3 - <<PY
import csv, glob, collections
for d in ("pmc_tcc", "pmc_sq", "pmc_sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "screen_" in k or "reduce" in k:
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
