#!/bin/bash
# Counters of the one-launch screening kernel hot / after a read-only sweep / after a dirty sweep (tools/cache_state_probe.py), ON THE GPU BOX
TAG=${1:-r05_cache}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
N=30
python3 tools/cache_state_probe.py $N > $OUT/plain.log 2>&1
for set in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$n -o p -- python3 tools/cache_state_probe.py $N > $OUT/pmc_$n.log 2> $OUT/pmc_$n.err
done
python3 - <<PY
import csv, glob, collections
N = $N
for d in sorted(glob.glob("$OUT/pmc_*/")):
    rows = collections.defaultdict(dict)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "screen_k31c_kernel" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)[5:]
    print(d)
    for name, part in (("hot", ids[:N]), ("read-swept", ids[N:2 * N]), ("write-swept", ids[2 * N:3 * N]), ("hot again", ids[3 * N:4 * N])):
        if not part:
            continue
        keys = sorted(rows[part[0]])
        print("  %-12s" % name, {k: round(sum(rows[i][k] for i in part) / len(part)) for k in keys}, "launches", len(part))
PY
cat $OUT/plain.log | grep hot
