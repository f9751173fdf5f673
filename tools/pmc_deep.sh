#!/bin/bash
# Memory-path counters of the screening kernel (and of kcfx_membench's "box" pattern beside it), ON THE GPU BOX:
#   bash tools/pmc_deep.sh <tag>
# One rocprofv3 --pmc pass per counter group (a block has few counter slots); prints per-launch averages.
TAG=${1:-deep}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
Q="python3 bench.py --steps 30 --warmup 3 --cpu-baseline-seconds 0 --no-end-to-end"
M="./kcftools_amd/bin/kcfx_membench 16 128 box"
declare -A G
G[a]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
G[b]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
G[c]="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
G[d]="SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
G[e]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
G[f]="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"
G[g]="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum"
G[h]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
for g in a b c d e f g h; do
  timeout 200 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $OUT/k_$g -o p -- $Q > /dev/null 2> $OUT/k_$g.err
  timeout 200 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $OUT/m_$g -o p -- $M > /dev/null 2> $OUT/m_$g.err
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for which, pat in (("kernel", "screen_"), ("membench", "gather_runs")):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s_*/**/*counter_collection.csv" % which[0], recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[which] = {c: round(sum(x) / len(x), 1) for c, x in sorted(agg.items())}
    print(which, json.dumps(res[which], indent=1))
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
