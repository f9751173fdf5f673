#!/bin/bash
# Counters of the screening kernel in the regime a real database lives in (round 6): a 2.3 Gb genome, the 2.04 G distinct 31-mers of
# its sample in ONE database (the size policy's 3 x 2^27 lines, extended, 0.32 per cell, 50 GiB), the genome's 50 kb tiling. ON THE GPU BOX:
#   bash tools/pmc_large_table.sh <tag> [ref_len]
# One rocprofv3 --pmc pass per counter group; then the composition of the queued look-ups from the KCFX_EXP_STATS build if present.
TAG=${1:-r06_large}; LEN=${2:-2300000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
D=/dev/shm/kcfx_large; mkdir -p $D
[ -f $D/g.kmc_suf ] || ./kcftools_amd/bin/kcfx_synth --out $D/g --ref-len $LEN --seed 41 --sample-seed 42 --noise 0 > $OUT/synth.log 2>&1
cat $D/g.kmc_suf > /dev/null        # (what the generator has just written reads slowly the first time)
python3 tools/large_table_probe.py --prefix $D/g --passes 10 > $OUT/plain.log 2>&1
rocprofv3 --list-avail 2>/dev/null | grep -io "[A-Z0-9_]*UTCL2[A-Z0-9_]*" | sort -u > $OUT/utcl2_counters.txt
declare -A G
G[a]="TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_HIT_sum"
G[b]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
G[c]="SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD"
G[d]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
for g in a b c d; do
  timeout 400 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $OUT/k_$g -o p -- python3 tools/large_table_probe.py --prefix $D/g --passes 3 > $OUT/k_$g.log 2> $OUT/k_$g.err
done
if [ -f build_exp/libkcfx_stats.so ]; then KCFX_LIB=$PWD/build_exp/libkcfx_stats.so python3 tools/large_table_probe.py --prefix $D/g --passes 2 --stats > $OUT/stats.log 2>&1; fi
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/k_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "screen_k31c_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("per launch of screen_k31c_kernel:", {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())}, "launches", {k: len(v) for k, v in agg.items()})
PY
grep -v amdgpu.ids $OUT/plain.log; [ -f $OUT/stats.log ] && grep -v amdgpu.ids $OUT/stats.log; echo "UTCL2 counters on this box:"; cat $OUT/utcl2_counters.txt | tr '\n' ' '
