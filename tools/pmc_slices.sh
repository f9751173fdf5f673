#!/bin/bash
# What a pass against a table slice executes, ON THE GPU BOX: instruction and wait counters of screen_k31c_slice_kernel (a planned
# reference's walk over the positions whose minimizer class is the slice's) beside screen_k31c_kernel's walk over every position
# (KCFX_SLICE_WALK_ALL inside tools/table_slices_probe.py), one counter group per pass.   usage: pmc_slices.sh <tag> [probe arguments]
TAG=${1:-r06_slices_pmc}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python3 tools/table_slices_probe.py "$@" > $OUT/probe_$i.log 2>&1
done
python3 - <<PY > $OUT/summary.log
import csv, glob, collections
print("# tools/pmc_slices.sh $TAG $* (one MI355X; rocprofv3 --pmc, one counter group per run of tools/table_slices_probe.py)")
print("# per launch = one pass of the whole reference against ONE slice (whole_table: against the unsliced table, the probe's first four launches)")
print("# walk_all = screen_k31c_kernel, every position, the other slices' lanes idle (KCFX_SLICE_WALK_ALL); slice_list = screen_k31c_slice_kernel")
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    whole = sorted({int(r["Dispatch_Id"]) for r in rows if "screen_k31c_kernel" in r["Kernel_Name"]})[:4]
    for r in rows:
        n, d = r["Kernel_Name"], int(r["Dispatch_Id"])
        if "screen_k31c_slice_kernel" in n: kind = "slice_list"
        elif "screen_k31c_kernel" in n: kind = "whole_table" if d in whole else "walk_all"
        else: continue
        a = acc[(kind, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"{'counter':24s} {'whole_table':>16s} {'walk_all':>16s} {'slice_list':>16s}   slice_list / walk_all")
for n in sorted({k[1] for k in acc}):
    v = [acc[(k, n)][0] / max(1, acc[(k, n)][1]) for k in ("whole_table", "walk_all", "slice_list")]
    print(f"{n:24s} {v[0]:16.0f} {v[1]:16.0f} {v[2]:16.0f}   {v[2] / v[1] if v[1] else 0:.2f}")
PY
cat $OUT/summary.log
