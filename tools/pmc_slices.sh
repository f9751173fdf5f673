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
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        kind = "slice_list" if "screen_k31c_slice_kernel" in n else "walk_all_or_whole" if "screen_k31c_kernel" in n else None
        if kind is None:
            continue
        a = acc[(kind, int(r["Grid_Size"]), r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (kind, grid, name), (v, c) in sorted(acc.items()):
    print(f"{kind:18s} grid {grid:9d} {name:24s} per launch {v / c:16.0f}  ({c} launches)")
PY
