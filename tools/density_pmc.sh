#!/bin/bash
# Density sweep of the cell table, ON THE GPU BOX: timings in one process (tools/density_sweep.py), then one
# rocprofv3 --pmc pass per table size for the HBM request count, merged into gpurun_out/<tag>_density.csv.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${TAG}_density
mkdir -p $OUT
LS=${2:-27,26,25,24,23}
python3 tools/density_sweep.py --lbits $LS --csv $OUT/timing.csv > $OUT/timing.log 2>&1
for L in ${LS//,/ }; do
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_L$L -o p -- python3 tools/density_sweep.py --one $L --steps 30 > $OUT/pmc_L$L.log 2>&1
done
python3 tools/density_sweep.py --lbits $LS --steps 100 --no-oracle --merge-pmc $OUT --csv gpurun_out/${TAG}_density.csv > $OUT/merged.log 2>&1
cat gpurun_out/${TAG}_density.csv
