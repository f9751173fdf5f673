#!/bin/bash
# Profile of the cohort-table step (bench.py --samples 8 --cohort), ON THE GPU BOX: bench line, rocprofv3 --kernel-trace --stats,
# and the --pmc passes (HBM traffic, requests, instruction mix). Summarised by tools/pmc_summary.py <tag> --cohort.
TAG=${1:-r04_cohort}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="--samples 8 --cohort"
python bench.py $B --steps 200 > $OUT/bench.json 2> $OUT/bench.err
P="python3 bench.py $B --steps 200 --warmup 5 --cpu-baseline-seconds 0 --no-end-to-end"
Q="python3 bench.py $B --steps 30 --warmup 3 --cpu-baseline-seconds 0 --no-end-to-end --spin-ms 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $P > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $Q > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $Q > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o t -- $Q > /dev/null 2> $OUT/pmc_tcc.err
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $Q > /dev/null 2> $OUT/pmc_sq.err
ls $OUT
