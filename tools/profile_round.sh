#!/bin/bash
# Round profile: run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r01'). Writes gpurun_out/<tag>_*;
# copy the summaries into profiles/ with tools/pmc_summary.py afterwards.
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# the same command as the bench line above, minus the CPU legs; enough launches that the cold first ones do not move the average
P="python3 bench.py --steps 300 --warmup 5 --cpu-baseline-seconds 0 --no-end-to-end --no-realistic --no-cache-state --no-resident --spin-ms 0"
Q="python3 bench.py --steps 40 --warmup 3 --cpu-baseline-seconds 0 --no-end-to-end --no-realistic --no-cache-state --no-resident --spin-ms 0"      # counter passes (slower per launch)
# per-kernel time (no counters)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $P > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
# HBM traffic: separate --pmc passes (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $Q > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $Q > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o t -- $Q > /dev/null 2> $OUT/pmc_tcc.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $Q > /dev/null 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o q -- $Q > /dev/null 2> $OUT/pmc_sq2.err
# calibration of FETCH_SIZE on the same access pattern with a known byte count (8 lanes x 16 B random 128-byte lines)
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cal -o c -- ./kcftools_amd/bin/kcfx_membench 4 64 > $OUT/membench.log 2> $OUT/pmc_cal.err
# the bench line that is committed: once more, now that profiles/traffic.json (roofline.traffic, the rocprof average it quotes)
# comes from THIS box's passes
python3 tools/pmc_summary.py $TAG > $OUT/pmc_summary.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
