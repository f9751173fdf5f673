cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --cpu-baseline-seconds 0 > /dev/null 2>&1
P="python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc4 -o stats -- $P > /dev/null 2>gpurun_out/pmc4_s.err
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc4 -o sq1 -- $P > /dev/null 2>gpurun_out/pmc4_d.err
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc4 -o sq2 -- $P > /dev/null 2>gpurun_out/pmc4_e.err
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc4/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'screen_tiles' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1].replace('_counter_collection.csv',''), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
cut -c1-150 gpurun_out/pmc4/stats_kernel_stats.csv | head -8
