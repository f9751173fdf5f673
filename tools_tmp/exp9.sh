cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --cpu-baseline-seconds 0 > /dev/null 2>&1
P="python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc7 -o sq1 -- $P > /dev/null 2>gpurun_out/pmc7_d.err
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc7 -o x -- $P > /dev/null 2>gpurun_out/pmc7_e.err
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc7/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'screen_tiles' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1].replace('_counter_collection.csv',''), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
