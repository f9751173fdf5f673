cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --cpu-baseline-seconds 0 > /dev/null 2>&1
for V in 3 2; do echo -n "VAR$V: "; KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=$V timeout 300 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms')"; done
P="python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0"
for V in 0 3; do
export KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=$V
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc5 -o v${V}_sq1 -- $P > /dev/null 2>gpurun_out/pmc5_d.err
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc5 -o v${V}_x -- $P > /dev/null 2>gpurun_out/pmc5_e.err
done
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc5/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'screen_tiles' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1].replace('_counter_collection.csv',''), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
