cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --cpu-baseline-seconds 0 > /dev/null 2>&1
echo -n "VAR3: "; KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=3 timeout 300 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms')"
echo -n "generic kernel (VAR10): "; KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=10 timeout 300 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms')"
for L in 0.3 0.4 0.6 0.7; do echo -n "load $L: "; KCFX_TABLE_LOAD=$L timeout 120 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms', d['value']/1e9, 'G/s', d['config']['table_bytes']/1e9,'GB')"; done
P="python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0"
MODE=minimizer
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc3 -o ${MODE}_tcc -- $P > /dev/null 2>gpurun_out/pmc3_c.err
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc3 -o ${MODE}_sq1 -- $P > /dev/null 2>gpurun_out/pmc3_d.err
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o ${MODE}_sq2 -- $P > /dev/null 2>gpurun_out/pmc3_e.err
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc3/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'screen_tiles' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1].replace('_counter_collection.csv',''), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
