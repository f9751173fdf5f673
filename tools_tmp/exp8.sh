cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --cpu-baseline-seconds 0 > /dev/null 2>&1
P="python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0"
for L in 0.1 0.2; do
export KCFX_TABLE_LOAD=$L
echo -n "load $L: "; timeout 120 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms', d['value']/1e9, 'G/s', d['config']['table_bytes']/1e9,'GB')"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc6 -o l${L}_sq1 -- $P > /dev/null 2>gpurun_out/pmc6_d.err
done
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc6/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'screen_tiles' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1].replace('_counter_collection.csv',''), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
