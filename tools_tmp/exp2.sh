cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py --steps 20 2>/dev/null > gpurun_out/bench_v1b.json; python -c "
import json; d=json.load(open('gpurun_out/bench_v1b.json')); print(d['value']/1e9,'G/s', d['roofline']['kernel_avg_ms'],'ms frac',d['roofline']['frac'], d['parity'], d['cpu_baseline']['value']/1e6)"
for V in 1 2 3 4; do
  echo -n "variant $V: "; KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=$V timeout 120 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms', d['value']/1e9, 'G/s')"
done
for L in 0.3 0.4 0.6 0.7; do echo -n "load $L: "; KCFX_TABLE_LOAD=$L timeout 120 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms', d['value']/1e9, 'G/s', d['config']['table_bytes']/1e9,'GB')"; done
