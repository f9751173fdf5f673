cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -15
for MODE in minimizer keyhash; do
  echo -n "mode $MODE: "; KCFX_TABLE_MODE=$MODE timeout 300 python bench.py --steps 20 --cpu-baseline-seconds 2 2>gpurun_out/bench_$MODE.err | tee gpurun_out/bench_v2_$MODE.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms', d['value']/1e9, 'G/s frac', d['roofline']['frac'], d['parity'], d['config']['table_bytes']/1e9)"
done
tail -3 gpurun_out/bench_minimizer.err
