cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --cpu-baseline-seconds 0 > /dev/null 2>&1
for MODE in minimizer keyhash; do
  echo -n "mode $MODE VAR3: "; KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=3 KCFX_TABLE_MODE=$MODE timeout 300 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms')"
done
echo -n "minimizer VAR2: "; KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=2 timeout 300 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms')"
P="python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0"
for MODE in minimizer keyhash; do
export KCFX_TABLE_MODE=$MODE
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc2 -o ${MODE}_fetch -- $P > /dev/null 2>gpurun_out/pmc2_a.err
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/pmc2 -o ${MODE}_tcc -- $P > /dev/null 2>gpurun_out/pmc2_c.err
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc2 -o ${MODE}_sq1 -- $P > /dev/null 2>gpurun_out/pmc2_d.err
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc2 -o ${MODE}_sq2 -- $P > /dev/null 2>gpurun_out/pmc2_e.err
done
ls gpurun_out/pmc2 | head -40
