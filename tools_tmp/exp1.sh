cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 5 --cpu-baseline-seconds 0 > /dev/null 2>&1   # generate + cache dataset
for V in 0 1 2 3 4; do
  echo -n "variant $V: "; KCFX_LIB=$PWD/kcftools_amd/libkcfx_exp.so KCFX_VARIANT=$V timeout 120 python bench.py --steps 10 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], 'ms', d['value']/1e9, 'G/s')"
done
P="python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o a -- $P > /dev/null 2>gpurun_out/pmc_a.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o b -- $P > /dev/null 2>gpurun_out/pmc_b.err
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/pmc1 -o c -- $P > /dev/null 2>gpurun_out/pmc_c.err
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc1 -o d -- $P > /dev/null 2>gpurun_out/pmc_d.err
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o e -- $P > /dev/null 2>gpurun_out/pmc_e.err
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d gpurun_out/pmc1 -o f -- $P > /dev/null 2>gpurun_out/pmc_f.err
ls gpurun_out/pmc1; for f in gpurun_out/pmc_?.err; do echo $f; grep -E "^F|error|exceeds" $f | head -3; done
