cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
python scripts/bench_conv.py > gpurun_out/bench_conv.log 2>&1; cat gpurun_out/bench_conv.log
export REPS=2 WHICH=${WHICH:-fwd}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc -o p1 -- python3 scripts/bench_conv.py dc5 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc -o p2 -- python3 scripts/bench_conv.py dc5 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc -o p3 -- python3 scripts/bench_conv.py dc5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o p4 -- python3 scripts/bench_conv.py dc5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc -o p5 -- python3 scripts/bench_conv.py dc5 > /dev/null 2>&1
ls gpurun_out/pmc
