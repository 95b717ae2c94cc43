cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2c/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2c/prof.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/r2c/pmc -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2c/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r2c/pmc2 -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2c/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_mfma.py $(find gpurun_out/r2c/pmc -name "*.db" | head -1) gpurun_out/r2c/pmc_mfma.json
find gpurun_out/r2c/prof -name "*kernel_stats.csv" | head -1 | xargs head -12
find gpurun_out/r2c -name "*.db" -size +30M -delete; find gpurun_out/r2c -name "*kernel_trace.csv" -size +20M -delete
