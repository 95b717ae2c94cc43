cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2a/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_mfma -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_mfma.py gpurun_out/r2a/pmc_mfma gpurun_out/r2a/pmc_mfma.json > gpurun_out/r2a/pmc_summary.log 2>&1
find gpurun_out/r2a/pmc_mfma -name "*.csv" -size +20M -delete
tail -3 gpurun_out/r2a/tests.log; tail -2 gpurun_out/r2a/bench.log | cut -c1-400; cat gpurun_out/r2a/pmc_summary.log
