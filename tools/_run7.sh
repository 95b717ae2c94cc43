cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2f; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for d in 4; do
NM355_UP2C_DIAG=$d rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/r2f/a$d -o pmc -- python3 $R/tools/time_up2c.py 3 > $R/gpurun_out/r2f/a$d.log 2>&1
NM355_UP2C_DIAG=$d rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -d $R/gpurun_out/r2f/b$d -o pmc -- python3 $R/tools/time_up2c.py 3 > $R/gpurun_out/r2f/b$d.log 2>&1
echo "== diag $d"; python3 $R/tools/pmc_dump.py $(find $R/gpurun_out/r2f/a$d -name "*.db") conv_up2c_kernel; python3 $R/tools/pmc_dump.py $(find $R/gpurun_out/r2f/b$d -name "*.db") conv_up2c_kernel
done
rm -rf $R/gpurun_out/r2f/a? $R/gpurun_out/r2f/b?
