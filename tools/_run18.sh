cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2m; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r2m/bench.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2m/prof -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r2m/prof.log 2>&1
cp $(find $R/gpurun_out/r2m/prof -name "*kernel_stats.csv") $R/gpurun_out/r2m/kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/r2m/pmc_mfma -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2m/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2m/pmc_f -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2m/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r2m/pmc_w -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2m/pmc3.log 2>&1
cd $R
python tools/pmc_mfma.py $(find gpurun_out/r2m/pmc_mfma -name "*.db") gpurun_out/r2m/pmc_mfma.json | head -8
python tools/pmc_dump.py $(find gpurun_out/r2m/pmc_f -name "*.db") conv_ > gpurun_out/r2m/fetch.txt
python tools/pmc_dump.py $(find gpurun_out/r2m/pmc_w -name "*.db") conv_ > gpurun_out/r2m/write.txt
rm -rf gpurun_out/r2m/pmc_mfma gpurun_out/r2m/pmc_f gpurun_out/r2m/pmc_w; find gpurun_out/r2m -name "*kernel_trace.csv" -size +20M -delete
tail -1 gpurun_out/r2m/bench.log | cut -c1-200
