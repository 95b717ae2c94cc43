cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2g; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2g/prof -o p -- python3 $R/tools/time_up2c.py 5 > $R/gpurun_out/r2g/prof.log 2>&1
head -8 $(find $R/gpurun_out/r2g/prof -name "*kernel_stats.csv") | cut -c1-150
rm -rf $R/gpurun_out/r2g/prof
