cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2j; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2j/prof -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2j/prof.log 2>&1
cp $(find $R/gpurun_out/r2j/prof -name "*kernel_stats.csv") $R/gpurun_out/r2j/kernel_stats.csv
find $R/gpurun_out/r2j -name "*kernel_trace.csv" -size +20M -delete
