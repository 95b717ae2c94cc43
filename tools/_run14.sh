cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2k; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2k/prof -o p -- python3 $R/bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r2k/prof.log 2>&1
cp $(find $R/gpurun_out/r2k/prof -name "*kernel_stats.csv") $R/gpurun_out/r2k/kernel_stats.csv
find $R/gpurun_out/r2k -name "*kernel_trace.csv" -size +20M -delete
tail -1 $R/gpurun_out/r2k/prof.log | cut -c1-300
