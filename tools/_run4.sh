cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
for d in 4 5 6 7 12 15 13; do
NM355_UP2C_DIAG=$d timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2d/b$d.log 2>&1
echo "diag=$d $(tail -1 gpurun_out/r2d/b$d.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])")"
done
